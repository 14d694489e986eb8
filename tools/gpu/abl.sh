mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/abl.log
: > $L
python tools/abl_gemm.py 2>&1 | grep -v amdgpu.ids | grep "lib\|conv" >> $L
for n in 1 2 64 128; do
  AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_abl_$n.so timeout 300 python tools/abl_gemm.py 2>&1 | grep -v "amdgpu.ids" | grep "lib\|conv" >> $L
done
