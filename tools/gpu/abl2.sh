mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/abl2.log
: > $L
export AVEC_NO_CONV_SHIFT=1
for n in 5 0; do
for cfg in "AVEC_NT_STG=3" "AVEC_NT_STG=4" "AVEC_NT_RB=128" "AVEC_NT_RB=64 AVEC_NT_STG=2"; do
  if [ $n = 0 ]; then env $cfg timeout 300 python tools/abl_gemm.py 2>&1 | grep -v "amdgpu.ids" | grep "lib\|conv" >> $L
  else env $cfg AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_abl_$n.so timeout 300 python tools/abl_gemm.py 2>&1 | grep -v "amdgpu.ids" | grep "lib\|conv" >> $L; fi
done
done
