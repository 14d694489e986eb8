# round 4: where the attention kernels' time goes (ablation builds, tools/build_abl_attn.sh) + the LDS-DMA source-alignment probe
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_attn_abl.log
: > $L
tools/_bin/glds_align_probe >> $L 2>&1
for v in base 1 2 3; do
  echo "== variant $v" >> $L
  if [ $v = base ]; then PYTHONPATH=. python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids >> $L
  else AVEC_LIB_PATH=tools/_bin/libavec_attn_abl_$v.so PYTHONPATH=. python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids >> $L; fi
done
cat $L
