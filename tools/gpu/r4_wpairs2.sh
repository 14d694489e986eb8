# round 4: pair weight-gradient kernel: parity + ablations
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_wpairs2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "wgrad" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 >> $L
for v in base 1 2 3 4; do
  echo "== variant $v (WP_ABL bits: 1 no DMA after the first stage, 2 no MFMA, 4 no atomics)" >> $L
  if [ $v = base ]; then PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids >> $L
  else AVEC_LIB_PATH=tools/_bin/libavec_wp_abl_$v.so PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids >> $L; fi
done
cat $L
