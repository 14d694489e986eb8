# round 4: pair weight-gradient kernel: parity, even work sharing: range-major (default) vs kind-major order, no-DMA ablation
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_wpairs3.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "wgrad" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 >> $L
for cfg in "AVEC_X=0" "AVEC_WP_RANGES=1" "AVEC_WP_RANGES=2" "AVEC_WP_RANGES=8" "AVEC_LIB_PATH=tools/_bin/libavec_wp_abl_1.so" "AVEC_LIB_PATH=tools/_bin/libavec_wp_abl_2.so"; do
  echo "== $cfg" >> $L
  env $cfg PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
