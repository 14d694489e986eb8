# round 4: pair-formulation weight-gradient kernel: parity tests + isolated timings against the slab kernel
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_wpairs.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -k "wgrad" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -12 >> $L
echo "== pairs" >> $L
PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== slab (AVEC_NO_WGRAD_PAIRS=1)" >> $L
AVEC_NO_WGRAD_PAIRS=1 PYTHONPATH=. timeout 300 python tools/bench_wgrad_wide.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
