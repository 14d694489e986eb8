# round 5: stage-1 slab kernels: parity + isolated times (+ optional step A/B)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_slab.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "slab or grouped_wgrad" 2>&1 | tail -5 >> $L
python tools/bench_slab.py 2>&1 | grep -v amdgpu >> $L
if [ "$1" = "step" ]; then
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r5_slab.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
fi
cat $L
