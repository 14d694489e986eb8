mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "wgrad or slab" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 > gpurun_out/r5_last.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP', d['ms_per_step'], d['value'], d['config']['loss'])" >> gpurun_out/r5_last.log
cat gpurun_out/r5_last.log
