mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('600 steps', d['ms_per_step'], d['value'], 'loss', d['config']['loss'])" > gpurun_out/long.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('25 steps', d['ms_per_step'], d['value'], 'loss', d['config']['loss'])" >> gpurun_out/long.log
