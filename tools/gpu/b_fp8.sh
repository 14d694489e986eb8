mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/b_fp8.log
: > $L
for v in "" "--fp8"; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing $v 2>>gpurun_out/b_fp8.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['dtype'], d['config']['loss'])" >> $L
done
