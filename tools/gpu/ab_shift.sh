mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/ab_shift.log
: > $L
for rep in 1 2 3; do
for v in 128 0; do
AVEC_SHIFT_BM=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BM$v', d['ms_per_step'], d['value'])" >> $L
done
done
