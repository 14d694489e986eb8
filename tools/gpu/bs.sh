mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/bs.log
: > $L
for v in 1 0; do
AVEC_BRANCH_STREAMS=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BRANCH=$v', d['ms_per_step'], d['value'])" >> $L
done
