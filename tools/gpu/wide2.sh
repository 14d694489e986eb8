mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/wide2.log
: > $L
for rep in 1 2 3; do
for n in 0 1 3; do
AVEC_NT_WIDE=$n python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIDE=$n', d['ms_per_step'], d['value'])" >> $L
done
done
