# round 4: same-box A/B of the step: $1.. = env settings (quoted strings), each run twice alternating
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_ab.log
: > $L
for rep in 1 2; do
for cfg in "$@"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_ab.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r4_ab.err >> $L
cat $L
