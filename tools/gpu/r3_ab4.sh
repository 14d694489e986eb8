# usage: bash tools/gpu/r3_ab4.sh reps "ENV_A" "ENV_B" ...   -- same-box comparison of several environments of the graphed bench step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_ab.log
: > $L
reps=$1; shift
for rep in $(seq 1 $reps); do
for cfg in "$@"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r3_ab.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
