# usage: bash tools/gpu/r3_ab.sh "ENV_A" "ENV_B" [reps]   -- same-box A/B of the graphed bench step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_ab.log
: > $L
for rep in $(seq 1 ${3:-2}); do
for cfg in "$1" "$2"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r3_ab.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
