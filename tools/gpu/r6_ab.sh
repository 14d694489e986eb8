# round 6: same-box A/B of the step over env settings ($@, each run twice alternating); TESTS="..." runs a pytest selection first
mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r06/ab_${TAG:-x}.log
: > $L
if [ -n "$TESTS" ]; then timeout 2400 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -6 >> $L; fi
for rep in 1 2; do
for cfg in "$@"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r06/ab_${TAG:-x}.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r06/ab_${TAG:-x}.err >> $L
cat $L
