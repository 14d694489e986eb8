# usage: bash tools/gpu/r3_ablib.sh "nameA nameB ..." [reps]  -- same-box A/B of variant libraries tools/_bin/libavec_<name>.so
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_ablib.log
: > $L
for rep in $(seq 1 ${2:-2}); do
for v in $1; do
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_$v.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r3_ablib.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
