mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_shift128.log
: > $L
for cfg in AVEC_SHIFT_BM=0 AVEC_SHIFT_BM=128; do
echo "== $cfg" >> $L
env $cfg python tools/abl_conv.py 2>&1 | grep conv >> $L
done
for rep in 1 2; do
for cfg in AVEC_SHIFT_BM=0 AVEC_SHIFT_BM=128; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'])" >> $L
done
done
cat $L
