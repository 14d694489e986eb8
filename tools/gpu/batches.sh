mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/batches.log
: > $L
for b in 1 5 16 48 64; do
python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/batches.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L || echo "B=$b FAILED" >> $L
done
python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --fp8 2>>gpurun_out/batches.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 fp8', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
