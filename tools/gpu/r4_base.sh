# round 4, call 1: new parity tests + same-box baseline of the round-3 tree (bench B=32 twice, B=1)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_base.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | grep -v "amdgpu.ids\|^$" | tail -25 >> $L
for rep in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_base.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=32', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_base.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
cat $L
