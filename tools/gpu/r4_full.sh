# round 4: full GPU test-suite + same-box A/B of the step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_full.log
: > $L
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -30 >> $L
for rep in 1 2; do
for cfg in "AVEC_POS_GROUP=0" "AVEC_POS_GROUP=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_full.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_full.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
cat $L
