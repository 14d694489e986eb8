# round 4: full GPU test-suite + bench + variants
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_full.log
: > $L
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -30 >> $L
for rep in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_full.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=32', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
python bench.py --batch 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_full.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
timeout 900 python tools/bench_variants.py --only lrs2_main --graphs 2>>gpurun_out/r4_full.err | grep "^{" >> $L
timeout 900 python tools/bench_variants.py --only lrs2_main_bucketed --graphs 2>>gpurun_out/r4_full.err | grep "^{" >> $L
timeout 900 python tools/bench_variants.py --only lrs2_main_bucketed 2>>gpurun_out/r4_full.err | grep "^{" >> $L
cat $L
