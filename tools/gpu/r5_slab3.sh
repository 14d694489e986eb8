mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_slab3.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "slab or grouped_wgrad" 2>&1 | tail -3 >> $L
python tools/bench_slab.py 2>&1 | grep -v amdgpu >> $L
for rep in 1 2; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r5_slab3.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP new', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_base.so python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r5_slab3.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP base', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
cat $L
