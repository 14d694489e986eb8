# round 5, experiment 2: register-direct epilogue of the shifted-window convolution: parity, isolated times, step A/B (AVEC_SHIFT_NO_TR=1 = staged epilogue);
# XCD-aware tile order of the plain product is in both arms
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_e2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py -m gpu -x -q -k "shift or conv3x3" 2>&1 | tail -15 >> $L
echo "== staged" >> $L
AVEC_SHIFT_NO_TR=1 python tools/abl_conv.py 2>&1 | grep conv >> $L
echo "== direct" >> $L
python tools/abl_conv.py 2>&1 | grep conv >> $L
for rep in 1 2; do
for cfg in "AVEC_SHIFT_NO_TR=1" "AVEC_X=0"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r5_e2.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
python tools/bench_small_gemm.py 2>&1 | tail -19 >> $L
tail -3 gpurun_out/r5_e2.err >> $L
cat $L
