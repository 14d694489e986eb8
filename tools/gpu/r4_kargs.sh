# round 4: the 64x64 product kernel with its DMA fields as preloaded leading arguments and the argument block read behind the first DMAs: parity, latency, step A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_kargs.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q 2>&1 | tail -2 >> $L
echo "== HEAD" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
echo "== argument block behind the first DMAs" >> $L
PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
for rep in 1 2 3; do
for cfg in "AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so" "AVEC_X=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_kargs.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
