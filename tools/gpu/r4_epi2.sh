# round 4: epilogue operands prefetched in front of the K loop (64x64 product) + the 128x64 three-stage variant (AVEC_NT_P128=1): parity, in-graph latency, step A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_epi2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== batched epilogue loads only (HEAD)" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_batched.so PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
echo "== prefetched epilogue operands" >> $L
PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
for rep in 1 2; do
for cfg in "AVEC_LIB_PATH=$PWD/tools/_bin/libavec_batched.so" "AVEC_X=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_epi2.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r4_epi2.err >> $L
cat $L
