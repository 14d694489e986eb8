# round 4: batched residual / act'(z) loads in the epilogue of the 64-column products: parity tests, in-graph latency per product, same-box A/B of the step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_epi.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== base" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_base.so PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
echo "== batched epilogue loads" >> $L
PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -18 >> $L
for rep in 1 2; do
for lib in tools/_bin/libavec_base.so avec_amd/libavec_hip.so; do
AVEC_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_epi.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r4_epi.err >> $L
cat $L
