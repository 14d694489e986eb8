# round 4: latency pass over the conformer chain (one memory round trip per kernel): depthwise-conv taps in registers, bn_finalize / LayerNorm-backward / product-epilogue
# operands requested up front.  Parity tests, per-launch block trace (HEAD library vs this tree), same-box A/B of the step.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_lat.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_layers.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== block trace, HEAD library" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_batched.so PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 32 100 256 >> $L 2>&1
echo "== block trace, this tree" >> $L
PYTHONPATH=$PWD timeout 300 python tools/block_trace.py 32 100 256 >> $L 2>&1
for rep in 1 2; do
for cfg in "AVEC_LIB_PATH=$PWD/tools/_bin/libavec_batched.so" "AVEC_X=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_lat.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r4_lat.err >> $L
cat $L
