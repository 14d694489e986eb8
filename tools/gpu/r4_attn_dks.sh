# round 4: attention kernels with the channel K-steps as a compile-time constant (unrolled operand reads): parity, in-graph latency, step A/B against HEAD
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_attn_dks.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_grouped.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== HEAD" >> $L
AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so PYTHONPATH=$PWD timeout 300 python tools/bench_attention.py >> $L 2>&1
echo "== unrolled K-steps" >> $L
PYTHONPATH=$PWD timeout 300 python tools/bench_attention.py >> $L 2>&1
for rep in 1 2; do
for cfg in "AVEC_LIB_PATH=$PWD/tools/_bin/libavec_head.so" "AVEC_X=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_attn_dks.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r4_attn_dks.err >> $L
cat $L
