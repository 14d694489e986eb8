# round 5: eight-stage ring for the deep-K 64x64 products with <= 256 tiles: parity, in-graph latency, step A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_s8.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -m gpu -x -q 2>&1 | tail -3 >> $L
for v in 0 512; do
echo "== AVEC_NT_S8=$v" >> $L
AVEC_NT_S8=$v python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " >> $L
done
for rep in 1 2 3; do
for cfg in "AVEC_NT_S8=0" "AVEC_NT_S8=512"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r5_s8.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
tail -3 gpurun_out/r5_s8.err >> $L
cat $L
