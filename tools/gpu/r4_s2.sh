# round 4: two-stage ring (32 KB, four workgroups per CU) for the 64x64 products with more than 512 tiles: parity, in-graph latency, step A/B
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_s2.log
: > $L
AVEC_NT_S2=512 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> $L
for v in 0 512 256; do
echo "== AVEC_NT_S2=$v" >> $L
AVEC_NT_S2=$v PYTHONPATH=$PWD python tools/bench_small_gemm.py 2>&1 | grep " res \| ffn1 \| plain " | head -12 >> $L
done
for rep in 1 2; do
for cfg in "AVEC_NT_S2=0" "AVEC_NT_S2=512" "AVEC_NT_S2=256"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_s2.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
