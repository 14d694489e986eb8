# round 4: chain kernels -- parity tests, micro-benchmark, same-box A/B of the graphed bench step
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_chain.log
: > $L
timeout 1500 python -m pytest tests/test_gpu_round4.py -x -q -k "chain or ln_gemm or attn" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -25 >> $L
timeout 600 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids >> $L
for rep in 1 2; do
for cfg in "AVEC_FFN_CHAIN=0 AVEC_LN_GEMM=0" "AVEC_FFN_CHAIN=1 AVEC_LN_GEMM=0" "AVEC_FFN_CHAIN=1 AVEC_LN_GEMM=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_chain.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
done
cat $L
