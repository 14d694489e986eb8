mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_chain3.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -k "chain or ln_gemm or dropout or drop or block or layernorm" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -12 >> $L
timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids | grep "M=3200\|M=1600" >> $L
timeout 600 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids >> $L
for cfg in "AVEC_FFN_CHAIN=0 AVEC_LN_GEMM=0" "AVEC_FFN_CHAIN=0 AVEC_LN_GEMM=1" "AVEC_FFN_CHAIN=1 AVEC_LN_GEMM=1"; do
env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>gpurun_out/r4_chain3.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" >> $L
done
cat $L
