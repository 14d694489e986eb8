mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_chain2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -k "chain or ln_gemm" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -8 >> $L
timeout 300 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids >> $L
timeout 600 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
