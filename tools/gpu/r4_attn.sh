# round 4: attention kernels -- parity tests + isolated latencies
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_attn.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_grouped.py tests/test_streaming.py -x -q -k "attention or attn or block or grouped or streaming or causal" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -12 >> $L
PYTHONPATH=. python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
