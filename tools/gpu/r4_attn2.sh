mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_attn2.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_grouped.py tests/test_streaming.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== full" >> $L
PYTHONPATH=. python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== staging only" >> $L
AVEC_LIB_PATH=tools/_bin/libavec_attn_abl_1.so PYTHONPATH=. python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
