mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_slab2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "slab or grouped_wgrad" 2>&1 | tail -5 >> $L
python tools/bench_slab.py 2>&1 | grep -v amdgpu >> $L
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_c3strace.so python tools/slab_trace.py 2>&1 | grep -v amdgpu >> $L
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_c3strace.so python tools/slab_trace.py bwd 2>&1 | grep -v amdgpu >> $L
cat $L
