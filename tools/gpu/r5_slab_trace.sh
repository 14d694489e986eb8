mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_c3strace.so python tools/slab_trace.py 2>&1 | grep -v amdgpu > gpurun_out/r5_slab_trace.log
cat gpurun_out/r5_slab_trace.log
