mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python tools/block_trace.py 32 100 256 2>&1 | grep -v amdgpu > gpurun_out/r5_block_trace.txt
python tools/block_trace.py 32 100 360 2>&1 | grep -v amdgpu >> gpurun_out/r5_block_trace.txt
cat gpurun_out/r5_block_trace.txt
