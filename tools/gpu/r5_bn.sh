mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python tools/bench_bn.py 2>&1 | grep -v amdgpu > gpurun_out/r5_bn.log
cat gpurun_out/r5_bn.log
