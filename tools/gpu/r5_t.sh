mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
timeout 1500 python -m pytest $TESTS -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -60 > gpurun_out/r5_t.log
cat gpurun_out/r5_t.log
