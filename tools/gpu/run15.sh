mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_grouped.py -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|^$" | tail -40 > gpurun_out/r2_tests15.log
