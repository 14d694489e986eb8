mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -k graphed 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -60 > gpurun_out/r2_tests11.log
