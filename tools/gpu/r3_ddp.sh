mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r3_ddp_test.log 2>&1
tail -15 gpurun_out/r3_ddp_test.log
