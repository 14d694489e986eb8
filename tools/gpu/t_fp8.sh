mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "fp8" 2>&1 | tail -30 > gpurun_out/t_fp8.log
