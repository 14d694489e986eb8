mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "tn_grouped" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/tng.log
