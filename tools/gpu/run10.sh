mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ddp.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r2_tests10.log
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "bench_self" 2>&1 | tail -15 >> gpurun_out/r2_tests10.log
