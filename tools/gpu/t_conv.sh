mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -15 > gpurun_out/t_conv.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t_conv.log
