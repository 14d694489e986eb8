mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "resnet or full_model or gemm" > gpurun_out/r3_q_test.log 2>&1
tail -12 gpurun_out/r3_q_test.log
bash tools/gpu/r3_ab.sh AVEC_BNB_FUSE=1 AVEC_BNB_FUSE=0 3
