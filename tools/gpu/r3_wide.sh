mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resnet or conv or full_model" > gpurun_out/r3_q_test.log 2>&1
tail -3 gpurun_out/r3_q_test.log
bash tools/gpu/r3_ablib.sh "base wide2" 3
