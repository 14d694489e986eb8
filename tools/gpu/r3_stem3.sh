mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "visual_stem" > gpurun_out/r3_stem_test.log 2>&1
tail -15 gpurun_out/r3_stem_test.log
timeout 300 python tools/bench_stem.py 2>&1 | grep stem3 | tail -4
