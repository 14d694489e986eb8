mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r3_q_test.log 2>&1
tail -6 gpurun_out/r3_q_test.log
: > gpurun_out/r3_variants.log
timeout 600 python tools/bench_variants.py --only lrs2_main --steps 16 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 900 python tools/bench_variants.py --only lrs2_main --steps 16 --graphs 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 900 python tools/bench_variants.py --only lrs2_main --steps 16 --graphs --bucket 25 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 600 python tools/bench_variants.py --only lrs2_pre --steps 16 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 900 python tools/bench_variants.py --only lrs2_pre --steps 16 --graphs 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 600 python tools/bench_variants.py --only av15s --steps 8 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
timeout 900 python tools/bench_variants.py --only av15s --steps 8 --graphs 2>&1 | grep "^{" >> gpurun_out/r3_variants.log
cat gpurun_out/r3_variants.log
