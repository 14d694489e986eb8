mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
: > gpurun_out/r3_variants_final.log
timeout 600 python tools/bench_variants.py --only lrs2_main --steps 16 2>&1 | grep "^{" >> gpurun_out/r3_variants_final.log
timeout 900 python tools/bench_variants.py --only lrs2_main --steps 16 --graphs 2>&1 | grep "^{" >> gpurun_out/r3_variants_final.log
timeout 600 python tools/bench_variants.py --only lrs2_pre --steps 16 2>&1 | grep "^{" >> gpurun_out/r3_variants_final.log
timeout 600 python tools/bench_variants.py --only av15s --steps 8 2>&1 | grep "^{" >> gpurun_out/r3_variants_final.log
timeout 900 python tools/bench_variants.py --only av15s --steps 8 --graphs 2>&1 | grep "^{" >> gpurun_out/r3_variants_final.log
cat gpurun_out/r3_variants_final.log
