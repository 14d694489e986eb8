mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "visual_stem" > gpurun_out/r3_stem_test.log 2>&1
tail -15 gpurun_out/r3_stem_test.log
: > gpurun_out/r3_stem_abl.log
for a in 0 8 4 12 2 14; do AVEC_S3P_ABL=$a timeout 120 python tools/bench_stem_abl.py 2>&1 | grep ABL >> gpurun_out/r3_stem_abl.log; done
cat gpurun_out/r3_stem_abl.log
timeout 300 python tools/bench_stem.py 2>&1 | grep stem3 | tail -4
