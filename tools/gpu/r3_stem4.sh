mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "visual_stem" > gpurun_out/r3_stem_test.log 2>&1
tail -15 gpurun_out/r3_stem_test.log
: > gpurun_out/r3_stem_wabl.log
for a in 0 8 1 2 4 3 5 6 7; do AVEC_S3W_ABL=$a timeout 120 python tools/bench_stem_wgrad.py 2>&1 | grep ABL >> gpurun_out/r3_stem_wabl.log; done
cat gpurun_out/r3_stem_wabl.log
