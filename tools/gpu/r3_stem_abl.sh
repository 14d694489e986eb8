mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
: > gpurun_out/r3_stem_abl.log
for a in 0 8 4 12 2 14 16 30; do AVEC_S3P_ABL=$a timeout 120 python tools/bench_stem_abl.py 2>&1 | grep ABL >> gpurun_out/r3_stem_abl.log; done
cat gpurun_out/r3_stem_abl.log
