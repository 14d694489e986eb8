mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
: > gpurun_out/r3_stem_wabl.log
for a in ${ABLS:-0 8 1 2 4 3 5 6 7}; do AVEC_S3W_ABL=$a timeout 120 python tools/bench_stem_wgrad.py 2>&1 | grep ABL >> gpurun_out/r3_stem_wabl.log; done
cat gpurun_out/r3_stem_wabl.log
