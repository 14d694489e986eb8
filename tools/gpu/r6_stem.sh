# round 6: visual stem kernels -- parity tests, then the forward / backward kernels alone (with ablation bits)
mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
O=gpurun_out/r06/stem_${TAG:-x}.log
: > $O
timeout 900 python -m pytest tests -m gpu -x -q -k "stem and not audio" 2>&1 | tail -5 >> $O
for a in ${FWD_ABL:-0 8 4 12}; do AVEC_S3P_ABL=$a timeout 120 python tools/bench_stem_abl.py 2>&1 | grep ABL >> $O; done
for a in ${BWD_ABL:-0}; do AVEC_S3W_ABL=$a timeout 120 python tools/bench_stem_wgrad.py 2>&1 | grep ABL >> $O; done
timeout 300 python tools/bench_stem.py 2>&1 | grep stem3p | tail -1 >> $O
cat $O
