mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_stem.py 2>&1 | grep stem3 | tail -4
bash tools/gpu/r3_ab.sh AVEC_STEM3P=1 AVEC_STEM3P=0 2
