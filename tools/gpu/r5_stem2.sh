mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_stem2.log
: > $L
timeout 900 python -m pytest tests -m gpu -x -q -k "stem and not audio" 2>&1 | tail -4 >> $L
for a in 0 1 2 3 4 7; do
AVEC_S3W_ABL=$a python tools/bench_stem_wgrad.py 2>&1 | grep -v amdgpu | tail -1 >> $L
done
AVEC_S3W_ROLES=0 python tools/bench_stem_wgrad.py 2>&1 | grep -v amdgpu | tail -1 | sed 's/^/lockstep /' >> $L
cat $L
