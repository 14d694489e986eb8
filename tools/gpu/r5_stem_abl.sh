mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_stem_abl.log
: > $L
for r in 1 0; do
for a in 0 1 2 3 4 7; do
AVEC_S3W_ROLES=$r AVEC_S3W_ABL=$a python tools/bench_stem_wgrad.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/roles=$r /" >> $L
done
done
cat $L
