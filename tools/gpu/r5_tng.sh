mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_tng.log
: > $L
for w in 64 128 256 512 1024; do
echo "== AVEC_TNG_WGS=$w" >> $L
AVEC_TNG_WGS=$w python tools/bench_tn_grouped.py 2>&1 | grep -v amdgpu >> $L
done
cat $L
