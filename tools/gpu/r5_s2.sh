mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_s2.log
: > $L
for cfg in "$@"; do
echo "== $cfg" >> $L
env $cfg python tools/bench_s2.py 2>&1 | grep -v amdgpu >> $L
done
cat $L
