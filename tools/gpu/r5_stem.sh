mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_stem.log
: > $L
timeout 900 python -m pytest tests -m gpu -x -q -k "stem and not audio" 2>&1 | tail -4 >> $L
for cfg in "$@"; do
env $cfg python tools/bench_stem.py 2>&1 | grep -v amdgpu | grep stem3p | sed "s/^/$cfg /" >> $L
done
cat $L
