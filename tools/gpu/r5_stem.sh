mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_stem.log
: > $L
timeout 900 python -m pytest tests -m gpu -x -q -k "stem and not audio" 2>&1 | tail -4 >> $L
python tools/bench_stem.py 2>&1 | grep -v amdgpu | grep stem3p >> $L
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_base.so python tools/bench_stem.py 2>&1 | grep -v amdgpu | grep stem3p | sed 's/^/base /' >> $L
cat $L
