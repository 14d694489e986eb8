mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tools
L=gpurun_out/r5_c3w.log
: > $L
python tools/bench_wgrad_c64.py 2>&1 | grep -v amdgpu >> $L
for n in "$@"; do
AVEC_LIB_PATH=tools/_bin/libavec_c3wabl_$n.so python tools/bench_wgrad_c64.py 2>&1 | grep -v amdgpu >> $L
done
cat $L
