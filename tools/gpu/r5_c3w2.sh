mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD:$PWD/tools
L=gpurun_out/r5_c3w2.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "wgrad" 2>&1 | grep -v "amdgpu.ids\|^$" | tail -6 >> $L
python tools/bench_wgrad_c64.py 2>&1 | grep -v amdgpu >> $L
AVEC_C64_WGRAD_RING=0 python tools/bench_wgrad_c64.py 2>&1 | grep -v amdgpu >> $L
cat $L
