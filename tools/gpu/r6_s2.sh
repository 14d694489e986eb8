# round 6: stride-2 shifted-window kernels -- parity test, then the isolated bench with and without them
mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
O=gpurun_out/r06/s2.log
: > $O
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "stride2" 2>&1 | tail -15 >> $O
echo "== new" >> $O
timeout 300 python tools/bench_s2.py 2>&1 | grep -v amdgpu >> $O
echo "== AVEC_NO_CONV_S2=1" >> $O
AVEC_NO_CONV_S2=1 timeout 300 python tools/bench_s2.py 2>&1 | grep -v amdgpu | grep "k3" >> $O
cat $O
