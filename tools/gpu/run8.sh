mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(AVEC_FFN_DBG=32 timeout 300 python tools/ffn_stamps.py 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/bench_ffn.py 2>&1 | grep -v amdgpu; timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "ffn" 2>&1 | tail -5) > gpurun_out/r2_ffn_stamps.log
