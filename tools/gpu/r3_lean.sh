cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
echo "== general kernel"; AVEC_NO_LEAN_NT=1 python tools/bench_small_gemm.py 2>&1 | grep -v amdgpu.ids
echo "== lean kernel"; python tools/bench_small_gemm.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests -q -m gpu -x -k "gemm or linear or golden or block or ffn" 2>&1 | tail -3
