mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -40 > gpurun_out/r2_tests_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r2_smoke.log 2>&1
