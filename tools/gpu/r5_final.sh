# round 5 final: full GPU suite + smoke, then the evidence set (profile_r05.sh all)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -15 > gpurun_out/r5_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5_final_smoke.log 2>&1
tail -2 gpurun_out/r5_final_tests.log; tail -1 gpurun_out/r5_final_smoke.log
bash tools/gpu/profile_r05.sh ${1:-all} > gpurun_out/r5_final_profile.log 2>&1
tail -25 gpurun_out/r5_final_profile.log
