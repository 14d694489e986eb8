mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
: > gpurun_out/flaky.log
timeout 2400 python tools/ddp_flaky.py 100 2>&1 | grep -v "amdgpu.ids\|socket\|Gloo\|OMP_NUM\|^\*\*\*" >> gpurun_out/flaky.log
