mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/bf16_grad_report.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r2_bf16_report.log
