mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/ipc_probe.py > gpurun_out/r2_ipc_probe.log 2>&1
echo rc=$? >> gpurun_out/r2_ipc_probe.log
