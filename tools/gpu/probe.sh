mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 60 ./tools/_bin/glds_align_probe > gpurun_out/probe.log 2>&1
