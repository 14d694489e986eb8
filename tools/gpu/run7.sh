mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 ./tools/_bin/ubench_load > gpurun_out/r2_ubench_load.log 2>&1
