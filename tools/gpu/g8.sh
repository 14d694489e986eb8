mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 300 ./tools/_bin/ubench_gemm8 > gpurun_out/g8.log 2>&1
