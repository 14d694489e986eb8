mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt 2>&1 || rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc_avail.txt 2>&1
