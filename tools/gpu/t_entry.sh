mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_entry_point.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_entry.log
