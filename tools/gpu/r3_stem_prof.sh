mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o p -- python $R/tools/bench_stem.py > /tmp/sp.log 2>&1
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r3_stem_kernel_stats.csv
head -25 $f
