mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  if [ $mode = eager ]; then F="--eager"; else F=""; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$mode -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing $F > /tmp/cp_$mode.log 2>&1
  f=$(find /tmp/cp_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode" >> $R/gpurun_out/r3_copies2.log
  grep -i "copyBuffer\|fillBuffer\|elementwise" $f >> $R/gpurun_out/r3_copies2.log
  grep "^{" /tmp/cp_$mode.log | head -c 300 >> $R/gpurun_out/r3_copies2.log
done
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/cp_hip -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/cp_hip.log 2>&1
f=$(find /tmp/cp_hip -name "*hip_api_stats.csv" | head -1)
echo "== hip api (graph)" >> $R/gpurun_out/r3_copies2.log
head -30 $f >> $R/gpurun_out/r3_copies2.log
