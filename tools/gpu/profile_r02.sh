mkdir -p gpurun_out/r02
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
# 1. default bench line (the driver's command)
python bench.py 2>$O/bench_default.err | grep "^{" > $O/r02_bench_line.json
# 2. kernel trace + stats of the graph-replayed step
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_bench.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r02_kernel_stats.csv \;
# 3. PMC passes (separate, kernel-trace only)
( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_f.log 2>&1 )
( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_w.log 2>&1 )
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/r02_pmc_traffic.json > $O/pmc_top.txt 2>&1
# 4. smoke + whole GPU suite
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -40 > $O/tests_all.log
