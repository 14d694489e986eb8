# round-4 evidence: default bench line, rocprofv3 kernel stats of the graph-replayed step, PMC traffic passes, timeline trace
mkdir -p gpurun_out/r04
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
if [ "${1:-all}" = "all" ] || [ "$1" = "bench" ]; then
python bench.py 2>$O/bench_default.err | grep "^{" > $O/r04_bench_line.json
fi
# kernel trace + stats of the graph-replayed step (1 eager warm-up step + 4 replays = 5 steps traced; model set-up copies included)
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r04 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_bench.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r04_kernel_stats.csv \;
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[int(len(rows)*0.55):]
out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Stream_Id",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Grid_Size_Y",""), r.get("LDS_Block_Size","")] for r in keep]
gzip.open("/tmp/r04_trace_tail.json.gz", "wt").write(json.dumps(out))
PY
cp /tmp/r04_trace_tail.json.gz $O/r04_trace_tail.json.gz
python tools/timeline.py /tmp/r04_trace_tail.json.gz > $O/timeline.txt 2>&1 || true
if [ "${1:-all}" = "all" ] || [ "$1" = "pmc" ]; then
( cd /tmp && rm -rf /tmp/pmc_f && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_f.log 2>&1 )
( cd /tmp && rm -rf /tmp/pmc_w && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_w.log 2>&1 )
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/r04_pmc_traffic.json > $O/pmc_top.txt 2>&1
fi
head -45 $O/r04_kernel_stats.csv | cut -c1-150
tail -30 $O/timeline.txt
if [ "${1:-all}" = "all" ]; then
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|OMP_NUM\|^\*\*\*\*\|^$" | tail -40 > $O/tests_all.log
tail -5 $O/tests_all.log
AVEC_STAMPS=1 timeout 600 python tools/step_stamps.py 2>&1 | grep -v amdgpu > $O/r04_step_stamps.txt
fi
