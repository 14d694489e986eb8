mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[int(len(rows)*0.55):]
out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Stream_Id",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Grid_Size_Y",""), r.get("LDS_Block_Size","")] for r in keep]
gzip.open(sys.argv[1].rsplit("/",1)[0] + "/trace_tail.json.gz", "wt").write(json.dumps(out))
PY
cp $(dirname $f)/trace_tail.json.gz $GRAFT_REPO_ROOT/gpurun_out/r2b_trace_tail.json.gz
