set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -f gpurun_out/r2_bench3.log
for cfg in "AVEC_AUDIO_FIRST=1" "AVEC_AUDIO_FIRST=0" "AVEC_AUDIO_FIRST=0 AVEC_DEFER_TN_MAX=64" "AVEC_AUDIO_FIRST=0 AVEC_DEFER_TN_MAX=128" "AVEC_AUDIO_FIRST=1 AVEC_DEFER_TN_MAX=128" "AVEC_AUDIO_FIRST=0 AVEC_BRANCH_STREAMS=0"; do
  echo "== $cfg" >> gpurun_out/r2_bench3.log
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> gpurun_out/r2_bench3.log
done
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, gzip, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[int(len(rows)*0.55):]
out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Stream_Id",""), r.get("Workgroup_Size_X",""), r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""), r.get("LDS_Block_Size","")] for r in keep]
gzip.open("/tmp/prof/trace_tail.json.gz", "wt").write(json.dumps(out))
PY
cp /tmp/prof/trace_tail.json.gz $GRAFT_REPO_ROOT/gpurun_out/r2_trace_tail3.json.gz
