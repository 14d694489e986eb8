# round 5: kernel traces (queue ids, timestamps) of the replayed step for two env settings ($1, $2): timeline views + compressed trace tails
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
i=$((i+1))
( cd /tmp && rm -rf /tmp/prof$i && env $cfg timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof$i -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r5_trace2_$i.log 2>&1 )
f=$(find /tmp/prof$i -name "*kernel_trace.csv" | head -1)
python - "$f" /tmp/tt$i.json.gz <<'PY'
import csv, sys, gzip, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[int(len(rows)*0.55):]
out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Stream_Id",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Grid_Size_Y",""), r.get("LDS_Block_Size","")] for r in keep]
gzip.open(sys.argv[2], "wt").write(json.dumps(out))
PY
cp /tmp/tt$i.json.gz gpurun_out/r5_trace2_$i.json.gz
echo "== $cfg" > gpurun_out/r5_timeline_$i.txt
python tools/timeline.py /tmp/tt$i.json.gz 0.5 >> gpurun_out/r5_timeline_$i.txt 2>&1
head -12 gpurun_out/r5_timeline_$i.txt
done
