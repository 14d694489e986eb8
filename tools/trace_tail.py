"""Compact tail of a rocprofv3 --kernel-trace CSV (the last ~45 % of the launches: the final graph replays) for tools/timeline.py / tools/chain_view.py.
    python tools/trace_tail.py <rocprof output dir> <out.json.gz>"""
import csv
import glob
import gzip
import json
import os
import sys


def main():
    fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = rows[int(len(rows) * 0.55):]
    out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", ""),
            r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", "")] for r in keep]
    gzip.open(sys.argv[2], "wt").write(json.dumps(out))
    print("trace tail: %d of %d launches -> %s" % (len(out), len(rows), sys.argv[2]))


if __name__ == "__main__":
    main()
