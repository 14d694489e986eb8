"""Summarise a rocprofv3 --pmc SQ_* pass per kernel: python tools/pmc_sq.py <dir>"""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1; dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
names = sorted({c for v in acc.values() for c in v})
print("%-62s %5s %9s " % ("kernel", "n", "ms") + " ".join("%14s" % c[-14:] for c in names))
for k, _ in sorted(dur.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-62s %5d %9.3f " % (k[:62], n[k], dur[k] / 1e6) + " ".join("%14.4g" % (acc[k][c] / n[k]) for c in names))
