"""Per-kernel roofline table for one training step: joins a rocprofv3 --kernel-trace --stats summary (time) with the --pmc FETCH_SIZE / WRITE_SIZE
summary made by tools/pmc_traffic.py (HBM bytes per launch) and prints, per kernel, average time, HBM traffic, achieved HBM GB/s and the fraction of the
8 TB/s HBM peak.  (MFMA-side numbers of the GEMM family come from bench.py's roofline block: events around each launch, algorithmic FLOPs.)

    python tools/roofline_table.py <kernel_stats.csv> <pmc_traffic.json> <steps traced> > profiles/r02_roofline_table.md"""
import csv, json, sys

stats, traffic, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
tr = json.load(open(traffic))["kernels"]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("Per-kernel view of one AV training step (B = 32, bf16, one MI355X): time from `rocprofv3 --kernel-trace --stats`, HBM bytes from the two `--pmc` passes\n(FETCH_SIZE doubled for gfx950, WRITE_SIZE as reported) -- see profiles/README.md.  The GEMM family's MFMA-side figures are in the bench line of the same round (`rNN_bench_line.json`: `roofline.rows`).\n")
print("| kernel | launches / step | avg us | ms / step | HBM fetch MB / launch | HBM write MB / launch | HBM GB/s | frac of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|")
for r in [r for r in rows if not r["Name"].startswith("__amd_rocclr")][:40]:      # (buffer copies of model initialisation are not part of a step)
    name = r["Name"]
    n, avg = int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3
    t = tr.get(name)
    if t:
        f, w = t["fetch_bytes_per_launch"], (t["write_bytes_per_launch"] or 0.0)
        gbs = (f + w) / (avg * 1e-6) / 1e9
        cols = "%.1f | %.1f | %.0f | %.2f" % (f / 1e6, w / 1e6, gbs, gbs / 8000.0)
    else:
        cols = "– | – | – | –"
    short = name.replace("void ", "").split("(")[0][:70]
    print("| `%s` | %.1f | %.1f | %.3f | %s |" % (short, n, avg, float(r["TotalDurationNs"]) / steps / 1e6, cols))
print()
print("kernel time per step (both streams): %.1f ms" % (tot / steps / 1e6))
