"""Timeline view of one training step from a rocprofv3 --kernel-trace CSV tail (tools/gpu/run1.sh writes gpurun_out/*trace_tail.json.gz):
per hardware queue the busy time, the idle gaps, and per 1-ms window the kernels that filled it.  usage: python tools/timeline.py <trace_tail.json.gz> [win_ms]"""
import collections
import gzip
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    rows = json.loads(gzip.open(sys.argv[1], "rt").read())
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    ad = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    a, b = ad[-2], ad[-1]
    step = rows[a + 1:b + 1]
    t0 = rows[a][2]                      # end of the previous Adam
    t1 = step[-1][2]
    print("step: %d kernels, %.3f ms wall (previous adam end -> this adam end)" % (len(step), (t1 - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in step:
        byq[r[3]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r[2] - r[1] for r in rs)
        print("queue %s: %d kernels, busy %.3f ms, first start %.3f ms, last end %.3f ms" % (q, len(rs), busy / 1e6, (rs[0][1] - t0) / 1e6, (rs[-1][2] - t0) / 1e6))
        gaps = sorted(((rs[i + 1][1] - rs[i][2]) for i in range(len(rs) - 1)), reverse=True)
        pos = [g for g in gaps if g > 0]
        print("   gaps: sum %.3f ms, median %.2f us, >5us: %d (sum %.3f ms), top: %s" % (sum(pos) / 1e6, (sorted(pos)[len(pos) // 2] / 1e3 if pos else 0),
              sum(1 for g in pos if g > 5000), sum(g for g in pos if g > 5000) / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))
    # union busy
    ev = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        last = t
        depth += d
    print("concurrency histogram (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
    # windows
    nwin = int((t1 - t0) / 1e6 / win) + 1
    for w in range(nwin):
        lo, hi = t0 + w * win * 1e6, t0 + (w + 1) * win * 1e6
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in step:
            ov = min(r[2], hi) - max(r[1], lo)
            if ov > 0:
                k = (r[3], short(r[0]))
                agg[k][0] += ov
                agg[k][1] += 1
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])[:6]
        print("%5.1f ms | " % (w * win) + " ; ".join("q%s %s x%d %.0fus" % (k[0], k[1][:44], v[1], v[0] / 1e3) for k, v in top))


if __name__ == "__main__":
    main()
