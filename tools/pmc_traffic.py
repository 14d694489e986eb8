"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate runs) into per-launch HBM traffic per kernel family.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python /root/repo/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python /root/repo/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing
    python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w profiles/r01_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 128-byte
requests as 64 B, so the fetch figure is doubled.  WRITE_SIZE is uncalibrated there and is reported as is."""
import csv, glob, json, os, sys, collections


def load(d, counter):
    f = [p for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)]
    assert f, "no counter_collection.csv under " + d
    acc = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != counter:
            continue
        a = acc[row["Kernel_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k in fetch:
        n, kb = fetch[k]
        wn, wkb = write.get(k, [0, 0.0])
        res[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * kb * 1024 / n, "write_bytes_per_launch": (wkb * 1024 / wn) if wn else None}
    top = sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch"] * kv[1]["launches"]))[:40]
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), 2 eager steps of bench.py at B=32 bf16; fetch doubled (gfx950 correction)",
               "kernels": dict(top)}, open(out, "w"), indent=1)
    for k, v in top[:12]:
        print("%-70s n=%4d fetch %8.1f MB  write %s MB" % (k[:70], v["launches"], v["fetch_bytes_per_launch"] / 1e6,
                                                          "%8.1f" % (v["write_bytes_per_launch"] / 1e6) if v["write_bytes_per_launch"] else "n/a"))


if __name__ == "__main__":
    main()
