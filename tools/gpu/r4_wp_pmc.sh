# round 4: HBM fetch bytes of the pair weight-gradient launch for the two share orders (range-major default, kind-major AVEC_WP_RANGES=1)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
L=$R/gpurun_out/r4_wp_pmc.log
: > $L
for cfg in "AVEC_X=0" "AVEC_WP_RANGES=1" "AVEC_WP_RANGES=4"; do
  rm -rf /tmp/wp_pmc
  env $cfg PYTHONPATH=$R timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/wp_pmc -o f -- python $R/tools/bench_wgrad_wide.py > /tmp/wp_pmc.log 2>&1
  python - "$cfg" >> $L <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/wp_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "pairs" in r["Kernel_Name"]:
        acc[(r["Grid_Size"] if "Grid_Size" in r else "")].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(sys.argv[1], "launches", len(v), "FETCH_SIZE per launch: min %.3e max %.3e (x64 B x2 on gfx950 = %.2f GB max)" % (min(v), max(v), max(v) * 64 * 2 / 1e9))
PY
done
cat $L
