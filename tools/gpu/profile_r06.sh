# round-6 evidence on the current tree: default bench line, rocprofv3 kernel stats of the graph-replayed step, PMC traffic passes, SQ counters, step stamps, block trace, variants
#   bash tools/gpu/profile_r06.sh [all|quick]
mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
MODE=${1:-all}
# kernel trace + stats of the graph-replayed step (1 eager warm-up step + 4 replays = 5 steps traced; model set-up copies included)
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_bench.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_final.csv \;
# PMC traffic (separate passes, kernel trace only)
( cd /tmp && rm -rf /tmp/pmc_f && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_f.log 2>&1 )
( cd /tmp && rm -rf /tmp/pmc_w && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > $O/pmc_w.log 2>&1 )
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $O/r06_pmc_traffic.json > $O/pmc_top.txt 2>&1
cp $O/r06_pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null      # (bench.py reads the newest profiles/rNN_pmc_traffic.json for roofline.traffic)
( cd /tmp && rm -rf /tmp/pmc_sq && timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_sq -o s -- python $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-timing > /tmp/pmc_sq.log 2>&1 )
python tools/pmc_sq.py /tmp/pmc_sq 40 > $O/r06_pmc_sq.txt 2>&1 || tail -5 /tmp/pmc_sq.log > $O/r06_pmc_sq.txt
cp $O/r06_kernel_stats_final.csv profiles/r06_kernel_stats_final.csv 2>/dev/null      # (bench.py times the roofline rows from the newest profiles/rNN_kernel_stats_final.csv)
python tools/roofline_table.py $O/r06_kernel_stats_final.csv $O/r06_pmc_traffic.json 5 > $O/r06_roofline_table.md 2>&1
AVEC_STAMPS=1 timeout 600 python tools/step_stamps.py 2>&1 | grep -v amdgpu > $O/r06_step_stamps.txt
python tools/block_trace.py 32 100 256 2>&1 | grep -v amdgpu > $O/r06_block_trace.txt
# the default bench line last (it reads the PMC summary copied above)
python bench.py 2>$O/bench_default.err | grep "^{" > $O/r06_bench_line.json
python -c "import json; d=json.loads(open('$O/r06_bench_line.json').read().strip().splitlines()[-1]); print('STEP', d['ms_per_step'], d['value'], d['config'].get('eager_two_stream_ms_per_step'), d['roofline']['frac'] if d.get('roofline') else None)"
if [ "$MODE" = "all" ]; then
: > $O/r06_variants.txt
for v in lrs2_main lrs2_pre av15s ao vo lrw; do
timeout 600 python tools/bench_variants.py --only $v --steps 12 2>&1 | grep "^{" >> $O/r06_variants.txt
done
timeout 900 python tools/bench_variants.py --only lrs2_main --steps 12 --graphs 2>&1 | grep "^{" >> $O/r06_variants.txt
timeout 600 python bench.py --fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" >> $O/r06_variants.txt
for b in 1 8 16 48 64; do
python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b', d['ms_per_step'], d['value'], d['config']['model_mfma_util'])" >> $O/r06_batches.txt || echo "B=$b FAILED" >> $O/r06_batches.txt
done
fi
head -30 $O/r06_kernel_stats_final.csv | cut -c1-150
cat $O/r06_step_stamps.txt
