# round 5 baseline: step time, kernel stats of the replayed step, step stamps
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>$O/bench.err | grep "^{" > $O/bench_line.json
python -c "import json; d=json.loads(open('$O/bench_line.json').read().strip().splitlines()[-1]); print('STEP', d['ms_per_step'], d['value'])"
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_bench.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/r05_kernel_stats.csv \;
head -60 $O/r05_kernel_stats.csv | cut -c1-160
AVEC_STAMPS=1 timeout 600 python tools/step_stamps.py 2>&1 | grep -v amdgpu > $O/r05_step_stamps.txt
cat $O/r05_step_stamps.txt
