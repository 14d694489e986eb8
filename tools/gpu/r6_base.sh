# round 6: bench line + ordered kernel trace of the graph-replayed step (last step, per queue) + step stamps on the current tree
#   bash tools/gpu/r6_base.sh [tag]
TAG=${1:-base}
mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>$O/${TAG}_bench.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP $TAG', d['ms_per_step'], d['value'], d['config']['loss'])" | tee $O/${TAG}_step.txt
( cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/${TAG}_prof.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
python tools/trace_tail.py /tmp/prof $O/${TAG}_trace_tail.json.gz 2>&1 | tail -3
AVEC_STAMPS=1 timeout 600 python tools/step_stamps.py 2>&1 | grep -v amdgpu > $O/${TAG}_step_stamps.txt
cat $O/${TAG}_step_stamps.txt
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>>$O/${TAG}_bench.err | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP $TAG', d['ms_per_step'], d['value'], d['config']['loss'])" | tee -a $O/${TAG}_step.txt
