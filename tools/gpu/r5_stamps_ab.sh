mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
for cfg in "$@"; do
echo "== $cfg"
env $cfg AVEC_STAMPS=1 timeout 600 python tools/step_stamps.py 2>&1 | grep -v amdgpu | tail -22
env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['step_path'])"
done > gpurun_out/r5_stamps_ab.log 2>&1
cat gpurun_out/r5_stamps_ab.log
