# shader clock and socket power while the captured step replays (rocm-smi samples every ~0.3 s beside bench.py)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
L=gpurun_out/r5_power.log
: > $L
rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" >> $L
( python bench.py --steps 4000 --warmup 20 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])" >> $L ) &
BP=$!
sleep 50
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' ' >> $L; echo >> $L
  sleep 0.3
done
wait $BP
echo "--- idle" >> $L
sleep 2
rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | tr '\n' ' ' >> $L; echo >> $L
cat $L
