cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06; export PYTHONPATH=$PWD OMP_NUM_THREADS=4
P=${1:-8}; IT=${2:-150}; R=${3:-3}
for r in $(seq 1 $R); do
  pids=""
  for p in $(seq 1 $P); do timeout 900 python tools/stress_audio_front.py $IT $r.$p 2>/tmp/se_$r.$p.txt & pids="$pids $!"; done
  for q in $pids; do wait $q || echo "pid $q rc=$?"; done
done
