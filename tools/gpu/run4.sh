set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "ffn" 2>&1 | tail -25 > gpurun_out/r2_tests4.log
timeout 600 python tools/bench_ffn.py > gpurun_out/r2_bench_ffn.log 2>&1
for cfg in "AVEC_NO_FFN_FUSED=1" "AVEC_X=1"; do
  echo "== $cfg" >> gpurun_out/r2_bench4.log
  env $cfg timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> gpurun_out/r2_bench4.log
done
