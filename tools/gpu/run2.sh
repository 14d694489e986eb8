set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "grouped or layernorm_rows" 2>&1 | tail -15 > gpurun_out/r2_tests2.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/r2_tests2.log
for cfg in "AVEC_DEFER_WGRAD=0" "AVEC_DEFER_WGRAD=1" "AVEC_DEFER_WGRAD=1 AVEC_TNG_TILE=64" "AVEC_DEFER_WGRAD=1 AVEC_TNG_TILE=128" "AVEC_DEFER_WGRAD=1 AVEC_TNG_WGS=768" "AVEC_DEFER_WGRAD=1 AVEC_TNG_WGS=3072" "AVEC_DEFER_WGRAD=1 AVEC_DEFER_TN_MAX=12"; do
  echo "== $cfg" >> gpurun_out/r2_bench2.log
  env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> gpurun_out/r2_bench2.log
done
