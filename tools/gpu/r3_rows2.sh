mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for v in $1; do
AVEC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/libavec_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3_rows_$v.json
python - <<PY
import json
d=json.loads(open('gpurun_out/r3_rows_$v.json').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'])
for x in d['roofline']['rows']:
    if 'wgrad' in x['kernel'] or 'shift' in x['kernel'] or 'c64' in x['kernel']: print('  ', x['kernel'], x['launches_per_step'], x['avg_us'], x['tflops'])
PY
done
