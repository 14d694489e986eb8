mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r3_rows.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_rows.json').read().strip().splitlines()[-1])
print(d['ms_per_step'])
for x in d['roofline']['rows']: print(x['kernel'], x['launches_per_step'], x['avg_us'], x['tflops'])
for k,v in d['roofline']['families'].items(): print(k, v)
PY
