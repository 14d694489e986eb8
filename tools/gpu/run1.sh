set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r2_tests1.log
python - <<'PY' > gpurun_out/r2_bf16_errs.log 2>&1
import torch, json
from tests import bf16_grad_probe as probe
from oracle import avec_oracle as O
model, sd0 = probe.build_model()
video, vlen, audio, alen, labels, llen = probe.av_inputs(2)
sd = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_(True)
out = O.av_forward(sd, video.double(), vlen, audio.double(), alen, train=True, stats_out={})
O.total_loss(out, labels, llen, O.AV_LOSS_WEIGHTS)["loss"].backward()
g64 = {k: v.grad.clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}
errs, losses, fin = probe.grad_errors(model, g64, "bf16")
e = sorted((v,k) for k,v in errs.items() if not k.endswith(("key_layer.bias", "pos_layer.bias", "conv_module.layers.3.bias", "layers.0.0.bias")))
print("n", len(e), "median", e[len(e)//2], "p90", e[int(len(e)*0.9)], "max", e[-10:])
PY
python bench.py > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1
ls -la /tmp/prof /tmp/prof/* | head; 
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); head -3 $f; wc -l $f
python - "$f" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
# keep the last 40% of dispatches by start time (last full graph replay(s))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import json
keep = rows[int(len(rows)*0.55):]
out = [[r["Kernel_Name"][:120], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Stream_Id",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Grid_Size_Y",""), r.get("LDS_Block_Size","")] for r in keep]
gzip.open(sys.argv[1].rsplit("/",1)[0] + "/trace_tail.json.gz", "wt").write(json.dumps(out))
PY
cp $(dirname $f)/trace_tail.json.gz $GRAFT_REPO_ROOT/gpurun_out/r2_trace_tail.json.gz
tail -5 /tmp/prof.log
