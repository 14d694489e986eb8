import os, sys, collections, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avec_amd, nnet
from bench import synthetic_batch
avec_amd.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
model = model.to(dev).train()
inputs, targets = synthetic_batch(2, dev, 0)
for _ in range(2):
    model.train_step(inputs, targets, precision=torch.bfloat16)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.train_step(inputs, targets, precision=torch.bfloat16)
    torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/trace.json")
tr = json.load(open("/tmp/trace.json"))["traceEvents"]
mem = [e for e in tr if e.get("cat") in ("gpu_memcpy", "gpu_memset")]
print("gpu memcpy/memset events:", len(mem))
print(collections.Counter((e["name"], e.get("args", {}).get("bytes")) for e in mem).most_common(20))
rt = {e["args"]["correlation"]: e for e in tr if e.get("cat") in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {})}
ops_ = sorted([e for e in tr if e.get("cat") in ("cpu_op", "python_function", "user_annotation") and "dur" in e], key=lambda e: e["ts"])
cnt = collections.Counter()
for m in mem:
    r = rt.get(m["args"].get("correlation"))
    if r is None:
        cnt[("?", m["name"])] += 1
        continue
    enc = [o for o in ops_ if o["ts"] <= r["ts"] and o["ts"] + o["dur"] >= r["ts"] + r.get("dur", 0)]
    enc = sorted(enc, key=lambda o: o["dur"])[:4]
    cnt[(r["name"],) + tuple(o["name"][-70:] for o in enc)] += 1
for k, c in cnt.most_common(25):
    print(c, k)
