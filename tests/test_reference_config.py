"""SURVEY 8(f) rank 2 -- config fidelity: the reference's OWN config file (configs/LRS23/AV/EffConfInterCTC.py, read from /root/reference, never copied here)
is imported unchanged against this `nnet`, with the `torchvision` stand-in and the synthetic asset tree (tools/make_synthetic_assets.py).  Build container only
(the reference tree does not travel to the GPU box); what it produced is recorded in tests/golden/ref_config_probe.json and re-checked here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/configs/LRS23/AV/EffConfInterCTC.py"
PROBE = os.path.join(ROOT, "tests", "golden", "ref_config_probe.json")

SCRIPT = r'''
import hashlib, json, os, sys, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, ROOT)
import torch
import main as entry                                  # the build's entry point: installs the torchvision stand-in, executes the config file
torch.manual_seed(0)
cfg = entry.load_config(REF_CFG)
import nnet
model = cfg.model
assert type(model).__module__.startswith("avec_amd.nnet"), type(model)
sd = model.state_dict()
out = {"model_class": type(model).__name__, "n_params": sum(p.numel() for p in model.parameters()),
       "state_keys_sha1": hashlib.sha1("\n".join(sd.keys()).encode()).hexdigest(), "n_state_keys": len(sd),
       "decoder": type(model.compiled_decoders["outputs"]).__name__, "tokenizer_loaded": model.compiled_decoders["outputs"].tokenizer is not None,
       "metric": type(model.compiled_metrics["outputs"]).__name__, "loss_weights": {k: (v.val if hasattr(v, "val") else float(v)) for k, v in cfg.loss_weights.items()},
       "precision": str(cfg.precision), "batch_size": cfg.batch_size, "accumulated_steps": cfg.accumulated_steps, "callback_path": cfg.callback_path}
# the LRW front-end transplant really happened: the visual front-end equals the synthetic LRW checkpoint's
ck = torch.load("callbacks/LRW/EffConfCE/checkpoints_epoch_30_step_57247.ckpt", map_location="cpu")["model_state_dict"]
fe = model.encoder.video_encoder.front_end.state_dict()
out["front_end_transplanted"] = all(torch.equal(v, ck["encoder.front_end." + k]) for k, v in fe.items())
# datasets: the training MultiDataset and the two evaluation sets, one collated batch each
tr = cfg.training_dataset
torch.manual_seed(1)
batch = tr.collate_fn([tr[i] for i in range(3)])
out["train_len"], out["train_batch_size"] = len(tr), tr.batch_size
out["train_batch"] = {"video": list(batch["inputs"][0].shape), "video_len": batch["inputs"][1].tolist(), "audio": list(batch["inputs"][2].shape),
                      "audio_len": batch["inputs"][3].tolist(), "label": list(batch["targets"][0].shape), "label_len": batch["targets"][1].tolist()}
ev = cfg.evaluation_dataset
eb = ev[0].collate_fn([ev[0][i] for i in range(2)])
out["eval_sets"], out["eval_video"] = len(ev), list(eb["inputs"][0].shape)
out["aligned"] = all(int(v) == int(a) // 640 + 1 for v, a in zip(batch["inputs"][1], batch["inputs"][3]))
print("PROBE " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="needs the reference tree (build container only)")
def test_reference_av_config_imports_unchanged(tmp_path):
    assets = str(tmp_path / "run")
    os.makedirs(assets)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_assets.py"), assets], check=True, capture_output=True, timeout=900)
    code = "ROOT = %r\nREF_CFG = %r\n" % (ROOT, REF_CFG) + SCRIPT
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")      # importing the config must not drop a __pycache__ into the read-only reference tree
    r = subprocess.run([sys.executable, "-B", "-c", code], cwd=assets, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("PROBE ")][-1][6:])
    assert got["front_end_transplanted"] and got["aligned"] and got["tokenizer_loaded"]
    assert got["model_class"] == "AudioVisualEfficientConformerInterCTC" and got["n_params"] == 61738836
    assert got["train_batch"]["video"][2:] == [88, 88, 1] and got["eval_video"][2:] == [88, 88, 1]
    if os.environ.get("AVEC_WRITE_PROBE") == "1":
        json.dump(got, open(PROBE, "w"), indent=1, sort_keys=True)
    ref = json.load(open(PROBE))
    assert got == ref, {k: (got.get(k), ref.get(k)) for k in set(got) | set(ref) if got.get(k) != ref.get(k)}
