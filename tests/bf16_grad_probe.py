"""Helper of tests/test_gpu_round2.py (not collected): full AV model, bf16 mode, B=2 -- relative-L2 error of every parameter gradient against
an fp64 oracle gradient file.  Run in a subprocess so that kernel-selection environment variables (read once per process by the C library:
AVEC_NT_RB, AVEC_TN_WGS, ...) can force the variants the B=32 bench shape selects.

    python -m tests.bf16_grad_probe <oracle_grads.pt> <out.json>
"""
import json
import sys

import torch


def av_inputs(B=2):
    torch.manual_seed(1)
    video = torch.randn(B, 100, 88, 88, 1)
    audio = 0.1 * torch.randn(B, 63840)
    vlen, alen = torch.tensor([100, 63][:B]), torch.tensor([63840, 40000][:B])
    labels = torch.randint(1, 256, (B, 20))
    llen = torch.tensor([20, 13][:B])
    return video, vlen, audio, alen, labels, llen


def build_model():
    import nnet
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    for x in model.modules():
        if isinstance(x, torch.nn.Dropout):
            x.p = 0.0
        if hasattr(x, "drop_rate"):
            x.drop_rate = 0.0
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(torch.device("cuda:0")).train()
    model.encoder.audio_encoder.spec_augment.eval()
    return model, sd0


def grad_errors(model, g64, dtype="bf16"):
    """{name: relative L2 error of the HIP gradient vs g64[name]} plus the losses of the same pass"""
    import avec_amd
    dev = torch.device("cuda:0")
    video, vlen, audio, alen, labels, llen = [t.to(dev) for t in av_inputs(2)]
    avec_amd.set_compute_dtype(dtype)
    avec_amd.manual_seed(1234)
    model.arena.zero_grad()
    losses, _, _, _ = model.forward_model([video, vlen, audio, alen], (labels, llen), compute_metrics=False)
    losses["loss"].backward()
    torch.cuda.synchronize()
    errs = {}
    for k, p in model.named_parameters():
        ref = g64[k].double()
        errs[k] = ((p.grad.detach().cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30)).item()
    finite = bool(torch.isfinite(model.arena.grad).all())
    avec_amd.set_compute_dtype("f32")
    return errs, {k: float(v) for k, v in losses.items()}, finite


def main():
    g64 = torch.load(sys.argv[1])
    model, _ = build_model()
    errs, losses, finite = grad_errors(model, g64)
    json.dump({"errs": errs, "losses": losses, "finite": finite}, open(sys.argv[2], "w"))


if __name__ == "__main__":
    main()
