"""Uninitialised / out-of-bounds read detector (GPU): the caching allocator's free blocks are filled with a poison value (NaN, 1e30 or 0) before a forward + backward pass of the
audio-visual model; loss and gradients must not depend on the poison.  One process; AVEC_DIST_SINGLE=1 sends the pass through the data-parallel code paths (SyncBatchNorm
exchange kernels, early all-reduce) with a one-rank group.   python tools/poison_check.py [--batch B] [--dtype f32|bf16] [--dist]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def poison(dev, val):
    xs = []
    for k in range(6, 27):
        for mult in (1, 1, 1, 3):
            n = (1 << k) * mult // (2 if mult == 3 else 1)
            xs.append(torch.full((n,), val, dtype=torch.float32, device=dev))
    for n in (77 * 400, 77 * 514, 80 * 77, 39 * 7200, 23040, 720, 1025, 2049, 180 * 9, 361, 12160):
        for _ in range(6):
            xs.append(torch.full((n,), val, dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    del xs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--noise", action="store_true", help="keep dropout and SpecAugment on (same masks in every pass: the RNG step is rewound)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if args.dist:
        os.environ["AVEC_DIST_SINGLE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
        torch.distributed.init_process_group(backend="gloo", rank=0, world_size=1)
    import avec_amd, nnet
    avec_amd.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    for m in (model.modules() if not args.noise else []):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_rate"):
            m.drop_rate = 0.0
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    if not args.noise:
        model.encoder.audio_encoder.spec_augment.eval()
    from avec_amd import runtime as rt
    if args.dist:
        model.distribute_strategy(0)
    B = args.batch
    g = torch.Generator().manual_seed(5)
    video, audio = torch.randn(B, 20, 88, 88, 1, generator=g), 0.1 * torch.randn(B, 12160, generator=g)
    vlen, alen = torch.tensor(([20, 17, 20, 11] * B)[:B]), torch.tensor(([12160, 10000, 12160, 7000] * B)[:B])
    labels, llen = torch.randint(1, 256, (B, 4), generator=g), torch.tensor(([4, 3, 4, 2] * B)[:B])
    inputs = [t.to(dev) for t in (video, vlen, audio, alen)]
    targets = (labels.to(dev), llen.to(dev))
    ref = None
    for name, val in (("zero", 0.0), ("nan", float("nan")), ("1e30", 1e30), ("-3e4", -3e4), ("nan again", float("nan"))):
        model.arena.grad.zero_()
        rt.rng_state(dev)[1] = 0
        poison(dev, val)
        if args.dist:
            model.arena.arm_early_all_reduce(True)
        losses, _, _, _ = model.forward_model(inputs, targets, compute_metrics=False)
        poison(dev, val)
        losses["loss"].backward()
        if args.dist:
            model.arena.all_reduce_grads()
        torch.cuda.synchronize()
        loss, grad = float(losses["loss"]), model.arena.grad.clone()
        line = "poison %-9s loss %.9g  |grad| %.9g  nan in grad %d" % (name, loss, float(grad.double().norm()), int(torch.isnan(grad).sum()))
        if ref is None:
            ref = (loss, grad)
        else:
            d = (grad - ref[1]).abs()
            line += "   vs zero: dloss %.3e  max|dgrad| %.3e (max|grad| %.3e)" % (abs(loss - ref[0]), float(d.max()), float(ref[1].abs().max()))
            bad = torch.isnan(grad) | (d > 1e-2 * ref[1].abs().max())
            if bool(bad.any()):
                off_of = {id(p_): o for p_, o in zip(model.arena.params, model.arena.offsets)}
                names = sorted((off_of[id(p_)], p_.numel(), k) for k, p_ in model.named_parameters())
                hit = [k for o, n, k in names if bool(bad[o:o + n].any())]
                line += "\n    parameters touched (%d): %s" % (len(hit), hit[:12])
        print(line, flush=True)
        del losses


if __name__ == "__main__":
    main()
