"""Data-parallel equivalence check (GPU): N ranks x (B/N) utterances with SyncBatchNorm statistics + flat gradient all-reduce must
reproduce the single-process gradients of the full batch.  Launched by tests/test_gpu_ddp.py (torch.distributed.run, gloo on one GPU;
on a multi-GPU node use --backend nccl without --share-gpu)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--batch", type=int, default=4, help="global batch (a multiple of 4 and of the world size): the four-utterance pattern repeated")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        torch.distributed.init_process_group(backend=args.backend, init_method="env://", **({"device_id": dev} if args.backend == "nccl" else {}))
    import avec_amd, nnet
    avec_amd.set_compute_dtype("f32")
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_rate"):
            m.drop_rate = 0.0
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    model.encoder.audio_encoder.spec_augment.eval()
    if world > 1:
        model.distribute_strategy(rank)
    B = args.batch
    assert B % 4 == 0 and B % world == 0
    g = torch.Generator().manual_seed(5)
    video, audio = torch.randn(B, 20, 88, 88, 1, generator=g), 0.1 * torch.randn(B, 12160, generator=g)
    vlen, alen = torch.tensor([20, 17, 20, 11] * (B // 4)), torch.tensor([12160, 10000, 12160, 7000] * (B // 4))
    labels, llen = torch.randint(1, 256, (B, 4), generator=g), torch.tensor([4, 3, 4, 2] * (B // 4))
    sl = slice(rank * B // world, (rank + 1) * B // world)
    inputs = [t[sl].to(dev) for t in (video, vlen, audio, alen)]
    targets = (labels[sl].to(dev), llen[sl].to(dev))
    log = []
    if world > 1 and os.environ.get("AVEC_DDP_DEBUG"):
        # keep (site, local vector, exchanged sum) of every peer exchange without touching the host-side timing; checked against torch.distributed after the pass
        from avec_amd import peer as _peer
        px_ = _peer.active()
        if px_ is not None:
            f0, f1 = px_.all_reduce_sum, px_.all_reduce_sum_fused
            def all_reduce_sum(vec, key):
                out = f0(vec, key); log.append((key, vec.clone(), out)); return out
            def all_reduce_sum_fused(src, nrep, n_in, tail, key, dgamma=None, dbeta=None, C=0):
                loc = src[:nrep * n_in].view(nrep, n_in).sum(0)
                if tail is not None:
                    loc = torch.cat([loc, torch.full((1,), float(tail), device=loc.device)])
                out = f1(src, nrep, n_in, tail, key, dgamma, dbeta, C); log.append((key, loc, out)); return out
            px_.all_reduce_sum, px_.all_reduce_sum_fused = all_reduce_sum, all_reduce_sum_fused
    taps = []
    if world > 1 and os.environ.get("AVEC_DDP_DEBUG_DIR"):
        # the audio front of the pass: REFERENCES to tensors the pass keeps alive anyway (no extra kernels, no clones: the timing of the pass is the product's); written out after the pass
        from avec_amd import ops as _ops
        a0 = _ops.AudioStemFn.apply
        def stem_apply(mel, *a, **k):
            out = a0(mel, *a, **k); taps.append(("stem_in", mel)); taps.append(("stem_out", out)); return out
        _ops.AudioStemFn.apply = stem_apply
    if world > 1 and os.environ.get("AVEC_EARLY_ALLREDUCE", "1") != "0":
        model.arena.arm_early_all_reduce(True)               # as train_step does: two arena ranges are exchanged while backward is still running
    losses, _, _, _ = model.forward_model(inputs, targets, compute_metrics=False)
    losses["loss"].backward()
    early = [(lo, hi) for lo, hi, _ in getattr(model.arena, "_early", [])]
    if log:
        torch.cuda.synchronize()
        name_of_ = {id(m_): n_ for n_, m_ in model.named_modules()}
        if os.environ.get("AVEC_DDP_DEBUG_DIR"):
            os.makedirs(os.environ["AVEC_DDP_DEBUG_DIR"], exist_ok=True)
            torch.save([((name_of_.get(k[0], "?"), k[1]) if isinstance(k, tuple) else k, l.cpu(), o.cpu()) for k, l, o in log], os.path.join(os.environ["AVEC_DDP_DEBUG_DIR"], "rank%d.pt" % rank))
            if taps:
                sv = taps[1][1].grad_fn.saved                     # AudioStemFn ctx: (mel, y, st, cp, count, ...)
                taps += [("stem_y", sv[1]), ("stem_stats", sv[2].stats[:2 * sv[2].C]), ("stem_ss", sv[2].ss)] + ([("stem_red", sv[2].red)] if sv[2].red is not None else [])
                torch.save([(n_, t_.detach().float().cpu()) for n_, t_ in taps], os.path.join(os.environ["AVEC_DDP_DEBUG_DIR"], "taps%d.pt" % rank))
        for idx, (key, loc, out) in enumerate(log):
            ref = loc.cpu()
            torch.distributed.all_reduce(ref)
            got = out.cpu()
            scale = loc.abs().cpu()
            torch.distributed.all_reduce(scale)
            bad = ~((got - ref).abs() <= 1e-5 * scale + 1e-30)          # (the two sums add in different orders: tolerance relative to the sum of magnitudes)
            if bool(bad.any()):
                i = int(bad.nonzero()[0])
                kn = (name_of_.get(key[0], "?"), key[1]) if isinstance(key, tuple) else key
                print("rank %d exchange #%d site %s n=%d: %d elements differ, first [%d] got %.9g ref %.9g local %.9g" % (rank, idx, kn, ref.numel(), int(bad.sum()), i, float(got[i]), float(ref[i]), float(loc[i])), flush=True)
    if world > 1:
        model.arena.all_reduce_grads()
    grad = model.arena.grad / world
    loss = losses["loss"].detach().clone()
    if os.environ.get("AVEC_DDP_DEBUG"):
        print("rank %d local loss %.9g |grad| %.9g" % (rank, float(loss), float(model.arena.grad.double().norm())), flush=True)
    if world > 1:
        torch.distributed.all_reduce(loss)
        loss /= world
    site_orders = None
    if world > 1:
        from avec_amd import peer as _peer
        px = _peer.active()
        name_of = {id(m_): n_ for n_, m_ in model.named_modules()}
        mine = [] if px is None else [(name_of.get(k[0], str(k[0])) if isinstance(k, tuple) else str(k), k[1] if isinstance(k, tuple) else "", v) for k, v in px.sites.items()]
        site_orders = [None] * world
        torch.distributed.all_gather_object(site_orders, mine)
        if px is not None:
            px.check()
    if rank == 0:
        bn = model.encoder.video_encoder.front_end[3].blocks[0].layers[1]
        off_of = {id(p_): o for p_, o in zip(model.arena.params, model.arena.offsets)}
        names = {k: (off_of[id(p_)], p_.numel()) for k, p_ in model.named_parameters()}
        from avec_amd import peer
        torch.save({"site_orders": site_orders, "peer": peer.active() is not None, "early": early, "numel": model.arena.numel, "grad": grad.cpu(), "names": names, "loss": loss.cpu(), "running_mean": bn.running_mean.cpu(), "running_var": bn.running_var.cpu()}, args.out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
