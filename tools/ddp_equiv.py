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
    if world > 1 and os.environ.get("AVEC_EARLY_ALLREDUCE", "1") != "0":
        model.arena.arm_early_all_reduce(True)               # as train_step does: two arena ranges are exchanged while backward is still running
    losses, _, _, _ = model.forward_model(inputs, targets, compute_metrics=False)
    losses["loss"].backward()
    early = [(lo, hi) for lo, hi, _ in getattr(model.arena, "_early", [])]
    if world > 1:
        model.arena.all_reduce_grads()
    grad = model.arena.grad / world
    loss = losses["loss"].detach().clone()
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
