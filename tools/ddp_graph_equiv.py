"""Two data-parallel ranks on one GPU (gloo): three optimisation steps of the AO model, eagerly (Model.train_step) or from a captured hipGraph
(Model.make_graphed_train_step: forward + backward with peer-write SyncBatchNorm exchanges inside the graph) -- launched by tests/test_gpu_ddp.py."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--mode", default="eager")
    ap.add_argument("--backend", default="gloo", help="gloo: both ranks on cuda:0; nccl (= RCCL): one GPU per rank")
    ap.add_argument("--replays", type=int, default=2, help="graph mode: replays after the warm-up step")
    args = ap.parse_args()
    plain = args.backend == "none"               # no process group at all: the single-process reference of the one-rank RCCL runs
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", 0 if args.backend in ("gloo", "none") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if plain:
        pass
    elif args.backend == "nccl":
        torch.distributed.init_process_group(backend="nccl", init_method="env://", device_id=dev)
    else:
        torch.distributed.init_process_group(backend="gloo", init_method="env://")
    import avec_amd
    import nnet
    from avec_amd import peer
    avec_amd.set_compute_dtype("bf16")
    avec_amd.manual_seed(7)
    torch.manual_seed(0)
    model = nnet.AudioEfficientConformerInterCTC(vocab_size=256, att_type="patch", interctc_blocks=[3, 6])
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "drop_rate"):
            m.drop_rate = 0.0
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False))
    model = model.to(dev).train()
    model.encoder.spec_augment.eval()
    if not plain:
        model.distribute_strategy(rank)
    master0 = model.arena.master.clone()
    g = torch.Generator().manual_seed(11 + rank)
    audio, alen = 0.1 * torch.randn(2, 32000, generator=g), torch.tensor([32000, 25000])
    labels, llen = torch.randint(1, 256, (2, 6), generator=g), torch.tensor([6, 4])
    inputs, targets = [audio.to(dev), alen.to(dev)], (labels.to(dev), llen.to(dev))
    graphed = args.mode == "graph"
    if graphed:
        step = model.make_graphed_train_step(inputs, targets, precision=torch.bfloat16, warmup=1)      # the warm-up pass is a real optimisation step
        for _ in range(args.replays):
            losses = step()
    else:
        for _ in range(3):
            losses, _, _ = model.train_step(inputs, targets, precision=torch.bfloat16)
    torch.cuda.synchronize()
    loss = losses["loss"].detach().float().clone()
    loss = loss if args.backend == "nccl" else loss.cpu()
    if not plain:
        torch.distributed.all_reduce(loss)
    if peer.active() is not None:
        peer.active().check()
    if rank == 0:
        from avec_amd import runtime as rt
        torch.save({"in_graph": bool(graphed and getattr(step, "collectives_in_graph", False)), "sync_bn": bool(rt.sync_batchnorm()), "branch_group": rt._BRANCH.get("group") is not None,
                    "peer": peer.active() is not None, "graphed": graphed, "master": model.arena.master.cpu(), "master0": master0.cpu(), "exp_avg": model.optimizer._flat["exp_avg"].cpu(), "step": int(model.model_step), "loss": float(loss) / world}, args.out)
    if not plain:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
