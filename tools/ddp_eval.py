"""Two ranks (one GPU, gloo): Model.evaluate with recompute_metrics sums losses over the ranks and gathers the decoded hypotheses, so that every rank reports the
loss / word error rate of the WHOLE evaluation set (nnet/model.py:899-931).  Checked against the single-rank evaluations of the two shards.  Launched by
tests/test_gpu_ddp.py through torch.distributed.run."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="gloo", init_method="env://")
    import avec_amd, nnet
    avec_amd.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = nnet.AudioVisualEfficientConformerInterCTC(vocab_size=256, v_interctc_blocks=[3, 6], a_interctc_blocks=[8, 11], f_interctc_blocks=[2])
    model.compile(losses=nnet.CTCLoss(zero_infinity=True, assert_shorter=False), decoders={"outputs": nnet.CTCGreedySearchDecoder()},
                  metrics={"outputs": nnet.WordErrorRate()})
    model = model.to(dev)
    collate = nnet.CollateFn(inputs_params=[{"axis": 0, "padding": True}, {"axis": 3}, {"axis": 1, "padding": True}, {"axis": 4}],
                             targets_params=({"axis": 2, "padding": True}, {"axis": 5}))
    ds = nnet.datasets.LRS(batch_size=2, collate_fn=collate, version="LRS2", mode="test", num_synthetic=8, seed=3, video_max_length=60)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, sampler=sampler, collate_fn=collate)
    local = model.evaluate(loader, recompute_metrics=True)                 # this rank's shard only (not distributed yet)
    model.distribute_strategy(rank)
    glob = model.evaluate(loader, recompute_metrics=True)                  # summed / gathered over the ranks
    locs = [None] * world
    dist.all_gather_object(locs, local)
    globs = [None] * world
    dist.all_gather_object(globs, glob)
    if rank == 0:
        torch.save({"local": locs, "global": globs}, args.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
