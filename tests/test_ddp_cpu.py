"""CPU, gloo, world_size 2: the host-side data-parallel logic (flat-gradient bucketed all-reduce, SyncBatchNorm statistic exchange,
max-over-ranks timing) -- the same functions the GPU path calls with RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avec_amd import runtime as rt
    # 1. bucketed flat all-reduce == plain sum, for a size that is not a multiple of the bucket
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    rt.all_reduce_flat(g, bucket_bytes=4 * 333)
    ok1 = torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3)
    # 1b. opt-in bf16 payload (AVEC_GRAD_ALLREDUCE_BF16): each rank's values are rounded to bf16 once, the sum is exact for these small integers / within bf16 otherwise
    gb = torch.arange(64, dtype=torch.float32) * (rank + 1)
    rt.all_reduce_flat(gb, bucket_bytes=2 * 21, bf16_payload=True)
    ok1 = ok1 and torch.equal(gb, torch.arange(64, dtype=torch.float32) * 3)
    gr = torch.randn(5000, generator=torch.Generator().manual_seed(3 + rank))
    ref = torch.randn(5000, generator=torch.Generator().manual_seed(3)) + torch.randn(5000, generator=torch.Generator().manual_seed(4))
    rt.all_reduce_flat(gr, bf16_payload=True)
    ok1 = ok1 and gr.dtype == torch.float32 and float((gr - ref).norm() / ref.norm()) < 1e-2
    # 2. SyncBatchNorm statistics: replicated partial sums + local counts -> global mean / var
    C, nrep = 4, 64
    torch.manual_seed(rank)
    x = torch.randn(10 + 5 * rank, C)                    # ranks hold different row counts (padded batches differ)
    stats = torch.zeros(nrep * 2 * C)
    rep = stats.view(nrep, 2 * C)
    for i, row in enumerate(x):                          # scatter partial sums over replicas like the GEMM epilogue does
        rep[i % nrep, :C] += row
        rep[i % nrep, C:] += row * row
    red = rt.sync_bn_stats(stats, nrep, C, x.shape[0])
    allx = torch.cat([torch.randn(10, C, generator=torch.Generator().manual_seed(0)), torch.randn(15, C, generator=torch.Generator().manual_seed(1))])
    n = red[2 * C].item()
    mean, var = red[:C] / n, red[C:2 * C] / n - (red[:C] / n) ** 2
    ok2 = n == 25 and torch.allclose(mean, allx.mean(0), atol=1e-5) and torch.allclose(var, allx.var(0, unbiased=False), atol=1e-5)
    # 3. bench.py timing rule: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok3 = float(t) == float(world)
    # 4. convert_sync_batchnorm switches the engine flag and keeps the module (state_dict keys unchanged)
    import nnet
    bn = nnet.BatchNorm1d(8)
    same = nnet.SyncBatchNorm.convert_sync_batchnorm(bn) is bn and rt.sync_batchnorm()
    out[rank] = bool(ok1 and ok2 and ok3 and same)
    dist.destroy_process_group()


def test_ddp_host_logic_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_length_bucket_sampler_covers_the_epoch_and_cuts_padding():
    """every sample exactly once per epoch (per rank shard: disjoint, same batch count), batches change with the epoch, padded-frame efficiency >= 0.8 on the LRS2-shaped
    duration distribution where uniform batches reach ~0.4"""
    import math
    import torch
    from avec_amd.nnet.samplers import LengthBucketBatchSampler
    g = torch.Generator().manual_seed(0)
    dur = torch.exp(math.log(2.0) + 0.6 * torch.randn(4096, generator=g)).clamp(0.8, 6.2).tolist()
    s = LengthBucketBatchSampler(dur, 32, shuffle=True, drop_last=True, seed=1)
    b0 = list(s)
    assert sorted(i for b in b0 for i in b) == list(range(4096)) and all(len(b) == 32 for b in b0)
    assert s.padded_frame_efficiency() >= 0.8
    uniform = LengthBucketBatchSampler(dur, 32, window=32, shuffle=True, drop_last=True, seed=1)      # window = one batch: the reference's uniform draw
    assert uniform.padded_frame_efficiency() < 0.55
    s.set_epoch(1)
    assert list(s) != b0
    shards = [LengthBucketBatchSampler(dur, 32, shuffle=True, seed=1, rank=r, world_size=3) for r in range(3)]
    lens = [len(list(x)) for x in shards]
    assert len(set(lens)) == 1 and sum(lens) >= 128
    seen = [tuple(b) for x in shards for b in x]
    assert len(set(seen)) >= 128                 # (the padding batches of the last round may repeat)
