"""N processes (default 2; `python tools/peer_stress.py 8` = avec_amd.peer.MAX_WORLD) on cuda:0: the SyncBatchNorm peer exchange (avec_amd/peer.py) against gloo all_reduce --
launched by tests/test_gpu_ddp.py."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def main(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from avec_amd import peer
    px = peer.setup(dev)
    assert px is not None, "peer exchange did not come up"
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    sizes = [129, 513, 2049, 1, 361, 721, 1025, 7]
    # eager: 40 rounds over 8 sites, unequal pacing between the ranks (rank 1 does extra work every few rounds)
    for it in range(40):
        for si, n in enumerate(sizes):
            v = torch.randn(n, generator=g).to(dev)
            got = px.all_reduce_sum(v, ("stress", si))
            ref = v.cpu()
            dist.all_reduce(ref)
            # (two ranks: a + b is the same in either order; more ranks: the slots are summed in rank order here and pairwise by gloo -- fp32 rounding apart)
            assert torch.equal(got.cpu(), ref) or (world > 2 and torch.allclose(got.cpu(), ref, rtol=1e-5, atol=1e-5)), (it, si)
        if rank == 1 and it % 7 == 0:
            torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
    # graph: the same sites replayed 10 times with fresh inputs copied into static buffers
    static = [torch.zeros(n, device=dev) for n in sizes]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        outs = [px.all_reduce_sum(v, ("stress", si)) for si, v in enumerate(static)]       # warm-up (allocations)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = [px.all_reduce_sum(v, ("stress", si)) for si, v in enumerate(static)]
    torch.cuda.synchronize()
    dist.barrier()
    for it in range(10):
        refs = []
        for v in static:
            x = torch.randn(v.numel(), generator=g)
            v.copy_(x)
            r = x.clone()
            dist.all_reduce(r)
            refs.append(r)
        graph.replay()
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            assert torch.equal(o.cpu(), r) or (world > 2 and torch.allclose(o.cpu(), r, rtol=1e-5, atol=1e-5)), it
    px.check()
    dist.barrier()
    print("PEER STRESS OK rank %d" % rank, flush=True)
    peer.reset()
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(main, args=(world, port), nprocs=world)
