"""Probe (GPU box): can two processes on one GPU map each other's device buffers through torch's CUDA-IPC storage sharing?"""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def main(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    buf = torch.full((1024,), float(rank + 1), device="cuda:0")
    share = buf.untyped_storage()._share_cuda_()
    print(rank, "share tuple types", [type(x).__name__ for x in share], flush=True)
    allsh = [None] * world
    dist.all_gather_object(allsh, share)
    peers = []
    for r, sh in enumerate(allsh):
        if r == rank:
            peers.append(buf)
            continue
        st = torch.UntypedStorage._new_shared_cuda(*sh)
        t = torch.empty(0, dtype=torch.float32, device="cuda:0").set_(st, 0, (1024,))
        peers.append(t)
    torch.cuda.synchronize()
    dist.barrier()
    print(rank, "sees peers:", [float(p[0]) for p in peers], [hex(p.data_ptr()) for p in peers], flush=True)
    dist.barrier()
    peers[(rank + 1) % world][rank] = 100.0 + rank           # write into the peer's buffer
    torch.cuda.synchronize()
    dist.barrier()
    print(rank, "own buffer after peer writes:", buf[:4].tolist(), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(main, args=(2, port), nprocs=2)
