"""SyncBatchNorm statistic exchange by direct peer writes over xGMI (csrc/peer.hip) instead of 90 latency-bound RCCL all-reduces per step.

One uncached exchange buffer per rank, mapped by every rank of the node through HIP IPC; an exchange is ONE single-workgroup kernel on the calling
stream (no host synchronisation, graph-capturable).  setup() verifies the mechanism against torch.distributed.all_reduce on this very node and the
caller falls back to RCCL collectives when anything fails (IPC refused, more than 8 ranks, ranks on several nodes, a wrong sum)."""
import ctypes
import os

import torch

from .lib import lib

MAX_WORLD = 8
MAXN = 2056          # granules per slot: vectors of up to 2*1024 + 1 floats (BatchNorm widths <= 1024)
SITES = 256          # distinct exchange sites (a BatchNorm layer uses two: forward statistics, backward sums)
TIMEOUT_MS = int(float(os.environ.get("AVEC_PEER_TIMEOUT_S", "600")) * 1000)      # device-side spin limit of one exchange


class PeerExchange:
    """Construction is split so that EVERY rank runs the same sequence of collectives whatever fails locally (setup() below):
    allocate() -> [all_gather_object of the handles] -> map(handles) -> [agreement all_reduce] -> ready."""

    def __init__(self, rank, world, device, group=None):
        from . import runtime as rt
        assert rt.dist_min_world() <= world <= MAX_WORLD, "peer exchange serves 2..%d ranks of one node" % MAX_WORLD
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.site_granules = 2 * world * MAXN                     # two parity pages of world slots
        self._own, self._opened, self.bases, self.sites = None, [], [], {}
        self.epochs = self.err = None
        # a late rank (dataloader stall, checkpoint I/O, an uneven last batch) must not poison the step: wait as long as a collective would
        # (RCCL's default is minutes); the Adam launch is additionally guarded by `err` (optimizers.Adam._launch)
        self.timeout_ms = TIMEOUT_MS

    def allocate(self):
        """-> the 64-byte IPC handle of this rank's exchange buffer"""
        own, handle = ctypes.c_void_p(), ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            lib.peer_buffer_alloc(ctypes.byref(own), SITES * self.site_granules * 8, handle)
        self._own = own.value
        return handle.raw

    def map(self, handles):
        with torch.cuda.device(self.device):
            for r, h in enumerate(handles):
                if r == self.rank:
                    self.bases.append(self._own)
                    continue
                p = ctypes.c_void_p()
                lib.peer_buffer_open(h, ctypes.byref(p))
                self.bases.append(p.value)
                self._opened.append(p.value)
        self.epochs = torch.zeros(SITES, dtype=torch.int32, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)

    def all_reduce_sum(self, vec, key):
        """sum of `vec` (fp32, contiguous, <= MAXN elements) over the ranks, identical bits on every rank; `key` names the exchange site (every rank must
        visit the sites in the same order)"""
        site = self.sites.get(key)
        if site is None:
            site = self.sites[key] = len(self.sites)
            assert site < SITES, "more than %d SyncBatchNorm exchange sites" % SITES
        n = vec.numel()
        assert vec.dtype == torch.float32 and vec.is_contiguous() and 0 < n <= MAXN
        out = torch.empty_like(vec)
        pages = (ctypes.c_void_p * self.world)(*[b + site * self.site_granules * 8 for b in self.bases])
        lib.peer_exchange_sum(vec.data_ptr(), out.data_ptr(), n, pages, self.world * MAXN, self.rank, self.world,
                              self.epochs.data_ptr() + 4 * site, self.err.data_ptr(), self.timeout_ms, torch.cuda.current_stream().cuda_stream)
        return out

    def all_reduce_sum_fused(self, src, nrep, n_in, tail, key, dgamma=None, dbeta=None, C=0):
        """all_reduce_sum of  [sum over `nrep` replicas of src[r][:n_in] | tail]  (tail None: no extra element), optionally adding the local sums to the affine gradients first:
        the collapse / affine-gradient launch in front of every SyncBatchNorm exchange folded into the exchange kernel"""
        site = self.sites.get(key)
        if site is None:
            site = self.sites[key] = len(self.sites)
            assert site < SITES, "more than %d SyncBatchNorm exchange sites" % SITES
        n = n_in + (0 if tail is None else 1)
        assert src.dtype == torch.float32 and src.is_contiguous() and 0 < n <= MAXN and src.numel() >= nrep * n_in
        out = torch.empty(n, dtype=torch.float32, device=src.device)
        pages = (ctypes.c_void_p * self.world)(*[b + site * self.site_granules * 8 for b in self.bases])
        lib.peer_exchange_sum_fused(src.data_ptr(), nrep, n_in, int(tail is not None), float(tail or 0.0), None if dgamma is None else dgamma.data_ptr(),
                                    None if dbeta is None else dbeta.data_ptr(), C, out.data_ptr(), pages, self.world * MAXN, self.rank, self.world,
                                    self.epochs.data_ptr() + 4 * site, self.err.data_ptr(), self.timeout_ms, torch.cuda.current_stream().cuda_stream)
        return out

    def check(self):
        """host-side: raise if a peer ever failed to arrive (synchronises)"""
        if int(self.err.item()) != 0:
            raise RuntimeError("avec_amd.peer: a SyncBatchNorm peer exchange timed out (a rank did not arrive within %d s); the optimizer steps since then were skipped" % (self.timeout_ms // 1000))

    def close(self):
        for p in self._opened:
            lib.raw("avec_peer_buffer_close")(ctypes.c_void_p(p))
        self._opened = []
        if self._own:
            lib.raw("avec_peer_buffer_free")(ctypes.c_void_p(self._own))
            self._own = None


_STATE = {"px": None, "tried": False}


def _agree(ok, device, group):
    import torch.distributed as dist
    flag = torch.tensor([int(ok)], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def setup(device, group=None):
    """collective over all ranks: create the exchange, prove it against all_reduce, agree on the verdict.  Returns the PeerExchange or None.
    Every rank executes the same collectives in the same order whatever fails locally."""
    import torch.distributed as dist
    if _STATE["tried"]:
        return _STATE["px"]
    _STATE["tried"] = True
    if os.environ.get("AVEC_PEER_SYNCBN", "1") == "0" or not (dist.is_available() and dist.is_initialized()):
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    from . import runtime as rt
    if not (rt.dist_min_world() <= world <= MAX_WORLD):        # (AVEC_DIST_SINGLE=1: a one-rank group exchanges with itself -- the same kernels, pages and epochs)
        return None
    px, handle, why = PeerExchange(rank, world, device, group), None, ""
    try:
        handle = px.allocate()
    except Exception as e:
        why = "allocation / IPC export refused: %s" % e
    gathered = [None] * world
    dist.all_gather_object(gathered, (handle, os.uname().nodename), group=group)
    ok = all(h is not None for h, _ in gathered) and all(n == gathered[0][1] for _, n in gathered)
    if ok:
        try:
            px.map([h for h, _ in gathered])
        except Exception as e:
            ok, why = False, "IPC mapping refused: %s" % e
    elif not why:
        why = "a peer could not export its buffer, or the ranks span several nodes"
    if not _agree(ok, device, group):
        print("[avec_amd.peer] rank %d: peer exchange unavailable (%s); SyncBatchNorm statistics go through torch.distributed" % (rank, why or "a peer failed"), flush=True)
        px.close()
        return None
    dist.barrier(group=group)                                     # every buffer is mapped everywhere before the first write
    # self-test: three rounds over two sites against all_reduce (exercises both parity pages and the epoch counters)
    good = True
    on_dev = dist.get_backend(group) == "nccl"
    px.timeout_ms = 3000                                          # a peer whose writes never become visible here must not cost minutes
    for it in range(3):
        for site_key, n in (("selftest_a", 1025), ("selftest_b", 7)):
            v = torch.arange(n, dtype=torch.float32, device=device) * (rank + 1) + it
            got = px.all_reduce_sum(v, site_key) if good else v
            ref = v.clone() if on_dev else v.cpu()
            dist.all_reduce(ref, group=group)
            torch.cuda.synchronize(device)
            good = good and torch.equal(got.cpu(), ref.cpu()) and int(px.err.item()) == 0
    px.timeout_ms = TIMEOUT_MS
    if not _agree(good, device, group):
        print("[avec_amd.peer] rank %d: peer exchange self-test failed; SyncBatchNorm statistics go through torch.distributed" % rank, flush=True)
        px.close()
        return None
    _STATE["px"] = px
    return px


def active():
    return _STATE["px"]


def reset():
    if _STATE["px"] is not None:
        _STATE["px"].close()
    _STATE["px"], _STATE["tried"] = None, False
