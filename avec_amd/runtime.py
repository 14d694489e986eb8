"""Engine state shared by the host-side modules: compute dtype, dropout RNG state, weight shadows and the
flat parameter arena (one fp32 master / gradient buffer per model => one Adam launch, one RCCL all-reduce).

PyTorch is used here for device memory, streams and torch.distributed only."""
import os

import torch

from .lib import BF16, F32, lib

_STATE = {"dtype": None, "rng": {}, "stream_id": 0, "seed": 0x5EED5EED, "sync_bn": False}


def set_compute_dtype(name):
    """'f32' : fp32 storage + fp32 MFMA (exact, parity mode);  'bf16' : bf16 storage + bf16 MFMA, fp32 accumulate."""
    if name is torch.float16 and not _STATE.get("warned_fp16"):
        _STATE["warned_fp16"] = True
        print("[avec_amd] precision torch.float16 (the reference configs' AMP dtype) runs as bf16 storage + bf16 MFMA with fp32 accumulation on MI355X: "
              "same 16-bit traffic, fp32 exponent range, no GradScaler")
    name = {torch.float32: "f32", torch.bfloat16: "bf16", torch.float16: "bf16"}.get(name, name)
    assert name in ("f32", "bf16"), name
    _STATE["dtype"] = name


def compute_dtype():
    if _STATE["dtype"] is None:
        _STATE["dtype"] = os.environ.get("AVEC_DTYPE", "f32")
    return _STATE["dtype"]


def dt():
    return BF16 if compute_dtype() == "bf16" else F32


def act_dtype():
    return torch.bfloat16 if compute_dtype() == "bf16" else torch.float32


def new_stream_id():
    _STATE["stream_id"] += 1
    return _STATE["stream_id"]


def manual_seed(seed):
    _STATE["seed"] = int(seed)
    for t in _STATE["rng"].values():
        t[0] = int(seed)
        t[1] = 0


def rng_state(device):
    """int64[2] = {seed, step} on `device`; kernels derive per-site masks from (seed, step, site id, element)."""
    key = str(device)
    if key not in _STATE["rng"]:
        _STATE["rng"][key] = torch.tensor([_STATE["seed"], 0], dtype=torch.int64, device=device)
    return _STATE["rng"][key]


def advance_rng(device):
    rng_state(device)[1] += 1


def set_sync_batchnorm(flag):
    _STATE["sync_bn"] = bool(flag)


def dist_min_world():
    """2 normally.  AVEC_DIST_SINGLE=1: a ONE-rank process group takes every data-parallel code path (SyncBatchNorm exchanges, second communicator, early / in-graph
    gradient all-reduce): how the RCCL paths are exercised on a single-GPU box (tests/test_gpu_ddp.py)"""
    return 1 if os.environ.get("AVEC_DIST_SINGLE", "0") == "1" else 2


def sync_batchnorm():
    return _STATE["sync_bn"] and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() >= dist_min_world()


# ---- side stream for the weight-gradient GEMMs ------------------------------------------------------------------------------------------
# dW = dY^T X only feeds the optimizer, while dX = dY W feeds the rest of the backward chain: the former is launched on a second HIP stream
# (fork: side waits for main; join: main waits for side at the end of the backward pass / before the gradients are consumed), so the small
# conformer GEMMs of the two kinds could overlap instead of running back to back.  MEASURED (MI355X, B=32, hipGraph step): 46.3 ms with the
# side stream against 43.8 ms without -- the two kernel streams fight for L2 / LDS and the fork/join edges cost more than the tail overlap
# gains -- so it is OFF by default; AVEC_WGRAD_STREAM=1 enables it for experiments.
_SIDE = {"streams": {}, "pending": set(), "enabled": False}


def wgrad_fork(*tensors):
    """-> the side stream to launch a weight-gradient kernel on (or None).  `tensors` are read by that kernel: their storage must outlive it."""
    if not _SIDE["enabled"]:
        return None
    dev = torch.cuda.current_device()
    side = _SIDE["streams"].get(dev)
    if side is None:
        side = _SIDE["streams"][dev] = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    if main == side:
        return None
    side.wait_stream(main)
    for t in tensors:
        t.record_stream(side)
    if dev not in _SIDE["pending"]:
        _SIDE["pending"].add(dev)
        try:                                     # join when the running backward pass finishes (also inside a stream capture)
            torch.autograd.Variable._execution_engine.queue_callback(wgrad_join)
        except RuntimeError:                     # not inside a backward pass: the caller joins
            pass
    return side


def wgrad_join():
    for dev in list(_SIDE["pending"]):
        with torch.cuda.device(dev):
            torch.cuda.current_stream().wait_stream(_SIDE["streams"][dev])
    _SIDE["pending"].clear()


# ---- second stream for the audio branch of the audio-visual encoder ------------------------------------------------------------------------
# The audio and the visual encoder are independent until the fusion module.  The audio branch consists of ~1000 small kernels (M = B*T <= 6400
# rows: a few hundred workgroups each) that cannot fill 256 CUs; running it on its own HIP stream lets them execute beside the ResNet kernels of
# the visual branch (one fork and one join per pass; autograd replays each backward node on the stream of its forward).  The side stream gets
# its own reduction workspace.  AVEC_BRANCH_STREAMS=0 disables it.
_BRANCH = {"streams": {}, "ws": {}, "enabled": os.environ.get("AVEC_BRANCH_STREAMS", "1") != "0"}


def set_branch_streams(flag):
    """Enable / disable the second stream of the audio branch (bench.py times kernels one by one with the streams off)."""
    _BRANCH["enabled"] = bool(flag)


def branch_stream():
    if not _BRANCH["enabled"] or not torch.cuda.is_available():
        return None
    dev = torch.cuda.current_device()
    side = _BRANCH["streams"].get(dev)
    if side is None:
        side = _BRANCH["streams"][dev] = torch.cuda.Stream(device=dev)
        buf = torch.zeros(WORKSPACE_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", dev))
        with torch.cuda.device(dev):
            lib.set_reduce_workspace_stream(buf.data_ptr(), WORKSPACE_BYTES, side.cuda_stream)
        _BRANCH["ws"][dev] = buf
    return None if torch.cuda.current_stream() == side else side


# ---- pool of pre-zeroed scratch --------------------------------------------------------------------------------------------------------
# BatchNorm statistic accumulators (64 replicas x 2C floats per layer, 2C per layer in backward) must start at zero; ~90 of them per step
# would each cost a fill launch.  They are carved out of one buffer that a training step zeroes ONCE at its start (Model.train_step / the
# captured graph body); allocations only move forward inside a step, so nothing handed out is reused before the next reset.  Outside a
# training step the pool is reset by eval_step only when it already exists (tensors returned by an earlier step that are views of it -- a captured step's
# static losses -- are invalidated by the next train or eval call: Model._own_losses hands out copies); direct op calls never reset it: once exhausted, plain torch.zeros
# takes over.
_ZPOOL = {"buf": {}, "off": {}, "floats": 128 << 20}       # 512 MB of address space; only what a step used (its high-water mark: ~0.36 GB at the bench shape, mostly the skewed dS matrices of the attention backward) is re-zeroed


def zeros_scratch(n, device):
    """fp32 [n], zero-initialised; valid until the next reset_zero_pool() on this device"""
    key = str(device)
    buf = _ZPOOL["buf"].get(key)
    if buf is None:
        return torch.zeros(n, dtype=torch.float32, device=device)
    off = _ZPOOL["off"][key]
    n4 = (n + 3) // 4 * 4
    if off + n4 > buf.numel():
        return torch.zeros(n, dtype=torch.float32, device=device)
    _ZPOOL["off"][key] = off + n4
    return buf[off:off + n]


def reset_zero_pool(device, create=True):
    """start of a training step: every scratch buffer of the previous step is dead (same-stream order); re-zero what was handed out.
    create=False (evaluation): a process that never trained does not allocate the pool -- zeros_scratch() then hands out plain torch.zeros."""
    key = str(device)
    buf = _ZPOOL["buf"].get(key)
    if buf is None and not create:
        return
    if buf is None:
        buf = _ZPOOL["buf"][key] = torch.zeros(_ZPOOL["floats"], dtype=torch.float32, device=device)
        _ZPOOL["off"][key] = 0
        _ZPOOL.setdefault("hi", {})[key] = 0
        return
    hi = max(_ZPOOL["hi"].get(key, 0), _ZPOOL["off"][key])      # under graph capture the memset must cover what any replay can use: the high-water mark
    _ZPOOL["hi"][key] = hi
    if hi:
        buf[:hi].zero_()
    _ZPOOL["off"][key] = 0


class _GradBoundaryFn(torch.autograd.Function):
    """identity; its backward runs when the gradient crosses the boundary, i.e. when everything downstream of it has been back-propagated"""

    @staticmethod
    def forward(ctx, x, arena, lo, hi):
        ctx.saved = (arena, lo, hi)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        arena, lo, hi = ctx.saved
        if not getattr(arena, "_boundary_fired", False):
            arena._boundary_fired = True
            arena.early_all_reduce(lo, hi)
        return g, None, None, None


def grad_boundary(x, arena, lo, hi):
    """mark `x` as an input of the sub-network whose parameters live in arena[lo:hi) (distributed training only)"""
    if arena is None or not getattr(arena, "_early_armed", False) or not x.requires_grad:
        return x
    return _GradBoundaryFn.apply(x, arena, lo, hi)


def arena_of(module):
    for p in module.parameters():
        return getattr(p, "_avec_arena", None)
    return None


def ensure_shadows_fresh(module):
    """Refresh the arena's weight shadows on the CURRENT stream (before work is forked to other streams)."""
    for p in module.parameters():
        sh = getattr(p, "_avec_shadow", None)
        if sh is not None and sh.arena is not None:
            sh.arena.ensure_fresh()
            return


_WORKSPACE = {}
WORKSPACE_BYTES = 64 << 20


def _register_workspace(dev):
    """Scratch for the two-level column reductions (include/avec_hip.h: avec_set_reduce_workspace), one per device, zero-initialised.
    The engine launches on one stream per device, as the registration requires."""
    buf = torch.zeros(WORKSPACE_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", dev))
    with torch.cuda.device(dev):
        lib.set_reduce_workspace(buf.data_ptr(), WORKSPACE_BYTES)
    _WORKSPACE[dev] = buf


def stream():
    dev = torch.cuda.current_device()
    if dev not in _WORKSPACE:
        _register_workspace(dev)
    return torch.cuda.current_stream().cuda_stream


def is_dense(t):
    """non-overlapping and dense: the strides are a permutation of a contiguous layout"""
    exp = 1
    for st, sz in sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1):
        if st != exp:
            return False
        exp *= sz
    return True


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("avec_amd: the HIP path needs tensors on a GPU (got %s); there is no CPU fallback -- "
                           "the CPU restatement lives in oracle/ and is test infrastructure only" % t.device)


# --------------------------------------------------------------------------------------------
# weight shadows
# --------------------------------------------------------------------------------------------
class Shadow:
    """Compute-dtype copies of one GEMM weight: `fwd` = physical order [A][Tm][C], `bwd` = [C][Tm][A] (row pitch `ldb`)."""
    __slots__ = ("A", "Tm", "C", "Cp", "need_bwd", "fwd", "bwd", "stamp", "arena", "_table", "ldb", "group")

    def __init__(self, A, Tm, C, need_bwd=True, c_pad=None):
        self.A, self.Tm, self.C, self.need_bwd = A, Tm, C, need_bwd
        self.Cp = C if c_pad is None else c_pad          # row stride of the fwd shadow (zero padded), Tm == 1 only
        assert self.Cp == C or Tm == 1
        self.fwd = self.bwd = None
        self.stamp = None
        self.arena = None
        self._table = None
        self.ldb = None          # row pitch of `bwd` when it is a column slice of a fused group's [C][G*A] matrix
        self.group = None        # FusedLinears this weight belongs to (arena only)


class FusedLinears:
    """G Linear layers that consume the same input (the Q, K, V projections): inside the arena their weights are stored back to back, so
    one [G*A][C] forward shadow, one [C][G*A] backward shadow, one [G*A][C] gradient and one [G*A] bias / bias-gradient vector exist and
    the three GEMMs of the forward pass, the backward-data pass and the weight-gradient pass become one each."""
    __slots__ = ("weights", "biases", "A", "C", "G", "fwd", "bwd", "wgrad", "bias", "bgrad")


def fuse_linears(module, weights, biases):
    """Declare that `weights` (same shape [A][C]) and `biases` should be laid out contiguously by ParamArena (state_dict names are unchanged)."""
    groups = module.__dict__.setdefault("_avec_fused", [])
    groups.append((tuple(weights), tuple(biases)))


def fused_group(weight):
    sh = getattr(weight, "_avec_shadow", None)
    if sh is None or sh.arena is None or sh.group is None:
        return None
    sh.arena.ensure_fresh()
    return sh.group


def register_weight(param, A, Tm, C, need_bwd=True, c_pad=None):
    param._avec_shadow = Shadow(A, Tm, C, need_bwd, c_pad)
    return param


def _shadow_blocks(sh):
    """workgroups of avec_shadow_refresh for one weight: one per 64 x 64 (A x C) tile and tap"""
    return sh.Tm * ((sh.A + 63) // 64) * ((sh.C + 63) // 64)


def _refresh_single(param, sh):
    n = sh.A * sh.Tm * sh.C
    nf = sh.A * sh.Tm * sh.Cp
    adt = act_dtype()
    buf = torch.zeros(nf + (n if sh.need_bwd else 0), dtype=adt, device=param.device)
    blocks = _shadow_blocks(sh)
    table = torch.tensor([0, 0, nf if sh.need_bwd else -1, sh.A, sh.Tm, sh.C, 0, blocks, sh.Cp, 0], dtype=torch.int64, device=param.device)
    src = param.detach()
    assert is_dense(src)
    lib.shadow_refresh(dt(), src.data_ptr(), buf.data_ptr(), table.data_ptr(), 1, blocks, stream())
    sh.fwd = buf[:nf]
    sh.bwd = buf[nf:] if sh.need_bwd else None
    sh._table = table  # keep alive until the kernel ran


def shadow(param):
    sh = param._avec_shadow
    if sh.arena is not None:
        sh.arena.ensure_fresh()
        return sh
    stamp = (param._version, param.data_ptr(), compute_dtype())
    if sh.stamp != stamp:
        _refresh_single(param, sh)
        sh.stamp = stamp
    return sh



# --------------------------------------------------------------------------------------------
# flat parameter arena
# --------------------------------------------------------------------------------------------
class ParamArena:
    def __init__(self, module):
        params = []
        seen = set()
        # fused groups (FusedLinears) are emitted together, in declaration order, at the position of their first member
        lead, self._fused = {}, []
        own = {id(p) for p in module.parameters()}
        for m in module.modules():
            for ws, bs in m.__dict__.get("_avec_fused", []):
                if not all(id(t) in own for t in ws + bs):
                    continue
                ok = all(w.shape == ws[0].shape and w.numel() % 8 == 0 and getattr(w, "_avec_shadow", None) is not None and w._avec_shadow.Tm == 1
                         and w._avec_shadow.Cp == w._avec_shadow.C and w._avec_shadow.need_bwd for w in ws) and all(b is not None and b.numel() % 4 == 0 for b in bs)
                if ok and not any(id(t) in lead for t in ws + bs):
                    self._fused.append((ws, bs))
                    for grp in (ws, bs):
                        for t in grp:
                            lead[id(t)] = grp
        for p in module.parameters():
            if id(p) in seen:
                continue
            for q in lead.get(id(p), (p,)):
                if id(q) not in seen:
                    seen.add(id(q))
                    params.append(q)
        assert params and all(p.is_cuda for p in params), "ParamArena needs the model on a GPU"
        dev = params[0].device
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(params, self.offsets):
            assert p.dtype == torch.float32 and is_dense(p), "unsupported parameter layout"
            new = self.master[o:o + p.numel()].as_strided(p.shape, p.stride())
            new.copy_(p.data)
            p.data = new
            p.grad = self.grad[o:o + p.numel()].as_strided(p.shape, p.stride())
            p._avec_arena = self
        self.device = dev
        self._shadow_dtype = None
        self.dirty = True
        self._vstamp = None
        # any load_state_dict on the model OR on one of its sub-modules (the reference configs transplant the LRW front-end this way,
        # configs/LRS23/AV/EffConfInterCTC.py:70-75) rewrites master weights: the compute-dtype shadows must follow
        for m in module.modules():
            if any(True for _ in m.parameters(recurse=False)):
                m.register_load_state_dict_post_hook(lambda _m, _keys, arena=self: arena.mark_dirty())
        self._build_shadows()

    def _build_shadows(self):
        adt = act_dtype()
        rows, soff, blocks = [], 0, 0
        member = {}                               # id(weight) -> (group index, position)
        for gi, (ws, _) in enumerate(self._fused):
            for k, w in enumerate(ws):
                member[id(w)] = (gi, k)
        gfwd, gbwd = {}, {}
        for p, o in zip(self.params, self.offsets):
            sh = getattr(p, "_avec_shadow", None)
            if sh is None:
                continue
            n = sh.A * sh.Tm * sh.C
            ldb = 0
            if id(p) in member:                   # fused: forward shadows back to back, backward shadows side by side in one [C][G*A] matrix
                gi, k = member[id(p)]
                G = len(self._fused[gi][0])
                if k == 0:
                    gfwd[gi] = soff
                    soff += G * n
                    gbwd[gi] = soff
                    soff += G * n
                fwd, bwd, ldb = gfwd[gi] + k * n, gbwd[gi] + k * sh.A, G * sh.A
            else:
                fwd = soff
                soff += (sh.A * sh.Tm * sh.Cp + 7) // 8 * 8
                bwd = -1
                if sh.need_bwd:
                    bwd = soff
                    soff += (n + 7) // 8 * 8
            nb = _shadow_blocks(sh)
            rows.append([o, fwd, bwd, sh.A, sh.Tm, sh.C, blocks, nb, sh.Cp, ldb])
            blocks += nb
        self.shadow = torch.zeros(max(soff, 8), dtype=adt, device=self.device)
        self.table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self.n_entries, self.total_blocks = len(rows), blocks
        i = 0
        for p in self.params:
            sh = getattr(p, "_avec_shadow", None)
            if sh is None:
                continue
            _, fwd, bwd, A, Tm, C, _, _, Cp, ldb = rows[i]
            n = A * Tm * C
            sh.fwd = self.shadow[fwd:fwd + A * Tm * Cp]
            if ldb:
                sh.bwd, sh.ldb = self.shadow[bwd:bwd + (C - 1) * ldb + A].as_strided((C, A), (ldb, 1)), ldb
            else:
                sh.bwd, sh.ldb = (self.shadow[bwd:bwd + n] if bwd >= 0 else None), None
            sh.arena = self
            sh.group = None
            i += 1
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        for gi, (ws, bs) in enumerate(self._fused):
            g = FusedLinears()
            g.weights, g.biases, g.G = ws, bs, len(ws)
            g.A, g.C = ws[0]._avec_shadow.A, ws[0]._avec_shadow.C
            n = g.A * g.C
            g.fwd = self.shadow[gfwd[gi]:gfwd[gi] + g.G * n]
            g.bwd = self.shadow[gbwd[gi]:gbwd[gi] + g.G * n]
            ow, ob = off_of[id(ws[0])], off_of[id(bs[0])]
            assert all(off_of[id(w)] == ow + k * n for k, w in enumerate(ws)) and all(off_of[id(b)] == ob + k * g.A for k, b in enumerate(bs))
            g.wgrad, g.bias, g.bgrad = self.grad[ow:ow + g.G * n], self.master[ob:ob + g.G * g.A], self.grad[ob:ob + g.G * g.A]
            for w in ws:
                w._avec_shadow.group = g
        self._shadow_dtype = compute_dtype()
        self.dirty = True

    def ensure_fresh(self):
        if self._shadow_dtype != compute_dtype():
            self._build_shadows()
        if self.dirty:
            lib.shadow_refresh(dt(), self.master.data_ptr(), self.shadow.data_ptr(), self.table.data_ptr(), self.n_entries,
                               self.total_blocks, stream())
            self.dirty = False
            self.refresh_count = getattr(self, "refresh_count", 0) + 1      # versions everything derived from the shadows (ops._pos_group_entry: the position projections)
            if getattr(self, "_fp8", None) is not None:          # e4m3 weight shadows follow the masters too (avec_amd/fp8.py)
                self._fp8.refresh()

    def prefix_blocks(self, module):
        """Workgroups of the shadow refresh that belong to `module`'s weights when those lead the shadow table (the visual front-end of the audio-visual
        encoder does), else 0: ensure_fresh_split() refreshes that prefix on the current stream and the rest on another one."""
        ids = {id(p) for p in module.parameters()}
        blocks, inside = 0, True
        for p in self.params:
            sh = getattr(p, "_avec_shadow", None)
            if sh is None:
                continue
            if id(p) in ids:
                if not inside:
                    return 0
                blocks += _shadow_blocks(sh)
            else:
                inside = False
        return blocks

    def ensure_fresh_split(self, first_blocks, side):
        """ensure_fresh() with the refresh cut in two: shadow blocks [0, first_blocks) on the current stream (their consumer follows at once), the rest on
        `side` (which must already wait for the current stream).  Returns the event the current stream has to wait for before it reads any other shadow,
        or None when nothing was stale.  The fp8 shadows and a dtype change take the one-launch path."""
        if self._shadow_dtype != compute_dtype() or getattr(self, "_fp8", None) is not None or not (0 < first_blocks < self.total_blocks):
            self.ensure_fresh()
            side.wait_stream(torch.cuda.current_stream())      # the one-launch refresh (and the fp8 one) was enqueued AFTER the caller's side.wait_stream: order `side` behind it
            return None
        if not self.dirty:
            return None
        lib.shadow_refresh_range(dt(), self.master.data_ptr(), self.shadow.data_ptr(), self.table.data_ptr(), self.n_entries, 0, first_blocks, stream())
        with torch.cuda.stream(side):
            lib.shadow_refresh_range(dt(), self.master.data_ptr(), self.shadow.data_ptr(), self.table.data_ptr(), self.n_entries, first_blocks,
                                     self.total_blocks - first_blocks, stream())
            ev = torch.cuda.Event()
            ev.record(side)
        self.dirty = False
        self.refresh_count = getattr(self, "refresh_count", 0) + 1
        return ev

    def mark_dirty(self):
        """The master weights changed outside the Adam kernel: refresh the shadows before the next GEMM.  Called automatically by load_state_dict
        (model or sub-module) and by check_versions(); call it by hand after editing weights through `.data` (which leaves no trace)."""
        self.dirty = True

    def check_versions(self):
        """In-place edits of parameters (p.copy_, init functions, EMA copies under no_grad) bump the tensors' version counters: compare their sum
        with the last one seen (once per forward pass, ~0.1 ms of host time) and mark the shadows stale when it moved."""
        stamp = 0
        for p in self.params:
            stamp += p._version
        if stamp != self._vstamp:
            if self._vstamp is not None:
                self.dirty = True
            self._vstamp = stamp

    def zero_grad(self):
        self.grad.zero_()

    def range_of(self, module):
        """[lo, hi) of the arena occupied by `module`'s parameters, or None if they are not one contiguous block"""
        ids = {id(p) for p in module.parameters()}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        last = idx[-1]
        hi = self.offsets[last + 1] if last + 1 < len(self.offsets) else self.numel
        return self.offsets[idx[0]], hi

    def early_all_reduce(self, lo, hi):
        """Start the gradient all-reduce of arena range [lo, hi) now (its gradients are final), on the current stream, asynchronously: the exchange
        of the audio-visual encoder's and the audio encoder's gradients overlaps the rest of the backward pass.  No-op unless armed by train_step."""
        if not getattr(self, "_early_armed", False) or hi <= lo:
            return
        import torch.distributed as dist
        from . import ops
        ops.flush_param_grads()                   # queued weight / LayerNorm gradients of this stream belong to the range: they must be in the arena first
        if getattr(self, "_sync_collectives", False):
            # inside a hipGraph capture: a blocking collective is enqueued on the calling stream's communicator in program order and becomes part of the graph
            # (no work handle to wait for on the host); from the audio branch's stream it runs beside the visual branch's backward
            dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=False, group=collective_group())
            self._early.append((lo, hi, None))
            return
        self._early.append((lo, hi, dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True, group=collective_group())))

    def arm_early_all_reduce(self, flag, sync=False):
        self._early_armed, self._early, self._sync_collectives = bool(flag), [], bool(sync)

    def all_reduce_grads(self, bucket_bytes=64 << 20):
        """DDP gradient averaging: sum over ranks in large contiguous buckets (RCCL over xGMI); the 1/world factor is
        folded into the Adam kernel's grad_scale.  Ranges already started by early_all_reduce() are only waited for."""
        early = sorted(getattr(self, "_early", []), key=lambda t: t[0])
        pos = 0
        if os.environ.get("AVEC_DIAG_SKIP_GRAD_ALLREDUCE") == "1":      # diagnostics only (two ranks sharing one GPU over gloo: isolates compute + SyncBatchNorm exchange cost)
            early = []
            pos = self.numel
        sync = getattr(self, "_sync_collectives", False)
        for lo, hi, _ in early + [(self.numel, self.numel, None)]:
            if lo > pos:
                all_reduce_flat(self.grad[pos:lo], bucket_bytes, sync=sync)
            pos = max(pos, hi)
        for _, _, work in early:
            if work is not None:
                work.wait()
        self._early, self._early_armed, self._sync_collectives = [], False, False


GRAD_ALLREDUCE_BF16 = os.environ.get("AVEC_GRAD_ALLREDUCE_BF16", "0") == "1"      # opt-in: bf16 payload for the flat gradient all-reduce: half the bytes on
# every xGMI link; each rank's gradient is rounded to bf16 once before the sum (the reference all-reduces fp32 gradients: default off)


def all_reduce_flat(flat, bucket_bytes=64 << 20, bf16_payload=None, sync=False):
    """Sum a flat buffer over all ranks in contiguous slices of `bucket_bytes` (asynchronous, then waited in order; sync: blocking calls in program order on the
    calling stream -- what a hipGraph capture records)."""
    import torch.distributed as dist
    if (GRAD_ALLREDUCE_BF16 if bf16_payload is None else bf16_payload) and flat.dtype == torch.float32:
        wire = flat.to(torch.bfloat16)
        all_reduce_flat(wire, bucket_bytes, bf16_payload=False, sync=sync)
        flat.copy_(wire)
        return flat
    n = flat.numel()
    step = max(bucket_bytes // flat.element_size(), 1)
    if sync:
        for s in range(0, n, step):
            dist.all_reduce(flat[s:min(s + step, n)], op=dist.ReduceOp.SUM, async_op=False)
        return flat
    works = [dist.all_reduce(flat[s:min(s + step, n)], op=dist.ReduceOp.SUM, async_op=True) for s in range(0, n, step)]
    for w in works:
        w.wait()
    return flat


def collective_group():
    """Process group for a collective issued from the CURRENT stream.  Collectives of one communicator execute in issue order on its internal stream: the
    audio encoder (side stream) issues all its SyncBatchNorm exchanges before the visual encoder issues its first one, so on the default group the visual
    branch would wait for the whole audio branch and the two-stream overlap would be lost on every multi-GPU run.  The side stream therefore gets its own
    communicator (created once, by every rank, at distribute time: ensure_branch_group)."""
    g = _BRANCH.get("group")
    if g is None or not torch.cuda.is_available():
        return None
    side = _BRANCH["streams"].get(torch.cuda.current_device())
    return g if (side is not None and torch.cuda.current_stream() == side) else None


def ensure_branch_group():
    """collective on all ranks: create the side-stream communicator (no-op without torch.distributed or when the branch streams are off)"""
    import torch.distributed as dist
    if _BRANCH.get("group") is None and _BRANCH["enabled"] and dist.is_available() and dist.is_initialized() and dist.get_world_size() >= dist_min_world():
        _BRANCH["group"] = dist.new_group()
    return _BRANCH.get("group")


def all_reduce_small(vec, key):
    """all-reduce(sum) of a short fp32 statistics vector: ONE peer-write kernel over xGMI when the exchange is up (avec_amd/peer.py), else a torch.distributed
    collective on the communicator of the current stream.  `key` identifies the exchange site (layer, direction)."""
    import torch.distributed as dist
    from . import peer
    px = peer.active()
    if px is not None and vec.is_cuda:
        return px.all_reduce_sum(vec, key)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=collective_group())
    return vec


def sync_bn_stats(stats, nrep, C, count, key=None):
    """SyncBatchNorm statistic exchange: collapse the `nrep` replicated [sum | sumsq] partials, append the local element count and sum
    the (2C+1)-vector over ranks.  Returns the reduced vector (global sum, global sumsq, global count)."""
    if stats.is_cuda:
        from . import peer
        px = peer.active()
        if px is not None and nrep * 2 * C <= 2048 :
            # few replicas: collapsed inside the exchange kernel (one launch).  The 64 replicas of the GEMM epilogues stay with the grid-wide collapse kernel: summed by
            # the exchange's single workgroup they cost more than the launch they save (measured: 22.45 vs 21.75 ms per step under a one-rank group)
            return px.all_reduce_sum_fused(stats, nrep, 2 * C, float(count), key)
        red = torch.empty(2 * C + 1, dtype=torch.float32, device=stats.device)
        lib.bn_collapse(stats.data_ptr(), nrep, float(count), red.data_ptr(), C, stream())
    else:       # (CPU tensors: the gloo unit tests of the exchange itself)
        red = torch.cat([stats.view(nrep, 2 * C).sum(0), torch.full((1,), float(count), dtype=stats.dtype, device=stats.device)])
    return all_reduce_small(red, key)
