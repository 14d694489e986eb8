"""Launchers over the C ABI (raw pointers + sizes) and the autograd Functions that compose them.

Every Function writes parameter gradients straight into `param.grad` (accumulating; the flat gradient
arena when the model owns one) and returns None for them, so the backward pass is a fixed sequence of
HIP launches on the current stream (graph-capturable)."""
import ctypes
import os

import torch

from . import fp8
from . import runtime as rt
from .lib import (ACT_NONE, ACT_RELU, ACT_SWISH, BF16, LN_GROUP_MAX, ROWS_CONV_BWD, ROWS_CONV_FWD, ROWS_PLAIN, ROWS_STEM3D, TN_GROUP_MAX, WGRAD_GROUP_MAX, Attn, Epilogue, LnItem, Rows, TnBatched, TnItem,
                  WgradItem, lib)

_byref = ctypes.byref


class KernelTimer:
    """HIP-event timing of the GEMM-family launches (bench.py roofline): events are recorded on the launch stream around each
    launch while enabled; durations are read after the timed region."""
    NAMES = {(0, ROWS_CONV_FWD): "gemm_nt<conv_fwd>", (0, ROWS_CONV_BWD): "gemm_nt<conv_bwd_data>", (0, ROWS_STEM3D): "gemm_nt<stem3d>",
             (0, ROWS_PLAIN): "gemm_nt<plain>", (1, ROWS_CONV_FWD): "gemm_tn<conv_wgrad>", (1, ROWS_STEM3D): "gemm_tn<stem3d_wgrad>",
             (1, ROWS_PLAIN): "gemm_tn<plain>", (2, 0): "conv3x3_slab<fwd>", (2, 1): "conv3x3_slab<bwd_data>", (2, 2): "conv3x3_slab<wgrad>"}

    def __init__(self):
        self.enabled = False
        self.records = []

    def reset(self, enabled):
        self.enabled, self.records = enabled, []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, key, flops, abytes=0.0):
        """abytes: ALGORITHMIC bytes of the launch (operands once + results once), next to the PMC traffic in the roofline rows"""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        name = lib.raw("avec_last_kernel")()                  # the kernel instance the entry point chose (api.hip)
        self.records.append((key, flops, e0, e1, name.decode() if name else "", float(abytes)))

    def bracket_overhead_us(self, n=200):
        """what an event pair around NOTHING measures (the record / timestamp cost that every bracketed launch carries): calibrated live, subtracted from the per-launch
        durations so that they agree with rocprofv3's kernel durations of the graph-replayed step (round 3: 16.0 us here vs 13.0 us there for the dominant row)"""
        evs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        d = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        return d[len(d) // 2]

    def summary(self, peak_tflops, steps=1):
        """`roofline` object of bench.py: one row per kernel INSTANCE (rocprofv3's kernel names), the top row = the kernel with the largest total time;
        `families` groups the same launches by addressing mode.  FLOPs are algorithmic (2 x MACs of the product the launch stands for: structurally
        zero taps of a strided backward-data convolution are not counted)."""
        if not self.records:
            return None
        torch.cuda.synchronize()
        ovh = self.bracket_overhead_us() * 1e-6
        agg, rows = {}, {}
        for key, flops, e0, e1, name, ab in self.records:
            dt = max(e0.elapsed_time(e1) * 1e-3 - ovh, 1e-7)
            a = agg.setdefault(key, [0.0, 0.0, 0])
            a[0] += dt; a[1] += flops; a[2] += 1
            r = rows.setdefault(name or self.NAMES.get(key, str(key)), [0.0, 0.0, 0, 0.0])
            r[0] += dt; r[1] += flops; r[2] += 1; r[3] += ab

        def row(name, v):
            return {"kernel": name, "launches_per_step": round(v[2] / steps, 1), "avg_us": round(1e6 * v[0] / v[2], 2), "ms_per_step": round(1e3 * v[0] / steps, 3),
                    "alg_gflop_per_launch": round(v[1] / v[2] / 1e9, 3), "alg_bytes_per_launch": round(v[3] / v[2]), "tflops": round(v[1] / v[0] / 1e12, 1),
                    "frac": round(v[1] / v[0] / 1e12 / peak_tflops, 4)}
        ordered = sorted(rows.items(), key=lambda kv: -kv[1][0])
        top_name, top = ordered[0]
        t_all = sum(v[0] for v in rows.values()); f_all = sum(v[1] for v in rows.values())
        achieved = top[1] / top[0] / 1e12
        return {"bound": "mfma", "kernel": top_name, "achieved": round(achieved, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(achieved / peak_tflops, 5), "traffic": None, "launches": round(top[2] / steps, 1), "avg_launch_ms": round(1e3 * top[0] / top[2], 5),
                "alg_gflop_per_launch": round(top[1] / top[2] / 1e9, 3), "alg_bytes_per_launch": round(top[3] / top[2]),
                "event_bracket_overhead_us": round(ovh * 1e6, 2),
                "rows": [row(n, v) for n, v in ordered[:8]],
                "gemm_family_total": {"ms_per_step": round(1e3 * t_all / steps, 3), "tflops": round(f_all / t_all / 1e12, 1), "frac": round(f_all / t_all / 1e12 / peak_tflops, 4)},
                "families": {self.NAMES.get(k, str(k)): {"ms_total": round(1e3 * v[0] / steps, 3), "tflops": round(v[1] / v[0] / 1e12, 2), "launches": round(v[2] / steps, 1)}
                             for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}}


KERNEL_TIMER = KernelTimer()


def _p(t):
    return None if t is None else t.data_ptr()


# ---- step stamps (measurement aid, AVEC_STAMPS=1): wall-clock marks written by one-wave kernels at named points of the forward and backward passes, on whatever stream the
# point runs on -- they replay with the captured graph, so tools/step_stamps.py sees the real overlap of the two branch streams without a profiler ----
STAMPS = {"on": os.environ.get("AVEC_STAMPS", "0") == "1", "buf": None, "names": []}


def stamp(name):
    if not STAMPS["on"]:
        return
    if STAMPS["buf"] is None:
        STAMPS["buf"] = torch.zeros(256, dtype=torch.int64, device="cuda")
    if name not in STAMPS["names"]:
        STAMPS["names"].append(name)
    lib.stamp(STAMPS["buf"].data_ptr(), STAMPS["names"].index(name), rt.stream())


class StampFn(torch.autograd.Function):
    """identity; stamps `name:f` when the forward pass gets here and `name:b` when the gradient comes back"""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        stamp(name + ":f")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        stamp(ctx.name + ":b")
        return g, None


def mark(x, name):
    return StampFn.apply(x, name) if (STAMPS["on"] and torch.is_tensor(x) and x.requires_grad) else (stamp(name + ":f") or x)


def grad_of(p):
    """fp32 gradient buffer of a parameter (same physical layout), created on first use."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def empty(shape, dtype, ref):
    """uninitialised device tensor.  16-bit matrices whose row is not a whole number of 16-byte chunks (the 180-channel audio stage) get 16 readable bytes behind
    the last row: the LDS-DMA kernels fetch whole chunks (avec_gemm_tn_grouped)."""
    if dtype == torch.bfloat16 and len(shape) == 2 and shape[1] % 8:
        n = shape[0] * shape[1]
        return torch.empty(n + 8, dtype=dtype, device=ref.device)[:n].view(shape)
    return torch.empty(shape, dtype=dtype, device=ref.device)


TN_MULTI = True                     # attention backward: dK, dV and dE as one launch (0: three)
ATTN_ODD_VALU = False             # A/B: odd head widths take the VALU column pass of the attention backward (round 2)
CAST_DEFER = True                   # fp32-input weight gradients: cast once and join the grouped launch (0: a launch of their own)
TNG_ALIGNED_ONLY = False       # A/B: only 16-byte-aligned operands take the grouped weight-gradient launch


def _chunk_readable(t, ld, width, rows):
    """rows of `width` 16-bit elements at stride `ld`: may the kernels read whole 8-element chunks behind the last row?"""
    if width % 8 == 0 and ld % 8 == 0 and t.data_ptr() % 16 == 0:
        return True
    if TNG_ALIGNED_ONLY:
        return False
    if ld >= (width + 7) // 8 * 8:
        return True
    st = t.untyped_storage()
    end = (t.storage_offset() + (rows - 1) * ld + (width + 7) // 8 * 8) * t.element_size()
    return end <= st.nbytes()


def rows_plain(ld, rows_out=1, rows_in=1, step=0):
    r = Rows()
    r.ld, r.rows_out, r.rows_in, r.step = ld, rows_out, rows_in, step
    return r


def rows_conv(H, W, C, KH, KW, stride, pad, OH, OW):
    r = Rows()
    r.H, r.W, r.C, r.KH, r.KW, r.stride, r.pad, r.OH, r.OW = H, W, C, KH, KW, stride, pad, OH, OW
    return r


def gemm_nt_fp8(A, ent, out, M, N, K, *, bias=None, act=ACT_NONE, out_pre=None, drop_p=0.0, sid=0, res=None, alpha=1.0, out_f32=False, ldo=None):
    """forward Linear product on e4m3 operands (avec_amd/fp8.py): quantize the bf16 activation with its own max|x|, then avec_gemm_nt_fp8"""
    Kp = ent.Kp
    q = torch.empty((M, Kp), dtype=torch.uint8, device=A.device)
    lib.fp8_quantize(rt.dt(), A.data_ptr(), K, q.data_ptr(), Kp, M, K, ent.a_amax, 1, rt.stream())
    ep = Epilogue()
    ep.out, ep.ldo, ep.out_f32 = out.data_ptr(), (N if ldo is None else ldo), int(out_f32)
    if out_pre is not None:
        ep.out_pre, ep.ldpre = out_pre.data_ptr(), N
    ep.bias, ep.act, ep.alpha = _p(bias), act, alpha
    if drop_p > 0.0:
        ep.drop_p, ep.rng, ep.rng_stream = drop_p, rt.rng_state(out.device).data_ptr(), sid
    if res is not None:
        ep.res, ep.ldres, ep.res_act = res.data_ptr(), N, 0
    ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
    lib.gemm_nt_fp8(q.data_ptr(), Kp, ent.wq, Kp, M, N, Kp, ent.a_amax, ent.w_amax, _byref(ep), rt.stream())
    if ev is not None:
        KERNEL_TIMER.stop(ev, (0, 0), 2.0 * M * N * K)
    return out


def gemm_nt(A, W, out, M, N, K, *, rows=None, mode=ROWS_PLAIN, a_f32=False, ldw=None, bias=None, act=ACT_NONE, out_pre=None,
            drop_p=0.0, sid=0, res=None, res_act=False, alpha=1.0, dact_z=None, dact=0, colsum=None, stats=None, out_f32=False,
            ldo=None, ldres=None, dtype=None, flops=None, bnb=None, res_cls0=False, abytes=None, res_mask=None):
    ep = Epilogue()
    ep.out, ep.ldo, ep.out_f32 = out.data_ptr(), (N if ldo is None else ldo), int(out_f32)
    if out_pre is not None:
        ep.out_pre, ep.ldpre = out_pre.data_ptr(), N
    ep.bias = _p(bias)
    ep.act = act
    if drop_p > 0.0:
        ep.drop_p, ep.rng, ep.rng_stream = drop_p, rt.rng_state(out.device).data_ptr(), sid
    if res is not None:
        ep.res, ep.ldres, ep.res_act, ep.res_cls0 = res.data_ptr(), (N if ldres is None else ldres), int(res_act), int(res_cls0)
        if res_mask is not None:
            ep.res_mask = res_mask.data_ptr()
    ep.alpha = alpha
    if dact_z is not None:
        ep.dact_z, ep.ldz, ep.dact = dact_z.data_ptr(), N, dact
    ep.colsum, ep.stats = _p(colsum), _p(stats)
    if bnb is not None:       # BatchNorm-backward fusion (avec_hip.h): (y, ss | None, mask tensor | None, replicated stats)
        ep.bnb_y, ep.ldby, ep.stats = bnb.y.data_ptr(), N, bnb.stats.data_ptr()
        if bnb.mask_z is not None:
            ep.dact_z, ep.ldz, ep.dact, ep.bnb_mask = bnb.mask_z.data_ptr(), N, 2, 0
        else:
            ep.bnb_ss, ep.bnb_mask = bnb.ss.data_ptr(), 1
    if rows is None:
        rows = rows_plain(K)
    ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
    lib.gemm_nt(rt.dt() if dtype is None else dtype, A.data_ptr(), _byref(rows), mode, int(a_f32), W.data_ptr(),
                K if ldw is None else ldw, M, N, K, _byref(ep), rt.stream())
    if ev is not None:
        esz = A.element_size()
        KERNEL_TIMER.stop(ev, (0, mode), 2.0 * M * N * K if flops is None else flops,
                          ((M * K + N * K) * esz + M * N * (4 if out_f32 else esz)) if abytes is None else abytes)
    return out


# (The module-level switches of this file -- TN_MULTI, PREP_FUSE, LN_PAIR, RELU_BITMASK, STEM3P, ... -- are plain constants since round 6: the alternatives they select
# are kept for the parity tests that flip them in-process (tests/test_gpu_round3.py, test_gpu_round4.py, test_gpu_parity.py), not as environment switches.)
# ---- deferred parameter gradients ------------------------------------------------------------------------------------------------------------
# Weight gradients (dW = dY^T X), bias gradients and LayerNorm gamma/beta gradients only feed the optimizer: nothing in the backward chain waits for
# them.  Launched one by one they are ~45 % of a conformer block's backward launches, each latency-bound (a few hundred workgroups, 12-21 us).  Inside a
# backward pass they are therefore QUEUED per stream (the operand tensors stay referenced) and submitted as grouped launches (avec_gemm_tn_grouped,
# avec_layernorm_param_grads_grouped: one grid over the tiles of up to 32 / 40 products) when a queue fills, at the explicit flush points (end of the audio
# branch, before an early gradient all-reduce) and by a final autograd callback when the backward pass ends.  ops.DEFER_WGRAD = False restores immediate launches.
DEFER_WGRAD = os.environ.get("AVEC_DEFER_WGRAD", "1") != "0"
_TN_WHY = False      # debugging aid: print every weight-gradient product that misses the grouped launch
_TN_FLUSH_AT = TN_GROUP_MAX
_DEFER = {"queues": {}, "task": -1}


class _PendingGrads:
    __slots__ = ("stream", "tn", "ln", "keep", "flops", "cw", "cw_flops", "cw64", "cw64_flops")

    def __init__(self, stream):
        self.stream, self.tn, self.ln, self.keep, self.flops = stream, [], [], [], 0.0
        self.cw, self.cw_flops = [], 0.0           # 3x3 weight gradients of the wide ResNet layers (avec_wgrad3x3_c128_grouped)
        self.cw64, self.cw64_flops = [], 0.0       # ... of the 64-channel layers (avec_wgrad3x3_c64_grouped)


def _in_backward():
    return DEFER_WGRAD and torch._C._current_graph_task_id() != -1


def reset_backward_state():
    """forget everything a backward pass leaves between its nodes (queued parameter-gradient products, hand-over tables, position-group counters): called when a pass
    was aborted (a graph capture that failed midway) before the step is run again"""
    for q in _DEFER["queues"].values():
        q.tn, q.ln, q.keep, q.flops, q.cw, q.cw_flops, q.cw64, q.cw64_flops = [], [], [], 0.0, [], 0.0, [], 0.0
    _DEFER["task"] = -1
    _PREP_READY["task"], _PREP_READY["m"] = -1, {}
    _LN2_READY["task"], _LN2_READY["m"] = -1, {}
    for ent in _POS_CACHE.values():
        ent[1].de_all, ent[1].remaining = None, ent[1].L


def _pending():
    st = torch.cuda.current_stream()
    q = _DEFER["queues"].get(st.cuda_stream)
    if q is None:
        q = _DEFER["queues"][st.cuda_stream] = _PendingGrads(st)
    task = torch._C._current_graph_task_id()
    if _DEFER["task"] != task:                    # first queued item of this backward pass: flush whatever is left when the pass ends
        if _DEFER["task"] != -1:                  # the previous pass never reached its end-of-pass callback (it raised midway): its queued products
            for old in _DEFER["queues"].values():  # belong to another batch and must not be added to this step's gradients
                old.tn, old.ln, old.keep, old.flops, old.cw, old.cw_flops, old.cw64, old.cw64_flops = [], [], [], 0.0, [], 0.0, [], 0.0
        _DEFER["task"] = task
        torch.autograd.Variable._execution_engine.queue_callback(_flush_at_end)
    return q


def _launch_pending(q, part=None):
    """submit queue `q` on the CURRENT stream (its own stream during the pass; the caller's stream, which has joined every branch, at the end of it)"""
    cur = torch.cuda.current_stream()
    if cur.cuda_stream != q.stream.cuda_stream:
        for t in q.keep:
            t.record_stream(cur)
    if q.tn and part in (None, "tn"):
        ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
        for i in range(0, len(q.tn), TN_GROUP_MAX):
            chunk = q.tn[i:i + TN_GROUP_MAX]
            lib.gemm_tn_grouped(BF16, (TnItem * len(chunk))(*chunk), len(chunk), rt.stream())
        if ev is not None:
            KERNEL_TIMER.stop(ev, (1, ROWS_PLAIN), q.flops, sum(2.0 * it.M * (it.I + it.J) + 4.0 * it.I * it.J for it in q.tn))      # P, Q once (bf16) + dW once (fp32)
        q.tn, q.flops = [], 0.0
    if q.ln and part in (None, "ln"):
        for i in range(0, len(q.ln), LN_GROUP_MAX):
            chunk = q.ln[i:i + LN_GROUP_MAX]
            lib.layernorm_param_grads_grouped(rt.dt(), (LnItem * len(chunk))(*chunk), len(chunk), rt.stream())
        q.ln = []
    if q.cw and part in (None, "cw"):
        ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
        for i in range(0, len(q.cw), WGRAD_GROUP_MAX):
            chunk = q.cw[i:i + WGRAD_GROUP_MAX]
            lib.wgrad3x3_c128_grouped((WgradItem * len(chunk))(*chunk), len(chunk), rt.stream())
        if ev is not None:
            KERNEL_TIMER.stop(ev, (2, 2), q.cw_flops, sum(2.0 * 2 * it.images * it.H * it.W * it.C + 4.0 * 9 * it.C * it.C for it in q.cw))      # x, dy once (bf16) + dW once (fp32)
        q.cw, q.cw_flops = [], 0.0
    if q.cw64 and part in (None, "cw"):
        ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
        for i in range(0, len(q.cw64), WGRAD_GROUP_MAX):
            chunk = q.cw64[i:i + WGRAD_GROUP_MAX]
            lib.wgrad3x3_c64_grouped((WgradItem * len(chunk))(*chunk), len(chunk), rt.stream())
        if ev is not None:
            KERNEL_TIMER.stop(ev, (2, 2), q.cw64_flops, sum(2.0 * 2 * it.images * it.H * it.W * it.C + 4.0 * 9 * it.C * it.C for it in q.cw64))
        q.cw64, q.cw64_flops = [], 0.0
    if not q.tn and not q.ln and not q.cw and not q.cw64:
        q.keep = []


def flush_param_grads(all_streams=False):
    """submit the queued parameter-gradient work of the current stream (all_streams: of every stream, on the current one)"""
    if all_streams:
        for q in list(_DEFER["queues"].values()):
            _launch_pending(q)
        return
    q = _DEFER["queues"].get(torch.cuda.current_stream().cuda_stream)
    if q is not None:
        _launch_pending(q)


def _flush_at_end():
    _DEFER["task"] = -1
    flush_param_grads(all_streams=True)


def gemm_tn(P, Q, O, M, I, J, *, ldp=None, q_rows=None, q_mode=ROWS_PLAIN, q_f32=False, ldo=None, dtype=None, p_colsum=None, side=False):
    """O[I][J] (fp32) += P[M][I]^T Q[M][J];  optionally p_colsum[I] += column sums of P (the bias gradient, fused into the same launch).
    side=True (weight gradients): may be queued for a grouped launch (see above) / launched on the side stream (runtime.wgrad_fork)."""
    if q_rows is None:
        q_rows = rows_plain(J)
    if side and q_mode == ROWS_PLAIN and q_f32 and CAST_DEFER and dtype is None and rt.compute_dtype() == "bf16" and _in_backward() and Q.dim() == 2 and Q.dtype == torch.float32:
        # fp32 activations (a layer fed by the residual stream): one cast launch, then the product joins the grouped launch like any other -- instead of a
        # launch of its own on the register-staged kernel plus a column-sum / col_finalize pair for its bias (same rounding: that kernel casts while staging)
        Qb = empty((Q.shape[0], J), rt.act_dtype(), Q)
        lib.cast_rows(rt.dt(), Q.data_ptr(), q_rows.ld, Qb.data_ptr(), J, Q.shape[0], J, rt.stream())
        Q, q_f32, q_rows = Qb, False, rows_plain(J, q_rows.rows_out, q_rows.rows_in, q_rows.step)
    if side and q_mode == ROWS_PLAIN and not q_f32 and dtype is None and rt.compute_dtype() == "bf16" and _in_backward():
        it = TnItem()
        it.P, it.Q, it.O, it.p_colsum = P.data_ptr(), Q.data_ptr(), O.data_ptr(), _p(p_colsum)
        it.ldp, it.ldq, it.ldo, it.M, it.I, it.J = (I if ldp is None else ldp), q_rows.ld, (J if ldo is None else ldo), M, I, J
        it.q_rows_out, it.q_rows_in, it.q_step = q_rows.rows_out, q_rows.rows_in, q_rows.step
        if lib.raw("avec_gemm_tn_grouped_ok")(BF16, _byref(it)) and _chunk_readable(P, it.ldp, I, M) and _chunk_readable(Q, it.ldq, J, M):
            q = _pending()
            q.tn.append(it)
            q.keep += [P, Q]
            q.flops += 2.0 * M * I * J
            if len(q.tn) >= _TN_FLUSH_AT:
                _launch_pending(q, "tn")
            return
    if _TN_WHY and side:
        print("gemm_tn not deferred: M=%d I=%d J=%d q_mode=%d q_f32=%s dtype=%s in_backward=%s P%%16=%d Q%%16=%d ldp=%s ldq=%d step=%d Pdtype=%s Qdtype=%s" % (
            M, I, J, q_mode, q_f32, dtype, _in_backward(), P.data_ptr() % 16, Q.data_ptr() % 16, ldp, q_rows.ld, q_rows.step, P.dtype, Q.dtype), flush=True)
    ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
    st = rt.stream()
    if side and not KERNEL_TIMER.enabled:
        sd = rt.wgrad_fork(P, Q)
        st = st if sd is None else sd.cuda_stream
    lib.gemm_tn_bias(rt.dt() if dtype is None else dtype, P.data_ptr(), I if ldp is None else ldp, Q.data_ptr(), _byref(q_rows), q_mode,
                     int(q_f32), O.data_ptr(), J if ldo is None else ldo, _p(p_colsum), M, I, J, st)
    if ev is not None:
        KERNEL_TIMER.stop(ev, (1, q_mode), 2.0 * M * I * J, (M * I + M * J) * 2.0 + I * J * 4.0)


def layernorm_fwd(x, w, b, M, D, out_f32, eps):
    y = empty((M, D), torch.float32 if out_f32 else rt.act_dtype(), x)
    mean = empty((M,), torch.float32, x)
    rstd = empty((M,), torch.float32, x)
    lib.layernorm_fwd(rt.dt(), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), int(out_f32), mean.data_ptr(), rstd.data_ptr(), M, D, eps,
                      rt.stream())
    return y, mean, rstd


# ---- the first launch of a module's backward, folded into the last launch of the module behind it -----------------------------------------------------
# Every residual module computes  out = x + alpha * Dropout(f(LN(x)))  and starts its backward with  dacc = act(alpha * dropmask * dy)  (avec_grad_prep: ~100
# launches per step).  dy is the dx that the NEXT module's LayerNorm backward wrote a moment ago -- so that kernel writes dacc as a second output
# (avec_layernorm_bwd_prep).  Forward: a module tags its output tensor with (alpha, drop_p, rng stream); the consumer module remembers the tag of its input.
# Backward: the consumer's LayerNorm backward produces the prepared gradient and leaves it under dx's address; the producer module takes it if the address, the
# backward pass, the shape and its own (alpha, drop_p, stream) all match, and launches grad_prep otherwise (gradient accumulated from several consumers, ...).
PREP_FUSE = True
_PREP_READY = {"task": -1, "m": {}}


def _tag_prep(out, alpha, drop_p, sid):
    if PREP_FUSE:
        out._avec_prep = (float(alpha), float(drop_p), int(sid))
    return out


def _prep_request(x):
    return getattr(x, "_avec_prep", None) if PREP_FUSE else None


def _take_prep(dy, M, N, alpha, drop_p, sid):
    if _PREP_READY["task"] != torch._C._current_graph_task_id():
        return None
    r = _PREP_READY["m"].pop(dy.data_ptr(), None)
    if r is not None and r[1:] == (M, N, float(alpha), float(drop_p), int(sid)) and r[0].dtype == rt.act_dtype():
        return r[0]
    return None


def layernorm_bwd(dy, dy_f32, x, mean, rstd, w, b, M, D, dres=None, prep=None):
    """prep = (alpha, drop_p, rng stream) of the module whose output this LayerNorm normalised: also produce its prepared gradient (see above)"""
    dx = empty((M, D), torch.float32, x)
    gw, gb = grad_of(w), grad_of(b)
    if D <= 1024 and D % 4 == 0 and _in_backward():          # (the dx-only kernel and the grouped launch use 4-wide accesses) dx now (one wave per row); d(gamma), d(beta) with the other queued parameter gradients
        if prep is not None:
            task = torch._C._current_graph_task_id()
            if _PREP_READY["task"] != task:
                _PREP_READY["task"], _PREP_READY["m"] = task, {}
            pt = empty((M, D), rt.act_dtype(), x)
            alpha, drop_p, sid = prep
            rng = rt.rng_state(x.device).data_ptr() if drop_p > 0 else None
            lib.layernorm_bwd_prep(rt.dt(), dy.data_ptr(), int(dy_f32), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(dres),
                                   pt.data_ptr(), alpha, drop_p, rng, sid, M, D, rt.stream())
            _PREP_READY["m"][dx.data_ptr()] = (pt, M, D, alpha, drop_p, sid)
        else:
            lib.layernorm_bwd(rt.dt(), dy.data_ptr(), int(dy_f32), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(dres),
                              None, None, M, D, rt.stream())
        it = LnItem()
        it.dy, it.x, it.mean, it.rstd, it.dgamma, it.dbeta = dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gw.data_ptr(), gb.data_ptr()
        it.M, it.D, it.dy_f32 = M, D, int(dy_f32)
        q = _pending()
        q.ln.append(it)
        q.keep += [dy, x, mean, rstd]
        if len(q.ln) >= LN_GROUP_MAX:
            _launch_pending(q, "ln")
        return dx
    lib.layernorm_bwd(rt.dt(), dy.data_ptr(), int(dy_f32), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(dres),
                      gw.data_ptr(), gb.data_ptr(), M, D, rt.stream())
    return dx


def layernorm_bwd_pair(dy2, x2, mean2, rstd2, w2, b2, dres2, x1, mean1, rstd1, w1, req1, ctx1, M, D):
    """LayerNorm backward of a module's pre-norm (dy2: act, residual gradient dres2) AND of the LayerNorm that produced its input (LayerNormFn ctx1; req1 = the prepared
    gradient wanted by the module in front of that one): one launch; returns dx2 and leaves dx1 for LayerNormFn.backward (see LN_PAIR)"""
    dx2, dx1 = empty((M, D), torch.float32, x2), empty((M, D), torch.float32, x2)
    task = torch._C._current_graph_task_id()
    pt, alpha, drop_p, sid = None, 1.0, 0.0, 0
    if req1 is not None:
        if _PREP_READY["task"] != task:
            _PREP_READY["task"], _PREP_READY["m"] = task, {}
        pt = empty((M, D), rt.act_dtype(), x2)
        alpha, drop_p, sid = req1
        _PREP_READY["m"][dx1.data_ptr()] = (pt, M, D, alpha, drop_p, sid)
    rng = rt.rng_state(x2.device).data_ptr() if (pt is not None and drop_p > 0) else None
    lib.layernorm_bwd2(rt.dt(), dy2.data_ptr(), x2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), w2.data_ptr(), dres2.data_ptr(), dx2.data_ptr(),
                       x1.data_ptr(), mean1.data_ptr(), rstd1.data_ptr(), w1.data_ptr(), dx1.data_ptr(), _p(pt), alpha, drop_p, rng, sid, M, D, rt.stream())
    if _LN2_READY["task"] != task:
        _LN2_READY["task"], _LN2_READY["m"] = task, {}
    _LN2_READY["m"][dx2.data_ptr()] = (dx1, ctx1, dx2, dx2._version)
    defer_ln_param_grads(dy2, False, x2, mean2, rstd2, w2, b2, M, D)
    return dx2


def grad_prep(dout, M, N, alpha=1.0, drop_p=0.0, sid=0, dbias=None):
    dacc = empty((M, N), rt.act_dtype(), dout)
    rng = rt.rng_state(dout.device).data_ptr() if drop_p > 0 else None
    lib.grad_prep(rt.dt(), dout.data_ptr(), N, dacc.data_ptr(), alpha, drop_p, rng, sid, _p(dbias), M, N, rt.stream())
    return dacc


def colsum(x, ld, out, M, N):
    lib.colsum(rt.dt(), x.data_ptr(), ld, out.data_ptr(), M, N, rt.stream())


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()




# ============================================================================================
# Linear family
# ============================================================================================
def linear_fwd(x2d, weight, bias, M, *, in_f32, out_f32, act=ACT_NONE, out_pre=None, drop_p=0.0, sid=0, res=None, alpha=1.0,
               rows=None, K=None):
    sh = rt.shadow(weight)
    N, Kw = sh.A, sh.Tm * sh.C
    out = empty((M, N), torch.float32 if out_f32 else rt.act_dtype(), x2d)
    ent = fp8.entry(weight, Kw) if (rows is None and not in_f32 and sh.group is None) else None
    if ent is not None and ent.N == N:
        return gemm_nt_fp8(x2d, ent, out, M, N, Kw, bias=bias, act=act, out_pre=out_pre, drop_p=drop_p, sid=sid, res=res, alpha=alpha, out_f32=out_f32)
    gemm_nt(x2d, sh.fwd, out, M, N, Kw, rows=rows, a_f32=in_f32, bias=bias, act=act, out_pre=out_pre, drop_p=drop_p, sid=sid,
            res=res, alpha=alpha, out_f32=out_f32)
    return out


def linear_bwd_weight(dacc, x2d, weight, M, *, q_f32=False, q_rows=None, ldp=None, bias=None):
    """dW += dacc^T x;  with `bias`: db += column sums of dacc, out of the same GEMM launch"""
    sh = rt.shadow(weight)
    gemm_tn(dacc, x2d, grad_of(weight), M, sh.A, sh.Tm * sh.C, q_f32=q_f32, q_rows=q_rows, ldp=ldp, p_colsum=None if bias is None else grad_of(bias), side=True)


def linear_bwd_input(dacc, weight, M, *, out_f32, res=None, res_act=False, dact_z=None, dact=0, drop_p=0.0, sid=0, colsum_to=None, lda=None, out=None):
    sh = rt.shadow(weight)
    N, K = sh.Tm * sh.C, sh.A            # bwd shadow is [C*Tm][A]
    if out is None:
        out = empty((M, N), torch.float32 if out_f32 else rt.act_dtype(), dacc)
    gemm_nt(dacc, sh.bwd, out, M, N, K, rows=rows_plain(K if lda is None else lda), ldw=sh.ldb, res=res, res_act=res_act, dact_z=dact_z, dact=dact,
            drop_p=drop_p, sid=sid, colsum=colsum_to, out_f32=out_f32)
    return out


class LinearFn(torch.autograd.Function):
    """layers.Linear.forward (nnet/layers.py:64-76): y = x W^T + b.  x fp32 or act; y fp32 or act."""

    @staticmethod
    def forward(ctx, x, weight, bias, in_f32, out_f32):
        rt.require_gpu(x)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M = x2.shape[0]
        y = linear_fwd(x2, weight, bias, M, in_f32=in_f32, out_f32=out_f32)
        ctx.saved = (x2, weight, bias, in_f32, out_f32, M, shp)
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, in_f32, out_f32, M, shp = ctx.saved
        N = weight.shape[0]
        dy = dy.reshape(M, N)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dacc = grad_prep(dy, M, N) if out_f32 else dy
        linear_bwd_weight(dacc, x2, weight, M, q_f32=in_f32, bias=bias)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_bwd_input(dacc, weight, M, out_f32=in_f32).view(shp)
        return dx, None, None, None, None


def linear(x, weight, bias, out_f32=True):
    """x fp32 (converted while staging in bf16 mode) or act; returns fp32 by default."""
    if rt.compute_dtype() == "f32":
        return LinearFn.apply(x.float(), weight, bias, False, True)
    if x.dtype == torch.float32:
        return LinearFn.apply(x, weight, bias, True, out_f32)
    return LinearFn.apply(x.to(rt.act_dtype()), weight, bias, False, out_f32)


# ---- two consecutive LayerNorms as one launch per pass (round 4) --------------------------------------------------------------------------------------------
# The LayerNorm that closes a ConformerBlock is followed at once by the pre-norm of the next block's first feed-forward module: rows are independent, so the closing
# norm's launch also produces the next one's output (avec_layernorm_fwd2) and leaves it on its result tensor; FeedForwardFn picks it up when the tensor, the
# parameters and eps are the ones it was made for.  Backward: the feed-forward module's LayerNorm backward also runs the closing norm's (avec_layernorm_bwd2) and
# leaves dx1 under the address of the gradient it returns; LayerNormFn.backward takes it if that very tensor comes back from autograd (one consumer), else computes.
LN_PAIR = True
_LN2_READY = {"task": -1, "m": {}}


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of an fp32 tensor -> fp32.  nxt = (weight, bias, eps) of a LayerNorm that will read the result next (see above)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, nxt=None):
        rt.require_gpu(x)
        shp = x.shape
        pair = None
        x2 = _f32c(x.reshape(-1, shp[-1]))
        M, D = x2.shape
        if nxt is not None and LN_PAIR and D <= 512 and D % 4 == 0:
            w2, b2, eps2 = nxt
            y, mean, rstd = empty((M, D), torch.float32, x2), empty((M,), torch.float32, x2), empty((M,), torch.float32, x2)
            h2, mean2, rstd2 = empty((M, D), rt.act_dtype(), x2), empty((M,), torch.float32, x2), empty((M,), torch.float32, x2)
            lib.layernorm_fwd2(rt.dt(), x2.data_ptr(), w.data_ptr(), b.data_ptr(), eps, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                               w2.data_ptr(), b2.data_ptr(), eps2, h2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr(), M, D, rt.stream())
            pair = (h2, mean2, rstd2, w2.data_ptr(), float(eps2), M, D, ctx)
        else:
            y, mean, rstd = layernorm_fwd(x2, w, b, M, D, True, eps)
        ctx.saved = (x2, mean, rstd, w, b, M, D, shp, _prep_request(x))
        out = y.view(shp)
        if pair is not None:
            out._avec_ln2 = pair
        return out

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, w, b, M, D, shp, req = ctx.saved
        if _LN2_READY["task"] == torch._C._current_graph_task_id():
            r = _LN2_READY["m"].pop(dy.data_ptr(), None)
            # the feed-forward module behind has done this LayerNorm's backward already -- valid only if dy still IS its dx2 (same storage, not accumulated into since:
            # a second consumer of this norm's output makes autograd add its gradient in place, which bumps the version counter)
            if r is not None and r[1] is ctx and dy.dtype == torch.float32 and dy.is_contiguous() and (dy is r[2] or dy._base is r[2]) and r[2]._version == r[3]:
                defer_ln_param_grads(dy.reshape(M, D), True, x2, mean, rstd, w, b, M, D)
                return r[0].view(shp), None, None, None, None
        dx = layernorm_bwd(_f32c(dy.reshape(M, D)), True, x2, mean, rstd, w, b, M, D, prep=req)
        return dx.view(shp), None, None, None, None


class ActivationFn(torch.autograd.Function):
    """stand-alone Swish / ReLU / GLU(dim=-1) (nnet/activations.py:39-69) on fp32 device tensors"""

    @staticmethod
    def forward(ctx, x, act, dim):
        rt.require_gpu(x)
        assert act in (1, 2, 3)
        assert act != 3 or dim in (-1, x.dim() - 1), "GLU over the last axis"
        x2 = _f32c(x)
        C = x2.shape[-1] // 2 if act == 3 else x2.shape[-1]
        assert act != 3 or x2.shape[-1] == 2 * C, "GLU needs an even last axis"
        rows = x2.numel() // x2.shape[-1]
        y = torch.empty(x2.shape[:-1] + (C,), dtype=torch.float32, device=x2.device)
        lib.act_f32(act, x2.data_ptr(), None, y.data_ptr(), rows, C, 0, rt.stream())
        ctx.saved = (x2, act, rows, C)
        return y.to(x.dtype) if x.dtype != torch.float32 else y

    @staticmethod
    def backward(ctx, dy):
        x2, act, rows, C = ctx.saved
        dyc = _f32c(dy)
        dx = torch.empty_like(x2)
        lib.act_f32(act, x2.data_ptr(), dyc.data_ptr(), dx.data_ptr(), rows, C, 1, rt.stream())
        return dx, None, None


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, sid):
        x = _f32c(x)
        y = torch.empty_like(x)
        lib.dropout_f32(x.data_ptr(), y.data_ptr(), p, rt.rng_state(x.device).data_ptr(), sid, x.numel(), rt.stream())
        ctx.saved = (p, sid)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, sid = ctx.saved
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        lib.dropout_f32(dy.data_ptr(), dx.data_ptr(), p, rt.rng_state(dy.device).data_ptr(), sid, dy.numel(), rt.stream())
        return dx, None, None


# ============================================================================================
# FeedForwardModule (nnet/modules.py:257-289) fused with its macaron residual (nnet/blocks.py:292,301)
#   y = x + alpha * Drop(W2 Drop(Swish(W1 LN(x) + b1)) + b2)
# ============================================================================================
def defer_ln_param_grads(dy, dy_f32, x, mean, rstd, w, b, M, D):
    """queue d(gamma), d(beta) of a LayerNorm whose input gradient was computed elsewhere (inside a backward pass), or compute them now"""
    it = LnItem()
    it.dy, it.x, it.mean, it.rstd, it.dgamma, it.dbeta = dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), grad_of(w).data_ptr(), grad_of(b).data_ptr()
    it.M, it.D, it.dy_f32 = M, D, int(dy_f32)
    if _in_backward() and D % 4 == 0 and D <= 1024:
        q = _pending()
        q.ln.append(it)
        q.keep += [dy, x, mean, rstd]
        if len(q.ln) >= LN_GROUP_MAX:
            _launch_pending(q, "ln")
    else:
        lib.layernorm_param_grads_grouped(rt.dt(), (LnItem * 1)(it), 1, rt.stream())


class FeedForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, eps, alpha, drop_p, sid1, sid2):
        rt.require_gpu(x)
        shp = x.shape
        x2 = _f32c(x.reshape(-1, shp[-1]))
        M, D = x2.shape
        F = w1.shape[0]
        adt = rt.act_dtype()
        ln_prev = None
        pre = getattr(x, "_avec_ln2", None)
        if pre is not None and pre[3] == ln_w.data_ptr() and pre[4] == float(eps) and pre[5:7] == (M, D) and x2.data_ptr() == x.data_ptr() and pre[0].dtype == adt:
            h0, mean, rstd = pre[0], pre[1], pre[2]          # made by the LayerNorm launch that produced x (LayerNormFn, nxt)
            ln_prev = pre[7]
        else:
            h0, mean, rstd = layernorm_fwd(x2, ln_w, ln_b, M, D, False, eps)
        z = empty((M, F), adt, x2)
        h1 = linear_fwd(h0, w1, b1, M, in_f32=False, out_f32=False, act=ACT_SWISH, out_pre=z, drop_p=drop_p, sid=sid1)
        y = linear_fwd(h1, w2, b2, M, in_f32=False, out_f32=True, drop_p=drop_p, sid=sid2, res=x2, alpha=alpha)
        ctx.saved = (x2, mean, rstd, h0, z, h1, ln_w, ln_b, w1, b1, w2, b2, alpha, drop_p, sid1, sid2, M, D, F, shp)
        ctx.prep_req = _prep_request(x)
        ctx.ln_prev = ln_prev
        return _tag_prep(y.view(shp), alpha, drop_p, sid2)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, h0, z, h1, ln_w, ln_b, w1, b1, w2, b2, alpha, drop_p, sid1, sid2, M, D, F, shp = ctx.saved
        dy = _f32c(dy.reshape(M, D))
        dacc = _take_prep(dy, M, D, alpha, drop_p, sid2)
        if dacc is None:
            dacc = grad_prep(dy, M, D, alpha=alpha, drop_p=drop_p, sid=sid2)
        linear_bwd_weight(dacc, h1, w2, M, bias=b2)
        dz = linear_bwd_input(dacc, w2, M, out_f32=False, dact_z=z, dact=ACT_SWISH, drop_p=drop_p, sid=sid1)
        linear_bwd_weight(dz, h0, w1, M, bias=b1)
        dh0 = linear_bwd_input(dz, w1, M, out_f32=False)
        pv = ctx.ln_prev
        if pv is not None and _in_backward() and D <= 512 and D % 4 == 0 and dh0.dtype == rt.act_dtype():
            x1, mean1, rstd1, w1n, b1n, M1, D1, shp1, req1 = pv.saved
            if (M1, D1) == (M, D):
                return (layernorm_bwd_pair(dh0, x2, mean, rstd, ln_w, ln_b, dy, x1, mean1, rstd1, w1n, req1, pv, M, D).view(shp),) + (None,) * 11
        dx = layernorm_bwd(dh0, False, x2, mean, rstd, ln_w, ln_b, M, D, dres=dy, prep=ctx.prep_req)
        return (dx.view(shp),) + (None,) * 11


# ============================================================================================
# AttentionModule (nnet/modules.py:320-339) with RelPos1d / RelPosPatch1d attention (nnet/attentions.py:280-382)
#   y = x + Drop(Wo . Attn(LN(x)) + bo)
# ============================================================================================
_PE_CACHE = {}


def rel_pos_table(T, D, device):
    """RelativeSinusoidalPositionalEncoding rows p = T-1 .. -(T-1) (nnet/embeddings.py:117-126,152), act dtype."""
    key = (T, D, str(device), rt.compute_dtype())
    if key not in _PE_CACHE:
        pos = torch.arange(T - 1, -T, -1, dtype=torch.float32).unsqueeze(1)
        inv = 10000 ** (2 * torch.arange(0, D // 2, dtype=torch.float32).unsqueeze(0) / D)
        ang = pos / inv
        pe = torch.zeros(2 * T - 1, D)
        pe[:, 0::2] = ang.sin()
        pe[:, 1::2] = ang.cos()
        _PE_CACHE[key] = pe.to(device=device, dtype=rt.act_dtype()).contiguous()
    return _PE_CACHE[key]


def _attn_args(qkv, e, lens, len_div, mask, o, lse, B, H, T, d, D, q_full=0):
    a = Attn()
    a.q_full = q_full
    esz = qkv.element_size()
    a.q, a.k, a.v, a.ld = qkv.data_ptr(), qkv.data_ptr() + D * esz, qkv.data_ptr() + 2 * D * esz, 3 * D
    a.e, a.lde = e.data_ptr(), (e.stride(0) if e.dim() == 2 else D)      # (a column slice of a stack's fused position projections: row stride L * D)
    a.lens, a.len_div = _p(lens), len_div
    if mask is not None:
        a.mask, a.mask_bstride = mask.data_ptr(), (T * T if mask.shape[0] > 1 else 0)
    a.o, a.ldo, a.lse = o.data_ptr(), D, lse.data_ptr()
    a.B, a.H, a.T, a.d, a.scale = B, H, T, d, 1.0 / d ** 0.5
    return a


# ---- position projections of a whole stack in one launch ---------------------------------------------------------------------------------------------------
# E_l = pos_layer_l(PE) depends on weights only (nnet/attentions.py:289, nnet/embeddings.py:158: the reference recomputes it per block AND per batch element).
# The pos_layer weights of the consecutive blocks of a stage are laid out back to back in the arena (nnet.ConformerInterCTC declares them with rt.fuse_linears), so
# ONE product PE [2T-1][D] x W_all^T [D][L D] gives every block's E as a column slice, and in backward the L gradients dE_l accumulate in one [2T-1][L D] buffer that
# takes ONE cast and ONE weight-gradient product when the last of them has arrived: 2 (L - 1) launches less per stage and pass.  Versioned by the arena's shadow
# refresh counter (E follows the weights), per sequence length.
POS_GROUP = True
_POS_CACHE = {}


class _PosGroupEntry:
    __slots__ = ("e_all", "pe", "de_all", "remaining", "L", "D", "Tp")


def _pos_group_entry(wp, Tp, D, device):
    """-> (entry, index of this layer inside its group) or (None, 0)"""
    if not POS_GROUP or rt.compute_dtype() != "bf16":
        return None, 0
    grp = rt.fused_group(wp)
    if grp is None or len(grp.weights) < 2 or fp8.enabled():
        return None, 0
    idx = next((k for k, w in enumerate(grp.weights) if w is wp), None)
    if idx is None:
        return None, 0
    arena = wp._avec_shadow.arena
    key = (id(grp), Tp, str(device))
    ent = _POS_CACHE.get(key)
    ver = (getattr(arena, "refresh_count", 0), torch.cuda.current_stream().cuda_stream)
    if ent is None or ent[0] != ver:
        L = len(grp.weights)
        e = _PosGroupEntry()
        e.L, e.D, e.Tp, e.pe = L, D, Tp, rel_pos_table(Tp, D, device)
        e.e_all = empty((2 * Tp - 1, L * D), rt.act_dtype(), e.pe)
        gemm_nt(e.pe, grp.fwd, e.e_all, 2 * Tp - 1, L * D, D, bias=grp.bias)
        e.de_all, e.remaining = None, L
        _POS_CACHE[key] = ent = (ver, e, grp)
        if len(_POS_CACHE) > 64:
            for k in list(_POS_CACHE)[:32]:
                _POS_CACHE.pop(k, None)
    if ent[1].remaining != ent[1].L or ent[1].de_all is not None:      # a backward pass that raised midway left its counters behind: a forward pass starts clean
        ent[1].de_all, ent[1].remaining = None, ent[1].L
    return ent[1], idx


def _pos_group_backward(entry, wp):
    """one layer of the group has accumulated its dE: when all have, cast once and queue ONE weight-gradient product for the whole group"""
    entry.remaining -= 1
    if entry.remaining > 0:
        return
    grp = rt.fused_group(wp)
    R, W = 2 * entry.Tp - 1, entry.L * entry.D
    dea = empty((R, W), rt.act_dtype(), entry.de_all)
    lib.cast_rows(rt.dt(), entry.de_all.data_ptr(), W, dea.data_ptr(), W, R, W, rt.stream())
    gemm_tn(dea, entry.pe, grp.wgrad, R, W, entry.D, p_colsum=grp.bgrad, side=True)
    entry.de_all, entry.remaining = None, entry.L


class AttentionModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lens, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wp, bp, H, patch, eps, drop_p, sid, residual):
        rt.require_gpu(x)
        B, T, D = x.shape
        M, d, adt = B * T, D // H, rt.act_dtype()
        grp = rt.fused_group(wq)
        grp_ok = grp is not None and grp.weights[1] is wk and grp.weights[2] is wv and bq is not None
        x2 = _f32c(x.reshape(-1, D))
        if ln_w is not None:
            h, mean, rstd = layernorm_fwd(x2, ln_w, ln_b, M, D, False, eps)
        else:                      # bare attention layer (forwardQKV called directly): no pre-norm
            mean = rstd = None
            h = empty((M, D), adt, x2)
            lib.cast_rows(rt.dt(), x2.data_ptr(), D, h.data_ptr(), D, M, D, rt.stream())
        if patch > 1:
            Tp = (T + patch - 1) // patch
            hp = empty((B * Tp, D), adt, x2)
            lib.patch_pool_fwd(rt.dt(), h.data_ptr(), hp.data_ptr(), B, T, D, patch, rt.stream())
        else:
            Tp, hp = T, h
        Mp = B * Tp
        if mask is not None:      # dense (B or 1,1,T,T) float mask as the reference API; patch variant min-pools it on the host side
            mask = mask.reshape(mask.shape[0], T, T) if patch == 1 else _pool_mask(mask, T, patch)
            mask = mask.float().contiguous()
        if grp_ok:
            qkv = empty((Mp, 3 * D), adt, x2)
            ent = fp8.entry(wq, D)
            if ent is not None and ent.N == 3 * D:
                gemm_nt_fp8(hp, ent, qkv, Mp, 3 * D, D, bias=grp.bias)
            else:
                gemm_nt(hp, grp.fwd, qkv, Mp, 3 * D, D, bias=grp.bias)               # Q|K|V in one launch
        else:
            qkv = empty((Mp, 3 * D), adt, x2)
            for i, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                sh = rt.shadow(w)
                gemm_nt(hp, sh.fwd, qkv[:, i * D:], Mp, D, D, bias=b, ldo=3 * D)
        pg, pgi = _pos_group_entry(wp, Tp, D, x.device)
        if pg is not None:
            pe, e = pg.pe, pg.e_all[:, pgi * D:(pgi + 1) * D]          # this block's slice of the stack's position projections (one launch per stage and pass)
        else:
            pe = rel_pos_table(Tp, D, x.device)
            e = linear_fwd(pe, wp, bp, 2 * Tp - 1, in_f32=False, out_f32=False)
        o = empty((Mp, D), adt, x2)
        lse = empty((B * H, Tp, 2), torch.float32, x2)
        q_full = T // patch if (patch > 1 and mask is None) else 0
        a = _attn_args(qkv, e, lens, patch, mask, o, lse, B, H, Tp, d, D, q_full)
        lib.relpos_attention_fwd(rt.dt(), _byref(a), rt.stream())
        res = x2 if residual else None
        if patch > 1:
            oo = linear_fwd(o, wo, bo, Mp, in_f32=False, out_f32=False)
            base = x2 if residual else torch.zeros_like(x2)
            y = empty((M, D), torch.float32, x2)
            rng = rt.rng_state(x.device).data_ptr() if drop_p > 0 else None
            lib.patch_unpool_add(rt.dt(), oo.data_ptr(), base.data_ptr(), y.data_ptr(), drop_p, rng, sid, B, T, D, patch, rt.stream())
        else:
            y = linear_fwd(o, wo, bo, M, in_f32=False, out_f32=True, drop_p=drop_p, sid=sid, res=res, alpha=1.0)
        ctx.saved = (x2, mean, rstd, h, hp, qkv, pe, e, o, lse, lens, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wp, bp,
                     H, patch, drop_p, sid, residual, B, T, Tp, D)
        ctx.pos_group = (pg, pgi)
        ctx.prep_req = _prep_request(x)
        return _tag_prep(y.view(B, T, D), 1.0, drop_p, sid) if patch == 1 else y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        (x2, mean, rstd, h, hp, qkv, pe, e, o, lse, lens, mask, ln_w, ln_b, wq, bq, wk, bk, wv, bv, wo, bo, wp, bp,
         H, patch, drop_p, sid, residual, B, T, Tp, D) = ctx.saved
        M, Mp, d, adt = B * T, B * Tp, D // H, rt.act_dtype()
        dy = _f32c(dy.reshape(M, D))
        if patch > 1:
            doo = empty((Mp, D), adt, dy)
            rng = rt.rng_state(dy.device).data_ptr() if drop_p > 0 else None
            lib.patch_unpool_bwd(rt.dt(), dy.data_ptr(), doo.data_ptr(), drop_p, rng, sid, B, T, D, patch, rt.stream())
        else:
            doo = _take_prep(dy, M, D, 1.0, drop_p, sid)
            if doo is None:
                doo = grad_prep(dy, M, D, drop_p=drop_p, sid=sid)
        linear_bwd_weight(doo, o, wo, Mp, bias=bo)
        # dV = P^T dO reads dO per head in whole 16-byte chunks (gemm_tn_batched: J = d rounded up to the vector width): for head widths that are not a multiple of it
        # (d = 45, 90) the last head's last chunk ends up to 12 bytes behind the row -- behind the TENSOR in its last row: 16 readable bytes follow it
        do = torch.empty(Mp * D + 8, dtype=adt, device=dy.device)[:Mp * D].view(Mp, D)
        do = linear_bwd_input(doo, wo, Mp, out_f32=False, out=do)
        dqkv = empty((Mp, 3 * D), adt, dy)
        pg, pgi = ctx.pos_group
        if pg is not None:
            if pg.de_all is None:                    # first layer of the group to reach its backward: the shared [2T-1][L D] accumulator (pre-zeroed pool)
                pg.de_all = rt.zeros_scratch((2 * Tp - 1) * pg.L * D, dy.device).view(2 * Tp - 1, pg.L * D)
            de = pg.de_all[:, pgi * D:(pgi + 1) * D]
        else:
            de = rt.zeros_scratch((2 * Tp - 1) * D, dy.device).view(2 * Tp - 1, D)      # pre-zeroed pool: no fill launch per layer (-0.15 ms per step)
        a = _attn_args(qkv, e, lens, patch, mask, o, lse, B, H, Tp, d, D, T // patch if (patch > 1 and mask is None) else 0)
        a.dout = do.data_ptr()
        esz = dqkv.element_size()
        a.dq, a.lddq = dqkv.data_ptr(), 3 * D
        a.dk, a.dv, a.ldd = dqkv.data_ptr() + D * esz, dqkv.data_ptr() + 2 * D * esz, 3 * D
        a.de, a.ldde = de.data_ptr(), de.stride(0)
        Tld, Rld = (Tp + 7) // 8 * 8, (2 * Tp - 1 + 7) // 8 * 8
        scratch = empty((2, B * H, Tp, Tld), adt, dy)                # P and dS of every (batch, head), row stride padded to 8
        a.pbuf, a.dsbuf, a.ldt = scratch.data_ptr(), scratch.data_ptr() + scratch[0].numel() * esz, Tld
        use_mfma = d % 2 == 0 or not ATTN_ODD_VALU                   # odd head widths (d = 45): the batched products read their Q operand from odd element offsets (unaligned dword loads)
        if use_mfma:
            n_rel = H * B * Tp * Rld                         # zero-initialised, from the step's pre-zeroed pool (was a fill launch per layer on the dependent chain)
            if adt == torch.bfloat16:
                dsrel = rt.zeros_scratch((n_rel + 1) // 2, dy.device).view(torch.bfloat16)[:n_rel].view(H, B * Tp, Rld)
            else:
                dsrel = rt.zeros_scratch(n_rel, dy.device).view(H, B * Tp, Rld)
            a.dsrel, a.ldr = dsrel.data_ptr(), Rld
        lib.relpos_attention_bwd(rt.dt(), _byref(a), rt.stream())
        if use_mfma:
            # dK = dS^T Q and dV = P^T dO per (batch, head); dE_h = sum_b skew(dS)^T Q : batched TN MFMA GEMMs, fp32 accumulation
            # dK and dV: one workgroup per (batch, head, tile) reduces over all Tp rows and stores straight into the K / V thirds of dqkv (no zero-filled
            # fp32 staging, no atomics, no cast pass); dE sums over the batch, so it keeps the accumulate form
            L6 = ctypes.c_longlong * 6
            sK, sV, sE = L6(H * Tp * Tld, Tp * Tld, Tp * 3 * D, d, Tp * 3 * D, d), L6(H * Tp * Tld, Tp * Tld, Tp * D, d, Tp * 3 * D, d), L6(0, B * Tp * Rld, 0, d, 0, d)
            if TN_MULTI:          # the three products as ONE launch (avec_gemm_tn_batched_multi)
                it = (TnBatched * 3)()
                it[0].P, it[0].ldp, it[0].Q, it[0].ldq, it[0].O_act, it[0].ldo = a.dsbuf, Tld, qkv.data_ptr(), 3 * D, dqkv.data_ptr() + D * esz, 3 * D
                it[0].M, it[0].I, it[0].J, it[0].nb_outer, it[0].nb_inner, it[0].strides6 = Tp, Tp, d, B, H, sK
                it[1].P, it[1].ldp, it[1].Q, it[1].ldq, it[1].O_act, it[1].ldo = a.pbuf, Tld, do.data_ptr(), D, dqkv.data_ptr() + 2 * D * esz, 3 * D
                it[1].M, it[1].I, it[1].J, it[1].nb_outer, it[1].nb_inner, it[1].strides6 = Tp, Tp, d, B, H, sV
                it[2].P, it[2].ldp, it[2].Q, it[2].ldq, it[2].O, it[2].ldo = dsrel.data_ptr(), Rld, qkv.data_ptr(), 3 * D, de.data_ptr(), de.stride(0)
                it[2].M, it[2].I, it[2].J, it[2].nb_outer, it[2].nb_inner, it[2].strides6 = B * Tp, 2 * Tp - 1, d, 1, H, sE
                lib.gemm_tn_batched_multi(rt.dt(), it, 3, rt.stream())
            else:
                lib.gemm_tn_batched_store(rt.dt(), a.dsbuf, Tld, qkv.data_ptr(), 3 * D, dqkv.data_ptr() + D * esz, 3 * D, Tp, Tp, d, B, H, sK, rt.stream())
                lib.gemm_tn_batched_store(rt.dt(), a.pbuf, Tld, do.data_ptr(), D, dqkv.data_ptr() + 2 * D * esz, 3 * D, Tp, Tp, d, B, H, sV, rt.stream())
                lib.gemm_tn_batched(rt.dt(), dsrel.data_ptr(), Rld, qkv.data_ptr(), 3 * D, de.data_ptr(), de.stride(0), B * Tp, 2 * Tp - 1, d, 1, H, sE, rt.stream())
        dhp = None
        grp = rt.fused_group(wq)
        if grp is not None and grp.weights[1] is wk and grp.weights[2] is wv and bq is not None:
            gemm_tn(dqkv, hp, grp.wgrad, Mp, 3 * D, D, p_colsum=grp.bgrad, side=True)                    # d(Wq|Wk|Wv), d(bq|bk|bv)
            dhp = empty((Mp, D), adt, dy)
            gemm_nt(dqkv, grp.bwd, dhp, Mp, D, 3 * D)                                            # d(input) = dQ Wq + dK Wk + dV Wv
        else:
            for i, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
                g = dqkv[:, i * D:]
                linear_bwd_weight(g, hp, w, Mp, ldp=3 * D, bias=b)
                dhp = linear_bwd_input(g, w, Mp, out_f32=False, lda=3 * D, res=dhp, res_act=True, out=dhp)
        if pg is not None:
            _pos_group_backward(pg, wp)
        else:
            dea = empty((2 * Tp - 1, D), adt, dy)
            lib.cast_rows(rt.dt(), de.data_ptr(), D, dea.data_ptr(), D, 2 * Tp - 1, D, rt.stream())
            linear_bwd_weight(dea, pe, wp, 2 * Tp - 1, bias=bp)
        if patch > 1:
            dh = empty((M, D), adt, dy)
            lib.patch_pool_bwd(rt.dt(), dhp.data_ptr(), dh.data_ptr(), B, T, D, patch, rt.stream())
        else:
            dh = dhp
        if ln_w is not None:
            dx = layernorm_bwd(dh, False, x2, mean, rstd, ln_w, ln_b, M, D, dres=dy if residual else None, prep=ctx.prep_req)
        else:
            dx = dy.clone() if residual else torch.zeros_like(dy)
            lib.to_f32_rows(rt.dt(), dh.data_ptr(), D, dx.data_ptr(), D, M, D, 1, rt.stream())
        return (dx.view(B, T, D),) + (None,) * 20


class RelPosCoreFn(torch.autograd.Function):
    """softmax((Q K^T + rel_to_abs(Q E^T)) * scale + mask) V per (batch, head) on packed activation-dtype operands (the core of nnet/attentions.py:299-315 /
    :621-640 without the projections): qkv [B*T][3*H*d] = q | k | v, e [2T-1][H*d] (row r <-> relative offset T-1-r) -> o [B*T][H*d].
    Used by the Transformer-XL style / grouped attention classes, whose u / v biases and frame grouping are expressed on the operands."""

    @staticmethod
    def forward(ctx, qkv, e, lens, len_div, mask, B, H, T, d, scale):
        rt.require_gpu(qkv)
        D, adt = H * d, rt.act_dtype()
        assert qkv.dtype == adt and e.dtype == adt and qkv.is_contiguous() and e.is_contiguous() and qkv.shape == (B * T, 3 * D) and e.shape == (2 * T - 1, D)
        assert d % 2 == 0, "the batched-GEMM backward needs an even head width (pad the operands)"
        if mask is not None:
            mask = mask.reshape(mask.shape[0], T, T).float().contiguous()
        o = empty((B * T, D), adt, qkv)
        lse = empty((B * H, T, 2), torch.float32, qkv)
        a = _attn_args(qkv, e, lens, len_div, mask, o, lse, B, H, T, d, D)
        a.scale = scale
        lib.relpos_attention_fwd(rt.dt(), _byref(a), rt.stream())
        ctx.saved = (qkv, e, o, lse, lens, len_div, mask, B, H, T, d, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, e, o, lse, lens, len_div, mask, B, H, T, d, scale = ctx.saved
        D, adt = H * d, rt.act_dtype()
        do = do.to(adt).contiguous()
        dqkv = empty((B * T, 3 * D), adt, do)
        de = torch.zeros((2 * T - 1, D), dtype=torch.float32, device=do.device)
        a = _attn_args(qkv, e, lens, len_div, mask, o, lse, B, H, T, d, D)
        a.scale = scale
        a.dout = do.data_ptr()
        esz = dqkv.element_size()
        a.dq, a.lddq = dqkv.data_ptr(), 3 * D
        a.dk, a.dv, a.ldd = dqkv.data_ptr() + D * esz, dqkv.data_ptr() + 2 * D * esz, 3 * D
        a.de, a.ldde = de.data_ptr(), D
        Tld, Rld = (T + 7) // 8 * 8, (2 * T - 1 + 7) // 8 * 8
        scratch = empty((2, B * H, T, Tld), adt, do)
        a.pbuf, a.dsbuf, a.ldt = scratch.data_ptr(), scratch.data_ptr() + scratch[0].numel() * esz, Tld
        dsrel = torch.zeros((H, B * T, Rld), dtype=adt, device=do.device)
        a.dsrel, a.ldr = dsrel.data_ptr(), Rld
        lib.relpos_attention_bwd(rt.dt(), _byref(a), rt.stream())
        L6 = ctypes.c_longlong * 6
        lib.gemm_tn_batched_store(rt.dt(), a.dsbuf, Tld, qkv.data_ptr(), 3 * D, dqkv.data_ptr() + D * esz, 3 * D, T, T, d, B, H,
                                  L6(H * T * Tld, T * Tld, T * 3 * D, d, T * 3 * D, d), rt.stream())
        lib.gemm_tn_batched_store(rt.dt(), a.pbuf, Tld, do.data_ptr(), D, dqkv.data_ptr() + 2 * D * esz, 3 * D, T, T, d, B, H,
                                  L6(H * T * Tld, T * Tld, T * D, d, T * 3 * D, d), rt.stream())
        lib.gemm_tn_batched(rt.dt(), dsrel.data_ptr(), Rld, qkv.data_ptr(), 3 * D, de.data_ptr(), D, B * T, 2 * T - 1, d, 1, H,
                            L6(0, B * T * Rld, 0, d, 0, d), rt.stream())
        return dqkv, de.to(adt), None, None, None, None, None, None, None, None


def relpos_core_infer(q, k, v, e, lens, mask, B, H, T, Tk, d, scale, want_probs=False):
    """inference form with a key/value cache: q [B*T][H*d], k / v [B*Tk][H*d] (Tk >= T), e [Tk+T-1][H*d], optional dense mask (Bm, T, Tk) -> o [B*T][H*d]
    (and the attention probabilities (B, H, T, Tk) when asked: the reference returns them next to the updated cache, nnet/attentions.py:552)"""
    rt.require_gpu(q)
    D, adt = H * d, rt.act_dtype()
    for t in (q, k, v, e):
        assert t.dtype == adt and t.is_contiguous()
    assert q.shape == (B * T, D) and k.shape == (B * Tk, D) and v.shape == (B * Tk, D) and e.shape == (Tk + T - 1, D)
    if mask is not None:
        mask = mask.reshape(mask.shape[0], T, Tk).float().contiguous()
    o = empty((B * T, D), adt, q)
    lse = empty((B * H, T, 2), torch.float32, q)
    a = Attn()
    a.q, a.k, a.v, a.ld = q.data_ptr(), k.data_ptr(), v.data_ptr(), D
    a.e, a.lde = e.data_ptr(), D
    a.lens, a.len_div = _p(lens), 1
    if mask is not None:
        a.mask, a.mask_bstride = mask.data_ptr(), (T * Tk if mask.shape[0] > 1 else 0)
    a.o, a.ldo, a.lse = o.data_ptr(), D, lse.data_ptr()
    a.B, a.H, a.T, a.d, a.scale, a.Tk = B, H, T, d, scale, Tk
    lib.relpos_attention_fwd(rt.dt(), _byref(a), rt.stream())
    if not want_probs:
        return o, None
    Tld = (Tk + 7) // 8 * 8
    scratch = empty((2, B * H, T, Tld), adt, q)
    dq = empty((B * T, D), adt, q)
    zero = torch.zeros((B * T, D), dtype=adt, device=q.device)
    dsrel = torch.zeros((H, B * T, (Tk + T - 1 + 7) // 8 * 8), dtype=adt, device=q.device)
    a.dout, a.dq, a.lddq = zero.data_ptr(), dq.data_ptr(), D
    a.pbuf, a.dsbuf, a.ldt = scratch.data_ptr(), scratch.data_ptr() + scratch[0].numel() * scratch.element_size(), Tld
    a.dsrel, a.ldr = dsrel.data_ptr(), dsrel.shape[2]
    lib.relpos_attention_bwd(rt.dt(), _byref(a), rt.stream())         # the row pass recomputes the probabilities from (max, sum) and stores them
    return o, scratch[0].view(B, H, T, Tld)[..., :Tk].float()


def _pool_mask(mask, T, P):
    """reference min-pool of the padded (B,1,T,T) mask (nnet/attentions.py:140-171,357-362) -- exact 0/1 index work on the host side API path"""
    pad = (P - T % P) % P
    m = torch.nn.functional.pad(mask.reshape(mask.shape[0], 1, T, T).float(), (0, pad, 0, pad), value=0.0)
    Tp = (T + pad) // P
    return m.reshape(m.shape[0], Tp, P, Tp, P).amin(dim=(2, 4))


# ============================================================================================
# ConvolutionModule (nnet/modules.py:341-385) fused with the block's conv residual (nnet/blocks.py:273-277,298)
#   y = R(x) + Drop(PW2(Swish(BN(DW(GLU(PW1(LN(x))))))))     R = identity | strided k=1 conv
# ============================================================================================
class BNState:
    """scratch for one BatchNorm application: stats [2C+1] (sum | sumsq | count) and ss [4C]"""

    NREP = 64      # AVEC_STAT_REPLICAS: the GEMM epilogue spreads its statistic atomics over this many [2C] copies

    def __init__(self, C, ref):
        self.stats = rt.zeros_scratch(self.NREP * 2 * C, ref.device)
        self.red = None
        self.ss = torch.empty(4 * C, dtype=torch.float32, device=ref.device)
        self.C = C


RELU_BITMASK = True    # ResNet block ends: the backward pass reads a 1-bit ReLU mask instead of the saved block output (0: reads `out`)
def bn_finalize(bn, st, count, training):
    """bn: module with weight/bias/running_mean/running_var/num_batches_tracked/momentum/eps"""
    C = st.C
    cptr = None
    sptr, nrep = st.stats.data_ptr(), st.NREP
    if training and rt.sync_batchnorm():
        # SyncBatchNorm: collapse the replicas, append the local count, sum over ranks (one small RCCL all-reduce)
        st.red = rt.sync_bn_stats(st.stats, st.NREP, C, count, key=(id(bn), "f"))
        sptr, nrep, cptr = st.red.data_ptr(), 1, st.red.data_ptr() + 2 * C * 4
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats and bn.running_mean is not None
    lib.bn_finalize(sptr, nrep, cptr, float(count), bn.weight.data_ptr(), bn.bias.data_ptr(),
                    bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                    bn.num_batches_tracked.data_ptr() if (track and training) else None, mom, bn.eps, st.ss.data_ptr(), C, int(training), rt.stream())
    return cptr


def bn_backward(bn, st, cptr, count, dout, y, out, act, M, want_dres=False, pre=None, mask=None):
    """pre: a completed BnbFuse -- `dout` is already masked and its (sum d, sum d*y) sit in pre.stats: no reduction pass, the apply pass reads d and y only
    mask: the ReLU mask of `out` as one bit per element (avec_bn_apply_fwd_mask): read instead of `out`"""
    C = st.C
    adt = rt.act_dtype()
    dstats = rt.zeros_scratch(2 * C, dout.device)
    if pre is not None:
        lib.bn_bwd_finalize(pre.stats.data_ptr(), BNState.NREP, st.ss.data_ptr(), dstats.data_ptr(), C, rt.stream())
        act, out, want_dres, mask = ACT_NONE, None, False, None
    elif mask is not None:
        lib.bn_bwd_reduce_mask(rt.dt(), dout.data_ptr(), y.data_ptr(), mask.data_ptr(), st.ss.data_ptr(), dstats.data_ptr(), M, C, rt.stream())
    else:
        lib.bn_bwd_reduce(rt.dt(), dout.data_ptr(), y.data_ptr(), _p(out), st.ss.data_ptr(), act, dstats.data_ptr(), M, C, rt.stream())
    gw, gb = grad_of(bn.weight), grad_of(bn.bias)
    dstats, synced = _add_local_affine_grads(dstats, gw, gb, C, (id(bn), "b"))
    if synced:
        gw = gb = None                    # (SyncBatchNorm) already added from the LOCAL sums; the kernel must not add the global ones
    dy = empty((M, C), adt, dout)
    dres = empty((M, C), adt, dout) if want_dres else None
    if mask is not None:
        lib.bn_bwd_apply_mask(rt.dt(), dout.data_ptr(), y.data_ptr(), mask.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cptr, float(count),
                              dy.data_ptr(), _p(dres), _p(gw), _p(gb), M, C, rt.stream())
    else:
        lib.bn_bwd_apply(rt.dt(), dout.data_ptr(), y.data_ptr(), _p(out), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cptr, float(count), act,
                         dy.data_ptr(), _p(dres), _p(gw), _p(gb), M, C, rt.stream())
    return dy, (dout if pre is not None else dres)


def _add_local_affine_grads(dstats, gw, gb, C, key):
    """SyncBatchNorm: d(gamma), d(beta) are the LOCAL sums (the gradient all-reduce averages them like any other parameter, as torch's
    SyncBatchNorm does); only the statistics entering dx are global.  Adds the local sums, then all-reduces `dstats`.
    Returns (dstats to use -- the global sums when synchronised --, whether it synchronised)."""
    if not rt.sync_batchnorm():
        return dstats, False
    from . import peer
    px = peer.active()
    if px is not None and dstats.is_cuda and dstats.is_contiguous() :
        return px.all_reduce_sum_fused(dstats, 1, 2 * C, None, key, dgamma=gw, dbeta=gb, C=C), True      # local affine gradients added inside the exchange kernel
    lib.bn_affine_grads(dstats.data_ptr(), gw.data_ptr(), gb.data_ptr(), C, rt.stream())
    return rt.all_reduce_small(dstats, key), True



def dw_pad_left(conv):
    """zero frames in front of the sequence for the depthwise conv (nnet/layers.py:137-156): "same" = K // 2, "causal" = K - 1"""
    K = conv.kernel_size[0]
    kind = getattr(conv, "padding_type", "same")
    if kind == "causal":
        return K - 1
    assert kind == "same", "the fused convolution module supports 'same' and 'causal' padding, got %r" % (kind,)
    return K // 2

CONVMOD_BN_FUSE = True      # conformer convolution module: BatchNorm finalize straight from the depthwise kernel's partial sums


class ConvModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, _anchor, mod, res_conv, drop_p, sid, training):
        """mod: ConvolutionModule (layers: 0 LN, 1 pw1, 3 dw, 4 BN, 6 pw2);  res_conv: strided k=1 Conv1d or None"""
        rt.require_gpu(x)
        B, T, D = x.shape
        ln, pw1, dw, bn, pw2 = mod.layers[0], mod.layers[1], mod.layers[3], mod.layers[4], mod.layers[6]
        Dp, K, stride = dw.weight.shape[0], dw.weight.shape[2], dw.stride[0]
        To = (T - 1) // stride + 1
        M, Mo, adt = B * T, B * To, rt.act_dtype()
        x2 = _f32c(x.reshape(M, D))
        h, mean, rstd = layernorm_fwd(x2, ln.weight, ln.bias, M, D, False, ln.eps)
        u = linear_fwd(h, pw1.weight, pw1.bias, M, in_f32=False, out_f32=False)
        c = empty((Mo, Dp), adt, x2)
        st = BNState(Dp, x2)
        use_batch = training and not getattr(bn, "frozen", False)
        if use_batch and not rt.sync_batchnorm() and CONVMOD_BN_FUSE:
            # local batch statistics: the finalize reads the depthwise kernel's column-reduction partials (one launch less in the block's dependent chain)
            track = bn.track_running_stats and bn.running_mean is not None
            lib.glu_dwconv_fwd_bn(rt.dt(), u.data_ptr(), dw.weight.data_ptr(), _p(dw.bias), c.data_ptr(), st.stats.data_ptr(), B, T, Dp, K, stride, dw_pad_left(dw),
                                  bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                  bn.num_batches_tracked.data_ptr() if track else None, bn.momentum if bn.momentum is not None else 0.1, bn.eps, st.ss.data_ptr(), rt.stream())
            cptr = None
        else:
            lib.glu_dwconv_fwd(rt.dt(), u.data_ptr(), dw.weight.data_ptr(), _p(dw.bias), c.data_ptr(), st.stats.data_ptr() if use_batch else None,
                               B, T, Dp, K, stride, dw_pad_left(dw), rt.stream())
            cptr = bn_finalize(bn, st, Mo, use_batch)
        a = empty((Mo, Dp), adt, x2)
        lib.bn_apply_fwd(rt.dt(), c.data_ptr(), st.ss.data_ptr(), None, ACT_SWISH, a.data_ptr(), Mo, Dp, rt.stream())
        if res_conv is not None:
            rs = res_conv.stride[0]
            R = linear_fwd(x2, res_conv.weight, res_conv.bias, Mo, in_f32=(adt != torch.float32), out_f32=True,
                           rows=rows_plain(D, To, T, rs) if rs > 1 else None)
        else:
            R = x2
        y = linear_fwd(a, pw2.weight, pw2.bias, Mo, in_f32=False, out_f32=True, drop_p=drop_p, sid=sid, res=R, alpha=1.0)
        ctx.saved = (x2, mean, rstd, h, u, c, a, st, cptr, use_batch, mod, res_conv, drop_p, sid, B, T, To, D, Dp, K, stride)
        ctx.prep_req = _prep_request(x)
        return _tag_prep(y.view(B, To, Dp), 1.0, drop_p, sid)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, h, u, c, a, st, cptr, use_batch, mod, res_conv, drop_p, sid, B, T, To, D, Dp, K, stride = ctx.saved
        ln, pw1, dw, bn, pw2 = mod.layers[0], mod.layers[1], mod.layers[3], mod.layers[4], mod.layers[6]
        M, Mo, adt = B * T, B * To, rt.act_dtype()
        dy = _f32c(dy.reshape(Mo, Dp))
        dacc = _take_prep(dy, Mo, Dp, 1.0, drop_p, sid)
        if dacc is None:
            dacc = grad_prep(dy, Mo, Dp, drop_p=drop_p, sid=sid)
        linear_bwd_weight(dacc, a, pw2.weight, Mo, bias=pw2.bias)
        da = linear_bwd_input(dacc, pw2.weight, Mo, out_f32=False)
        du = empty((M, 2 * Dp), adt, dy)
        if use_batch and CONVMOD_BN_FUSE and stride == 1 and not rt.sync_batchnorm():
            # local batch statistics, stride 1: the reduction pass, then the BatchNorm-backward arithmetic inside the depthwise kernel's staging pass (no apply launch, no dc tensor)
            dstats = rt.zeros_scratch(2 * Dp, da.device)
            lib.bn_bwd_reduce(rt.dt(), da.data_ptr(), c.data_ptr(), None, st.ss.data_ptr(), ACT_SWISH, dstats.data_ptr(), Mo, Dp, rt.stream())
            lib.dwconv_glu_bwd_bn(rt.dt(), da.data_ptr(), c.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), float(Mo), u.data_ptr(), dw.weight.data_ptr(),
                                  du.data_ptr(), grad_of(dw.weight).data_ptr(), None if dw.bias is None else grad_of(dw.bias).data_ptr(),
                                  grad_of(bn.weight).data_ptr(), grad_of(bn.bias).data_ptr(), B, T, Dp, K, dw_pad_left(dw), rt.stream())
        else:
            if use_batch:
                dc, _ = bn_backward(bn, st, cptr, Mo, da, c, None, ACT_SWISH, Mo)
            else:   # eval / frozen statistics: dc = da * swish'(pre) * scale  -> reuse the apply kernel with zero batch terms
                dc = _bn_eval_backward(bn, st, da, c, ACT_SWISH, Mo)
            lib.dwconv_glu_bwd(rt.dt(), dc.data_ptr(), u.data_ptr(), dw.weight.data_ptr(), du.data_ptr(), grad_of(dw.weight).data_ptr(),
                               None if dw.bias is None else grad_of(dw.bias).data_ptr(), B, T, Dp, K, stride, dw_pad_left(dw), rt.stream())
        linear_bwd_weight(du, h, pw1.weight, M, bias=pw1.bias)
        dh = linear_bwd_input(du, pw1.weight, M, out_f32=False)
        if res_conv is None:
            dx = layernorm_bwd(dh, False, x2, mean, rstd, ln.weight, ln.bias, M, D, dres=dy, prep=ctx.prep_req)
        else:
            rs = res_conv.stride[0]
            dx = layernorm_bwd(dh, False, x2, mean, rstd, ln.weight, ln.bias, M, D)
            dracc = grad_prep(dy, Mo, Dp)
            linear_bwd_weight(dracc, x2, res_conv.weight, Mo, q_f32=(adt != torch.float32), q_rows=rows_plain(D, To, T, rs) if rs > 1 else None, bias=res_conv.bias)
            dxr = linear_bwd_input(dracc, res_conv.weight, Mo, out_f32=True)
            lib.strided_rows_add(dx.data_ptr(), dxr.data_ptr(), B, T, To, D, rs, rt.stream())
        return dx.view(B, T, D), None, None, None, None, None, None


def _bn_eval_backward(bn, st, dout, y, act, M, out=None):
    """BatchNorm backward with fixed (running) statistics: dy = dr * scale; the batch-mean terms vanish (dstats = 0)."""
    C = st.C
    dstats = torch.zeros(2 * C, dtype=torch.float32, device=dout.device)
    dstats2 = torch.zeros(2 * C, dtype=torch.float32, device=dout.device)
    lib.bn_bwd_reduce(rt.dt(), dout.data_ptr(), y.data_ptr(), _p(out), st.ss.data_ptr(), act, dstats2.data_ptr(), M, C, rt.stream())
    dy = empty((M, C), rt.act_dtype(), dout)
    lib.bn_bwd_apply(rt.dt(), dout.data_ptr(), y.data_ptr(), _p(out), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), None, 1.0, act,
                     dy.data_ptr(), None, None, None, M, C, rt.stream())
    # parameter gradients from the true sums
    grad_of(bn.weight).add_(dstats2[C:])
    grad_of(bn.bias).add_(dstats2[:C])
    return dy


# ============================================================================================
# InterCTCResModule (nnet/modules.py:395-400):  logits = P1 x;  y = x + P2 softmax(logits)
# ============================================================================================
class InterCTCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        rt.require_gpu(x)
        B, T, D = x.shape
        M, V = B * T, w1.shape[0]
        x2 = _f32c(x.reshape(M, D))
        conv = rt.act_dtype() != torch.float32
        logits = linear_fwd(x2, w1, b1, M, in_f32=conv, out_f32=True)
        probs = empty((M, V), rt.act_dtype(), x2)
        lib.softmax_fwd(rt.dt(), logits.data_ptr(), probs.data_ptr(), M, V, rt.stream())
        y = linear_fwd(probs, w2, b2, M, in_f32=False, out_f32=True, res=x2, alpha=1.0)
        ctx.saved = (x2, logits, probs, w1, b1, w2, b2, B, T, D, V, conv)
        return y.view(B, T, D), logits.view(B, T, V)

    @staticmethod
    def backward(ctx, dy, dlogits_ext):
        x2, logits, probs, w1, b1, w2, b2, B, T, D, V, conv = ctx.saved
        M = B * T
        dy = _f32c(dy.reshape(M, D))
        dacc = grad_prep(dy, M, D)
        linear_bwd_weight(dacc, probs, w2, M, bias=b2)
        dprobs = linear_bwd_input(dacc, w2, M, out_f32=False)
        dl = empty((M, V), torch.float32, dy)
        dext = None if dlogits_ext is None else _f32c(dlogits_ext.reshape(M, V))
        lib.softmax_bwd(rt.dt(), dprobs.data_ptr(), logits.data_ptr(), dl.data_ptr(), _p(dext), M, V, rt.stream())
        dla = grad_prep(dl, M, V)
        linear_bwd_weight(dla, x2, w1, M, q_f32=conv, bias=b1)
        dx = linear_bwd_input(dla, w1, M, out_f32=True, res=dy)
        return dx.view(B, T, D), None, None, None, None


# ============================================================================================
# FusionModule (nnet/modules.py:421-426): cat -> Linear -> Swish -> Linear
# ============================================================================================
class FusionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, v, w1, b1, w2, b2):
        rt.require_gpu(a)
        B, T, Da = a.shape
        Dv = v.shape[-1]
        M, F, adt = B * T, w1.shape[0], rt.act_dtype()
        a2, v2 = _f32c(a.reshape(M, Da)), _f32c(v.reshape(M, Dv))
        xc = empty((M, Da + Dv), adt, a2)
        lib.cast_rows(rt.dt(), a2.data_ptr(), Da, xc.data_ptr(), Da + Dv, M, Da, rt.stream())
        lib.cast_rows(rt.dt(), v2.data_ptr(), Dv, xc.data_ptr() + Da * xc.element_size(), Da + Dv, M, Dv, rt.stream())
        z = empty((M, F), adt, a2)
        h = linear_fwd(xc, w1, b1, M, in_f32=False, out_f32=False, act=ACT_SWISH, out_pre=z)
        y = linear_fwd(h, w2, b2, M, in_f32=False, out_f32=True)
        ctx.saved = (xc, z, h, w1, b1, w2, b2, B, T, Da, Dv, F)
        return y.view(B, T, -1)

    @staticmethod
    def backward(ctx, dy):
        xc, z, h, w1, b1, w2, b2, B, T, Da, Dv, F = ctx.saved
        M, N = B * T, w2.shape[0]
        dy = _f32c(dy.reshape(M, N))
        dacc = grad_prep(dy, M, N)
        linear_bwd_weight(dacc, h, w2, M, bias=b2)
        dz = linear_bwd_input(dacc, w2, M, out_f32=False, dact_z=z, dact=ACT_SWISH)
        linear_bwd_weight(dz, xc, w1, M, bias=b1)
        sh = rt.shadow(w1)                     # bwd shadow [Da+Dv][F]: rows 0..Da-1 -> d(audio), rest -> d(video)
        da = empty((M, Da), torch.float32, dy)
        dv = empty((M, Dv), torch.float32, dy)
        gemm_nt(dz, sh.bwd, da, M, Da, F, out_f32=True)
        gemm_nt(dz, sh.bwd[Da * F:], dv, M, Dv, F, out_f32=True)
        return da.view(B, T, Da), dv.view(B, T, Dv), None, None, None, None


# ============================================================================================
# CTC loss (nnet/losses.py:311-334) -> batch mean
# ============================================================================================
class CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, logit_len, targets, target_len, blank, zero_infinity):
        rt.require_gpu(logits)
        B, T, V = logits.shape
        lg = _f32c(logits)
        tg = targets.to(device=lg.device, dtype=torch.int64).contiguous()
        if tg.dim() == 1:
            tg = tg.view(B, -1)
        Lmax = tg.shape[1]
        il = logit_len.to(device=lg.device, dtype=torch.int64).contiguous()
        tl = target_len.to(device=lg.device, dtype=torch.int64).contiguous()
        nll = empty((B,), torch.float32, lg)
        mean = torch.zeros((), dtype=torch.float32, device=lg.device)
        need_grad = ctx.needs_input_grad[0]
        grad = empty((B, T, V), torch.float32, lg) if need_grad else None
        ws = empty((lib.raw("avec_ctc_workspace_floats")(B, T, max(Lmax, 1)),), torch.float32, lg)
        lib.ctc_loss(lg.data_ptr(), il.data_ptr(), tg.data_ptr(), tl.data_ptr(), nll.data_ptr(), mean.data_ptr(), _p(grad), ws.data_ptr(),
                     B, T, V, Lmax, blank, int(zero_infinity), rt.stream())
        ctx.saved = (grad, B)
        ctx.nll = nll
        return mean

    @staticmethod
    def backward(ctx, dloss):
        grad, B = ctx.saved
        out = torch.empty_like(grad)
        dl = dloss.float().contiguous()
        lib.scale_by_scalar(grad.data_ptr(), dl.data_ptr(), 1.0 / B, out.data_ptr(), grad.numel(), rt.stream())
        return out, None, None, None, None, None


class CTCLossMultiFn(torch.autograd.Function):
    """n CTC heads over the same batch and labels in one launch (avec_ctc_loss_multi): apply(blank, zero_infinity, targets, target_len, weights, logits_0, len_0, ...)
    -> (sum_i weights[i] * loss_i, loss_0, ..., loss_{n-1}); the individual batch-mean losses are returned for logging (not differentiable), the weighted sum
    carries the gradient: backward is one scaling launch for all heads.  Same arithmetic per head as CTCLossFn (the kernel body is shared)."""

    @staticmethod
    def forward(ctx, blank, zero_infinity, targets, target_len, weights, *flat):
        n = len(flat) // 2
        logits, lens = [_f32c(t) for t in flat[0::2]], flat[1::2]
        rt.require_gpu(logits[0])
        B, _, V = logits[0].shape
        dev = logits[0].device
        tg = targets.to(device=dev, dtype=torch.int64).contiguous()
        if tg.dim() == 1:
            tg = tg.view(B, -1)
        Lmax = tg.shape[1]
        tl = target_len.to(device=dev, dtype=torch.int64).contiguous()
        ils = [l.to(device=dev, dtype=torch.int64).contiguous() for l in lens]
        nll = torch.empty((n, B), dtype=torch.float32, device=dev)
        means = rt.zeros_scratch(n + 1, dev)                 # [n]: the weighted total, accumulated by the same launch (was a rocBLAS dot + fill + add on the critical chain)
        grads = [empty(tuple(lg.shape), torch.float32, lg) if ctx.needs_input_grad[5 + 2 * i] else None for i, lg in enumerate(logits)]
        arr_p = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)
        Ts = (ctypes.c_int * n)(*[lg.shape[1] for lg in logits])
        lib.ctc_loss_multi(n, arr_p([lg.data_ptr() for lg in logits]), arr_p([l.data_ptr() for l in ils]), Ts, arr_p([nll[i].data_ptr() for i in range(n)]),
                           arr_p([means[i:].data_ptr() for i in range(n)]), arr_p([g.data_ptr() if g is not None else None for g in grads]),
                           tg.data_ptr(), tl.data_ptr(), (ctypes.c_float * n)(*[float(x) for x in weights]), means[n:].data_ptr(), B, V, Lmax, blank, int(zero_infinity), rt.stream())
        total = means[n]
        ctx.set_materialize_grads(False)                     # (the n logging outputs never receive a gradient: no zero-filled scalars made for them in backward)
        ctx.saved = (grads, B, n, [float(x) for x in weights])
        ctx.keep = (logits, ils, tg, tl, nll)
        outs = tuple(means[i] for i in range(n))
        ctx.mark_non_differentiable(*outs)
        return (total,) + outs

    @staticmethod
    def backward(ctx, dtotal, *_unused):
        grads, B, n, weights = ctx.saved
        out = [None, None, None, None, None]
        if dtotal is None:
            return tuple(out + [None, None] * n)
        dt = dtotal.float().contiguous()
        live = [i for i in range(n) if grads[i] is not None]
        outs = {i: torch.empty_like(grads[i]) for i in live}
        if live:                                             # one scaling launch for all heads
            k = len(live)
            lib.scale_by_scalar_multi(k, (ctypes.c_void_p * k)(*[grads[i].data_ptr() for i in live]), (ctypes.c_void_p * k)(*[outs[i].data_ptr() for i in live]),
                                      (ctypes.c_longlong * k)(*[grads[i].numel() for i in live]), (ctypes.c_float * k)(*[weights[i] / B for i in live]), dt.data_ptr(), rt.stream())
        for i in range(n):
            out += [outs.get(i), None]
        return tuple(out)


_WCACHE = {}


def ctc_multi_fits(T, V, Lmax):
    return bool(lib.raw("avec_ctc_loss_multi_fits")(int(T), int(V), int(Lmax)))


class TimeMeanFn(torch.autograd.Function):
    """mean over the time axis of fp32 (B, T, V) logits (VisualEfficientConformerCE.forward, nnet/models_zoo.py:41) = the channels-last average pool"""

    @staticmethod
    def forward(ctx, x):
        from .lib import F32
        rt.require_gpu(x)
        B, T, V = x.shape
        x = _f32c(x)
        y = empty((B, V), torch.float32, x)
        lib.avgpool_fwd(F32, x.data_ptr(), y.data_ptr(), B, T, V, rt.stream())
        ctx.saved = (B, T, V)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .lib import F32
        B, T, V = ctx.saved
        dy = _f32c(dy)
        dx = empty((B, T, V), torch.float32, dy)
        lib.avgpool_bwd(F32, dy.data_ptr(), dx.data_ptr(), B, T, V, rt.stream())
        return dx


class SoftmaxCEFn(torch.autograd.Function):
    """mean over rows of cross-entropy(logits[M,V], targets[M]) with ignore_index (losses.SoftmaxCrossEntropy, nnet/losses.py:258-290)"""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        rt.require_gpu(logits)
        M, V = logits.shape
        lg = _f32c(logits)
        tg = targets.to(device=lg.device, dtype=torch.int64).contiguous()
        loss = empty((M,), torch.float32, lg)
        mean = torch.zeros((), dtype=torch.float32, device=lg.device)
        grad = empty((M, V), torch.float32, lg) if ctx.needs_input_grad[0] else None
        lib.softmax_ce(lg.data_ptr(), tg.data_ptr(), int(ignore_index), loss.data_ptr(), mean.data_ptr(), _p(grad), M, V, rt.stream())
        ctx.saved = (grad, M)
        return mean

    @staticmethod
    def backward(ctx, dloss):
        grad, M = ctx.saved
        out = torch.empty_like(grad)
        lib.scale_by_scalar(grad.data_ptr(), dloss.float().contiguous().data_ptr(), 1.0 / M, out.data_ptr(), grad.numel(), rt.stream())
        return out, None, None


# ============================================================================================
# convolutions on channels-last activations (ResNet-18 front-end, nnet/blocks.py:29-91, nnet/networks.py:32-146)
# ============================================================================================
def conv2d_fwd(x, weight, N, H, W, Cin, stride, stats=None):
    """x: act [N,H,W,Cin] (contiguous NHWC); weight: Conv2d param (physical [Cout][KH][KW][Cin]); 'same' zero padding ((k-1)//2)."""
    Cout, KH, KW = weight.shape[0], weight.shape[2], weight.shape[3]
    pad = (KH - 1) // 2
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M = N * OH * OW
    sh = rt.shadow(weight)
    y = empty((M, Cout), rt.act_dtype(), x)
    if _slab_conv(H, W, Cin, Cout, KH, KW, stride):
        ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
        lib.conv3x3_c64(x.data_ptr(), sh.fwd.data_ptr(), y.data_ptr(), None, _p(stats), N, H, W, 0, rt.stream())
        if ev is not None:
            KERNEL_TIMER.stop(ev, (2, 0), 2.0 * M * Cout * KH * KW * Cin, 2.0 * (N * H * W * Cin + Cout * KH * KW * Cin + M * Cout))
        return y, OH, OW
    gemm_nt(x, sh.fwd, y, M, Cout, KH * KW * Cin, rows=rows_conv(H, W, Cin, KH, KW, stride, pad, OH, OW), mode=ROWS_CONV_FWD, stats=stats,
            abytes=2.0 * (N * H * W * Cin + Cout * KH * KW * Cin + M * Cout))
    return y, OH, OW


SLAB_CONV = True
SLAB_WGRAD128 = True
SHORTCUT_SUBGRID = True        # ResNet projection shortcuts: input gradient on the subsampled grid (res_cls0)
GROUP_WGRAD128 = True          # the wide layers' weight gradients as one grouped launch at the end of the backward pass


def _slab_conv(H, W, Cin, Cout, KH, KW, stride):
    """ResNet stage 1 (3x3, stride 1, 64 -> 64 channels, 22x22 images), bf16: weights resident in LDS + one image slab per iteration (csrc/conv3x3.hip)"""
    return SLAB_CONV and rt.act_dtype() == torch.bfloat16 and bool(lib.raw("avec_conv3x3_c64_supported")(H, W, Cin, Cout, KH, KW, stride))


class BnbFuse:
    """request to fold a BatchNorm(+ReLU) backward reduction into the epilogue of the product that computes its output gradient:
    y = the BatchNorm input, mask_z = saved post-ReLU activation (mask z > 0) or None (mask from scale*y + shift > 0 with `ss`), stats = replicated [sum d | sum d*y]"""
    __slots__ = ("y", "ss", "mask_z", "stats", "done")

    def __init__(self, y, ss, mask_z, C):
        self.y, self.ss, self.mask_z, self.done = y, ss, mask_z, False
        self.stats = rt.zeros_scratch(BNState.NREP * 2 * C, y.device)


# BatchNorm-backward reductions inside the backward-data epilogues (ResNet stages 2-4).  Correct (tests/test_gpu_round3.py) but SLOWER in the step (25.32 vs 25.07 ms, same box):
# the extra y / mask tile loads sit on the critical path of each workgroup's epilogue while the stand-alone reductions run at 0.5 of the HBM peak.  Opt-in.
BNB_FUSE = False


def conv2d_bwd(dy, x, weight, N, H, W, Cin, stride, OH, OW, need_dx=True, dx_res=None, bnb=None, dx_res_cls0=False, dx_res_mask=None):
    """dx_res_cls0: dx_res holds rows for the (even row, even column) input pixels only (stride-2 layers, bf16: include/avec_hip.h res_cls0)
    dx_res_mask: one bit per element of dx_res (the ReLU mask written by avec_bn_apply_fwd_mask): dx = conv^T(dy) + (bit ? dx_res : 0) (see res_mask_ok)"""
    Cout, KH, KW = weight.shape[0], weight.shape[2], weight.shape[3]
    pad = (KH - 1) // 2
    M = N * OH * OW
    sh = rt.shadow(weight)
    if _slab_conv(H, W, Cin, Cout, KH, KW, stride) and H * (W + 1) <= 512:
        if GROUP_WGRAD128 and _in_backward():
            it = WgradItem()
            it.x, it.dy, it.dw, it.images, it.C, it.H, it.W = x.data_ptr(), dy.data_ptr(), grad_of(weight).data_ptr(), N, Cin, H, W
            q = _pending()
            q.cw64.append(it)
            q.keep += [x, dy]
            q.cw64_flops += 2.0 * M * Cout * KH * KW * Cin
        else:
            ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
            lib.wgrad3x3_c64(x.data_ptr(), dy.data_ptr(), grad_of(weight).data_ptr(), N, H, W, rt.stream())
            if ev is not None:
                KERNEL_TIMER.stop(ev, (2, 2), 2.0 * M * Cout * KH * KW * Cin)
    elif SLAB_CONV and SLAB_WGRAD128 and rt.act_dtype() == torch.bfloat16 and bool(lib.raw("avec_wgrad3x3_c128_supported")(H, W, Cin, Cout, KH, KW, stride)):
        if GROUP_WGRAD128 and _in_backward():
            # queued: one grouped launch for all the wide layers when the backward pass is through (the final atomics of a launch of its own are ~30 % of it)
            it = WgradItem()
            it.x, it.dy, it.dw, it.images, it.C, it.H, it.W = x.data_ptr(), dy.data_ptr(), grad_of(weight).data_ptr(), N, Cin, H, W
            q = _pending()
            q.cw.append(it)
            q.keep += [x, dy]
            q.cw_flops += 2.0 * M * Cout * KH * KW * Cin
        else:
            ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
            lib.wgrad3x3_c128(x.data_ptr(), dy.data_ptr(), grad_of(weight).data_ptr(), N, Cin, H, W, rt.stream())
            if ev is not None:
                KERNEL_TIMER.stop(ev, (2, 2), 2.0 * M * Cout * KH * KW * Cin)
    else:
        gemm_tn(dy, x, grad_of(weight), M, Cout, KH * KW * Cin, q_rows=rows_conv(H, W, Cin, KH, KW, stride, pad, OH, OW), q_mode=ROWS_CONV_FWD, side=True)
    if not need_dx:
        return None
    dx = empty((N * H * W, Cin), rt.act_dtype(), dy)
    if _slab_conv(H, W, Cin, Cout, KH, KW, stride):
        ev = KERNEL_TIMER.start() if KERNEL_TIMER.enabled else None
        if dx_res_mask is not None:
            lib.conv3x3_c64_res_masked(dy.data_ptr(), sh.bwd.data_ptr(), dx.data_ptr(), dx_res.data_ptr(), dx_res_mask.data_ptr(), N, H, W, 1, rt.stream())
        else:
            lib.conv3x3_c64(dy.data_ptr(), sh.bwd.data_ptr(), dx.data_ptr(), _p(dx_res), None, N, H, W, 1, rt.stream())
        if ev is not None:
            KERNEL_TIMER.stop(ev, (2, 1), 2.0 * N * H * W * Cin * KH * KW * Cout, 2.0 * (2 * N * H * W * Cin + Cout * KH * KW * Cin + M * Cout))
        return dx
    # algorithmic work of the backward-data product: one MAC per (output pixel, tap, Cin, Cout) -- for stride 2 three of four taps of the implicit GEMM
    # over input pixels are structurally zero and are skipped by the parity-class kernel: they are not counted
    fuse = bnb if (bnb is not None and BNB_FUSE and rt.act_dtype() == torch.bfloat16 and Cin % 4 == 0) else None
    gemm_nt(dy, sh.bwd, dx, N * H * W, Cin, KH * KW * Cout, rows=rows_conv(H, W, Cout, KH, KW, stride, pad, OH, OW), mode=ROWS_CONV_BWD,
            res=dx_res, res_act=True, flops=2.0 * N * OH * OW * Cout * KH * KW * Cin, bnb=fuse, res_cls0=dx_res_cls0, res_mask=dx_res_mask,
            abytes=2.0 * (N * H * W * Cin * (2 if dx_res is not None else 1) + Cout * KH * KW * Cin + M * Cout))
    if fuse is not None:
        fuse.done = True
    return dx


RES_MASK = True      # identity-residual ResNet blocks: the block-output gradient enters the first convolution's backward-data epilogue through the 1-bit ReLU mask (no masked copy of it is written)


def res_mask_ok(H, W, Cin, Cout, stride):
    """can conv2d_bwd add a bit-masked residual for this 3x3 layer?  The stage-1 slab kernel, or the shifted-window kernel with its register-direct epilogue (csrc/gemm.hip)"""
    if rt.act_dtype() != torch.bfloat16 or stride != 1 or Cin != Cout:
        return False
    return _slab_conv(H, W, Cin, Cout, 3, 3, 1) or (Cin >= 128 and Cin % 32 == 0 and W <= 31)


class ResNetBlockFn(torch.autograd.Function):
    """x: act NHWC [N,H,W,Cin] -> act NHWC [N,OH,OW,Cout]"""

    # BatchNorm-backward fusion across blocks: a block whose output feeds ONLY the next block (chain=True, set by nnet.ResNet) registers (y2, out, ss2) under its
    # output's address; the next block's backward folds the mask + (sum d, sum d*y2) reduction of that BatchNorm into the epilogue of the product that computes its
    # input gradient and leaves the completed request under the gradient's address for the producer block's backward
    _CHAIN, _READY = {}, {}

    @staticmethod
    def forward(ctx, x, _anchor, blk, training, chain=False):
        rt.require_gpu(x)
        N, H, W, Cin = x.shape
        conv1, bn1, conv2, bn2 = blk.layers[0], blk.layers[1], blk.layers[3], blk.layers[4]
        stride = conv1.stride[0]
        Cout = conv1.weight.shape[0]
        adt = rt.act_dtype()
        x = x.contiguous()
        st1, st2 = BNState(Cout, x), BNState(Cout, x)
        y1, OH, OW = conv2d_fwd(x, conv1.weight, N, H, W, Cin, stride, stats=st1.stats if training else None)
        Mo = N * OH * OW
        c1 = bn_finalize(bn1, st1, Mo, training)
        a1 = empty((Mo, Cout), adt, x)
        lib.bn_apply_fwd(rt.dt(), y1.data_ptr(), st1.ss.data_ptr(), None, ACT_RELU, a1.data_ptr(), Mo, Cout, rt.stream())
        y2, _, _ = conv2d_fwd(a1, conv2.weight, N, OH, OW, Cout, 1, stats=st2.stats if training else None)
        c2 = bn_finalize(bn2, st2, Mo, training)
        has_proj = not isinstance(blk.residual, torch.nn.Identity)
        yr = str_ = cr = None
        if has_proj:
            convr, bnr = blk.residual[0], blk.residual[1]
            str_ = BNState(Cout, x)
            yr, _, _ = conv2d_fwd(x, convr.weight, N, H, W, Cin, stride, stats=str_.stats if training else None)
            cr = bn_finalize(bnr, str_, Mo, training)
        use_mask = RELU_BITMASK and training and Cout % 8 == 0 and Cout <= 2048 and Mo * Cout // 8 < (1 << 31)
        if has_proj and not use_mask:
            r = empty((Mo, Cout), adt, x)
            lib.bn_apply_fwd(rt.dt(), yr.data_ptr(), str_.ss.data_ptr(), None, ACT_NONE, r.data_ptr(), Mo, Cout, rt.stream())
        elif not has_proj:
            r = x
        out = empty((Mo, Cout), adt, x)
        ctx.relu_mask = None
        if use_mask:
            # the backward pass needs `out` only as a ReLU mask: one bit per element, written here, instead of re-reading the tensor twice (reduction + apply pass);
            # a projection shortcut is normalised on the fly from its raw convolution output (no tensor, no launch of its own)
            ctx.relu_mask = torch.empty(Mo * Cout // 8, dtype=torch.uint8, device=x.device)
            lib.bn_apply_fwd_mask(rt.dt(), y2.data_ptr(), st2.ss.data_ptr(), (yr if has_proj else r).data_ptr(), str_.ss.data_ptr() if has_proj else None, out.data_ptr(),
                                  ctx.relu_mask.data_ptr(), Mo, Cout, rt.stream())
        else:
            lib.bn_apply_fwd(rt.dt(), y2.data_ptr(), st2.ss.data_ptr(), r.data_ptr(), ACT_RELU, out.data_ptr(), Mo, Cout, rt.stream())
        ctx.saved = (x, y1, a1, y2, yr, out, st1, st2, str_, c1, c2, cr, blk, training, N, H, W, Cin, Cout, OH, OW, stride, has_proj)
        ctx.chain = bool(chain and training and BNB_FUSE and rt.act_dtype() == torch.bfloat16)
        if ctx.chain:
            if len(ResNetBlockFn._CHAIN) > 64:
                ResNetBlockFn._CHAIN.clear()
            ResNetBlockFn._CHAIN[out.data_ptr()] = (y2, out, st2, Cout)
        return out.view(N, OH, OW, Cout)

    @staticmethod
    def backward(ctx, dout):
        x, y1, a1, y2, yr, out, st1, st2, str_, c1, c2, cr, blk, training, N, H, W, Cin, Cout, OH, OW, stride, has_proj = ctx.saved
        conv1, bn1, conv2, bn2 = blk.layers[0], blk.layers[1], blk.layers[3], blk.layers[4]
        Mo = N * OH * OW
        dout = dout.reshape(Mo, Cout).to(rt.act_dtype()).contiguous()
        assert training, "ResNetBlock backward is implemented for training-mode BatchNorm"
        ResNetBlockFn._CHAIN.pop(out.data_ptr(), None)
        pre2 = ResNetBlockFn._READY.pop(dout.data_ptr(), None) if ctx.chain else None      # the consumer block already masked dout and reduced it against y2
        if pre2 is not None and pre2.y is not y2:
            raise RuntimeError("ResNetBlock backward: a fused BatchNorm-backward request does not belong to this block (the block output has another consumer?)")
        # the gradient that passed the block's final ReLU is needed twice more (second BatchNorm's backward above all, then as the residual branch's gradient): with the bit
        # mask at hand nobody needs it as a tensor -- the consumers below apply the mask themselves
        lazy_dres = (RES_MASK and not BNB_FUSE and ctx.relu_mask is not None and pre2 is None and
                     (has_proj or (conv1.weight.shape[2] == 3 and res_mask_ok(H, W, Cin, Cout, stride) and not KERNEL_TIMER.enabled)))
        dy2, dres = bn_backward(bn2, st2, c2, Mo, dout, y2, out, ACT_RELU, Mo, want_dres=not lazy_dres, pre=pre2, mask=ctx.relu_mask)
        f1 = BnbFuse(y1, st1.ss, None, Cout)        # BatchNorm 1 + ReLU: the mask comes from the pre-activation itself
        da1 = conv2d_bwd(dy2, a1, conv2.weight, N, OH, OW, Cout, 1, OH, OW, bnb=f1)
        dy1, _ = bn_backward(bn1, st1, c1, Mo, da1, y1, None, ACT_RELU, Mo, pre=f1 if f1.done else None)      # no residual before this ReLU: the mask is recomputed from y1 (one tensor less to read)
        need_dx = ctx.needs_input_grad[0]
        prev = ResNetBlockFn._CHAIN.get(x.data_ptr()) if need_dx else None
        fx = BnbFuse(prev[0], prev[2].ss, prev[1], prev[3]) if prev is not None else None
        if has_proj:
            convr, bnr = blk.residual[0], blk.residual[1]
            if lazy_dres:
                dyr, _ = bn_backward(bnr, str_, cr, Mo, dout, yr, None, ACT_NONE, Mo, mask=ctx.relu_mask)      # (mask variant: d = bit ? dout : 0, no activation of its own)
            else:
                dyr, _ = bn_backward(bnr, str_, cr, Mo, dres, yr, None, ACT_NONE, Mo)
            kr = convr.weight.shape[2]
            if (SHORTCUT_SUBGRID and need_dx and stride == 2 and kr == 1 and rt.act_dtype() == torch.bfloat16 and Cin % 64 == 0 and Cout % 64 == 0
                    and conv1.weight.shape[2] == 3 and not _slab_conv(H, W, Cin, Cout, 3, 3, stride)):
                # the 1x1 / stride-2 shortcut sends its gradient to the (even, even) input pixels only: a plain product on the subsampled grid, added by the 3x3
                # layer's backward-data epilogue to its class-0 tiles -- instead of a full-size, three-quarters-zero tensor written and read back
                conv2d_bwd(dyr, x, convr.weight, N, H, W, Cin, stride, OH, OW, need_dx=False)
                dxs = empty((Mo, Cin), rt.act_dtype(), dyr)
                gemm_nt(dyr, rt.shadow(convr.weight).bwd, dxs, Mo, Cin, Cout)
                dx = conv2d_bwd(dy1, x, conv1.weight, N, H, W, Cin, stride, OH, OW, need_dx=True, dx_res=dxs, bnb=fx, dx_res_cls0=True)
            else:
                dx = conv2d_bwd(dyr, x, convr.weight, N, H, W, Cin, stride, OH, OW, need_dx=need_dx)
                dx = conv2d_bwd(dy1, x, conv1.weight, N, H, W, Cin, stride, OH, OW, need_dx=need_dx, dx_res=dx, bnb=fx)
        else:
            if lazy_dres:
                dx = conv2d_bwd(dy1, x, conv1.weight, N, H, W, Cin, stride, OH, OW, need_dx=need_dx, dx_res=dout, dx_res_mask=ctx.relu_mask, bnb=fx)
            else:
                dx = conv2d_bwd(dy1, x, conv1.weight, N, H, W, Cin, stride, OH, OW, need_dx=need_dx, dx_res=dres, bnb=fx)
        if fx is not None and fx.done:
            if len(ResNetBlockFn._READY) > 64:
                ResNetBlockFn._READY.clear()
            ResNetBlockFn._READY[dx.data_ptr()] = fx
        return (dx.view(N, H, W, Cin) if dx is not None else None), None, None, None, None


STEM3D_DIRECT = True      # direct (no im2col) bf16 visual-stem kernels
STEM3P_FUSED_WGRAD = True  # ... and the weight gradient computed from the recomputed tiles in the same kernel (no dz tensor)
STEM_GRAD_CLONE = False   # the stem's backward masks the incoming gradient IN PLACE (it is the first ResNet block's dx: 99 MB, nothing else reads it); 1 = work on a copy
STEM3P = True                    # ... with the max pool inside the convolution kernel and the pre-pool tensor recomputed in backward (stem3p.hip)


class VideoStemFn(torch.autograd.Function):
    """Conv3d(1->C,(5,7,7),s(1,2,2),'same',bias) + BatchNorm3d + ReLU + MaxPool3d((1,3,3),s(1,2,2),'same')
    video fp32 [B,T,H,W] -> act NHWC [B*T, H/4, W/4, C]      (nnet/networks.py:459-470)"""

    @staticmethod
    def forward(ctx, video, _anchor, conv, bn, training):
        rt.require_gpu(video)
        B, T, H, W = video.shape
        C = conv.weight.shape[0]
        v = _f32c(video)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        M = B * T * OH * OW
        sh = rt.shadow(conv.weight)
        st = BNState(C, v)
        K, Kp = sh.Tm * sh.C, sh.Cp
        y = empty((M, C), rt.act_dtype(), v)
        if (STEM3P and STEM3D_DIRECT and rt.compute_dtype() == "bf16" and C == 64 and K == 245 and lib.raw("avec_stem3p_supported")(B, T, H, W)
                and lib.raw("avec_stem3d_supported")(B, T, H, W)):
            # round 3 (stem3p.hip): the max pool runs on the raw conv output inside the convolution kernel (BatchNorm + ReLU is monotone per channel); the
            # 793 MB pre-pool tensor never reaches memory, the backward pass recomputes it
            w8 = torch.zeros((C, 36, 8), dtype=sh.fwd.dtype, device=v.device)         # (kd,kh) rows: zero slot, then the 7 taps
            w8[:, :35, 1:] = sh.fwd.view(C, Kp)[:, :K].view(C, 35, 7)
            vb = v.to(torch.bfloat16)
            PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
            zp = torch.empty((B * T, PH, PW, C), dtype=torch.bfloat16, device=v.device)
            idx = torch.empty((B * T, PH, PW, C), dtype=torch.uint8, device=v.device)
            lib.stem3p_fwd(vb.data_ptr(), w8.data_ptr(), _p(conv.bias), bn.weight.data_ptr(), zp.data_ptr(), idx.data_ptr(), st.stats.data_ptr() if training else None,
                           B, T, H, W, rt.stream())
            cp = bn_finalize(bn, st, M, training)
            out = torch.empty_like(zp)
            lib.bn_apply_fwd(rt.dt(), zp.data_ptr(), st.ss.data_ptr(), None, ACT_RELU, out.data_ptr(), B * T * PH * PW, C, rt.stream())
            ctx.saved = (v, None, idx, st, cp, conv, bn, ("p", vb, w8, zp, PH, PW), B, T, OH, OW, C, M, K, training, None)
            return out
        if STEM3D_DIRECT and rt.compute_dtype() == "bf16" and C == 64 and K == 245 and lib.raw("avec_stem3d_supported")(B, T, H, W):
            # direct kernels (stem3d.hip): the input band is staged in LDS, no im2col matrix in HBM
            w8 = torch.zeros((C, 36, 8), dtype=sh.fwd.dtype, device=v.device)         # (kd,kh) rows of 7 taps + a zero slot (stem3d.hip)
            w8[:, :35, :7] = sh.fwd.view(C, Kp)[:, :K].view(C, 35, 7)
            lib.stem3d_fwd(v.data_ptr(), w8.data_ptr(), 288, _p(conv.bias), y.data_ptr(), st.stats.data_ptr() if training else None, B, T, H, W, rt.stream())
            r = None
        else:
            # im2col once (shared by the forward GEMM and the weight-gradient GEMM), then a plain MFMA GEMM with K padded to Kp
            A = empty((M, Kp), rt.act_dtype(), v)
            lib.stem_im2col(rt.dt(), v.data_ptr(), A.data_ptr(), B, T, H, W, Kp, rt.stream())
            gemm_nt(A, sh.fwd, y, M, C, Kp, bias=conv.bias, stats=st.stats if training else None)
            r = A
        cp = bn_finalize(bn, st, M, training)
        PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
        out = empty((B * T, PH, PW, C), rt.act_dtype(), v)
        idx = torch.empty((B * T, PH, PW, C), dtype=torch.uint8, device=v.device)
        ymax = torch.empty_like(out) if (training and C % 8 == 0) else None           # winners' pre-BN values: the backward statistics pass then reads pooled-size tensors only
        lib.stem_pool_fwd(rt.dt(), y.data_ptr(), st.ss.data_ptr(), out.data_ptr(), idx.data_ptr(), _p(ymax), B * T, OH, OW, C, rt.stream())
        ctx.saved = (v, y, idx, st, cp, conv, bn, r, B, T, OH, OW, C, M, K, training, ymax)
        return out

    @staticmethod
    def backward(ctx, dpool):
        v, y, idx, st, cp, conv, bn, r, B, T, OH, OW, C, M, K, training, ymax = ctx.saved
        assert training, "VideoStem backward is implemented for training-mode BatchNorm"
        dpool_in = dpool
        dpool = dpool.to(rt.act_dtype()).contiguous()
        if isinstance(r, tuple) and dpool.data_ptr() == dpool_in.data_ptr() and STEM_GRAD_CLONE:
            dpool = dpool.clone()                               # stem3p_reduce masks it in place (AVEC_STEM_GRAD_CLONE=1: for code that inspects the stem output's gradient through hooks / retain_grad)
        if isinstance(r, tuple):                                # stem3p: ReLU mask + BatchNorm-backward sums over the pooled tensor, then dz from the RECOMPUTED conv output
            _, vb, w8, zp, PH, PW = r
            H, W = v.shape[2], v.shape[3]
            dstats = torch.zeros(2 * C, dtype=torch.float32, device=v.device)
            lib.stem3p_reduce(dpool.data_ptr(), zp.data_ptr(), st.ss.data_ptr(), dstats.data_ptr(), B * T, PH, PW, rt.stream())      # (dpool is masked in place)
            gw, gb = grad_of(bn.weight), grad_of(bn.bias)
            dstats, synced = _add_local_affine_grads(dstats, gw, gb, C, (id(bn), "b"))
            if synced:
                gw = gb = None
            if STEM3P_FUSED_WGRAD and lib.raw("avec_stem3p_wgrad_supported")(B, T, H, W):
                # recompute + routing + weight gradient in ONE kernel: neither z nor dz exist in memory
                lib.stem3p_wgrad(vb.data_ptr(), w8.data_ptr(), _p(conv.bias), dpool.data_ptr(), idx.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cp, float(M),
                                 grad_of(conv.weight).data_ptr(), _p(gw), _p(gb), B, T, H, W, rt.stream())
            else:
                dy = empty((M, C), rt.act_dtype(), v)
                lib.stem3p_dz(vb.data_ptr(), w8.data_ptr(), _p(conv.bias), dpool.data_ptr(), idx.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cp, float(M),
                              dy.data_ptr(), _p(gw), _p(gb), B, T, H, W, rt.stream())
                lib.stem3d_wgrad(v.data_ptr(), dy.data_ptr(), grad_of(conv.weight).data_ptr(), B, T, H, W, rt.stream())
            if conv.bias is not None:
                grad_of(conv.bias)
            stamp("v_stem_end:b")
            return None, None, None, None, None
        dstats = torch.zeros(2 * C, dtype=torch.float32, device=v.device)
        args = (dpool.data_ptr(), idx.data_ptr(), y.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cp, float(M))
        if ymax is not None:
            lib.stem_pool_bwd_reduce_pooled(rt.dt(), dpool.data_ptr(), idx.data_ptr(), ymax.data_ptr(), st.ss.data_ptr(), dstats.data_ptr(), B * T, OH, OW, C, rt.stream())
        else:
            lib.stem_pool_bwd(rt.dt(), *args, 0, None, None, None, B * T, OH, OW, C, rt.stream())
        gw, gb = grad_of(bn.weight), grad_of(bn.bias)
        dstats, synced = _add_local_affine_grads(dstats, gw, gb, C, (id(bn), "b"))
        if synced:
            gw = gb = None
            args = args[:5] + (dstats.data_ptr(),) + args[6:]
        dy = empty((M, C), rt.act_dtype(), v)
        lib.stem_pool_bwd(rt.dt(), *args, 1, dy.data_ptr(), _p(gw), _p(gb), B * T, OH, OW, C, rt.stream())
        if r is None:
            H, W = v.shape[2], v.shape[3]
            lib.stem3d_wgrad(v.data_ptr(), dy.data_ptr(), grad_of(conv.weight).data_ptr(), B, T, H, W, rt.stream())
        else:
            gemm_tn(dy, r, grad_of(conv.weight), M, C, K, q_rows=rows_plain(r.shape[1]), side=True)
        if conv.bias is not None:
            grad_of(conv.bias)   # d(bias) before training-mode BatchNorm is analytically zero: left at 0
        return None, None, None, None, None


class AvgPoolFn(torch.autograd.Function):
    """GlobalAvgPool2d on act NHWC [N,H,W,C] -> act [N,C]"""

    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        x = x.contiguous()
        y = empty((N, C), x.dtype, x)
        lib.avgpool_fwd(rt.dt(), x.data_ptr(), y.data_ptr(), N, H * W, C, rt.stream())
        ctx.saved = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.saved
        dy = dy.to(rt.act_dtype()).contiguous()
        dx = empty((N, H, W, C), dy.dtype, dy)
        lib.avgpool_bwd(rt.dt(), dy.data_ptr(), dx.data_ptr(), N, H * W, C, rt.stream())
        return dx


class AudioStemFn(torch.autograd.Function):
    """Conv2d(1->C,3x3,s2,'same') + BatchNorm2d + Swish on mel [B,80,F] -> act [B, T', C*40]   (nnet/networks.py:359-377)"""

    @staticmethod
    def forward(ctx, mel, _anchor, conv, bn, training):
        rt.require_gpu(mel)
        B, NM, F = mel.shape
        C = conv.weight.shape[0]
        mel = _f32c(mel)
        Fo, To = (NM - 1) // 2 + 1, (F - 1) // 2 + 1
        adt = rt.act_dtype()
        y = empty((B * To, C * Fo), adt, mel)
        st = BNState(C, mel)
        lib.audio_stem_conv_fwd(rt.dt(), mel.data_ptr(), conv.weight.data_ptr(), _p(conv.bias), y.data_ptr(), st.stats.data_ptr() if training else None,
                                B, NM, F, C, rt.stream())
        count = B * To * Fo
        cp = bn_finalize(bn, st, count, training)
        a = empty((B, To, C * Fo), adt, mel)
        lib.audio_stem_act_fwd(rt.dt(), y.data_ptr(), st.ss.data_ptr(), a.data_ptr(), B, NM, F, C, rt.stream())
        ctx.saved = (mel, y, st, cp, count, conv, bn, B, NM, F, C, training)
        return a

    @staticmethod
    def backward(ctx, da):
        mel, y, st, cp, count, conv, bn, B, NM, F, C, training = ctx.saved
        assert training, "AudioStem backward is implemented for training-mode BatchNorm"
        flush_param_grads()                                   # last node of the audio branch: its queued parameter gradients go out on this (the branch's) stream
        da = da.to(rt.act_dtype()).contiguous()
        dstats = torch.zeros(2 * C, dtype=torch.float32, device=mel.device)
        base = (da.data_ptr(), y.data_ptr(), mel.data_ptr(), st.ss.data_ptr(), bn.weight.data_ptr(), dstats.data_ptr(), cp, float(count))
        lib.audio_stem_bwd(rt.dt(), *base, 0, None, None, None, None, B, NM, F, C, rt.stream())
        gw, gb = grad_of(bn.weight), grad_of(bn.bias)
        dstats, synced = _add_local_affine_grads(dstats, gw, gb, C, (id(bn), "b"))
        if synced:
            gw = gb = None
            base = base[:5] + (dstats.data_ptr(),) + base[6:]
        lib.audio_stem_bwd(rt.dt(), *base, 1, grad_of(conv.weight).data_ptr(), None if conv.bias is None else grad_of(conv.bias).data_ptr(),
                           _p(gw), _p(gb), B, NM, F, C, rt.stream())
        stamp("a_stem_end:b")
        arena = rt.arena_of(bn)
        if arena is not None and getattr(arena, "_early_armed", False) and getattr(arena, "_audio_range", None):
            lo, hi = arena._audio_range                       # the stem is the first op of the audio encoder: its backward is the last one of that branch
            arena._audio_range = None
            arena.early_all_reduce(lo, hi)
        return None, None, None, None, None


# ============================================================================================
# mel front-end (no gradient): AudioPreprocessing.forward (nnet/preprocessing.py:57-85)
# ============================================================================================
def mel_spectrogram(audio, window, dft, fb, n_fft, win, hop, n_mels):
    """audio fp32 [B,L]; window [win]; dft fp32 [2*nb][win] (cos rows then -sin rows); fb [nb][n_mels] -> [B,n_mels,L//hop+1] fp32"""
    rt.require_gpu(audio)
    from .lib import F32
    B, L = audio.shape
    audio = _f32c(audio)
    F = L // hop + 1
    nb = n_fft // 2 + 1
    frames = empty((B * F, win), torch.float32, audio)
    lib.mel_frames(audio.data_ptr(), window.data_ptr(), frames.data_ptr(), B, L, n_fft, win, hop, rt.stream())
    spec = empty((B * F, 2 * nb), torch.float32, audio)
    gemm_nt(frames, dft, spec, B * F, 2 * nb, win, out_f32=True, dtype=F32)
    out = empty((B, n_mels, F), torch.float32, audio)
    lib.mel_power_log(spec.data_ptr(), fb.data_ptr(), out.data_ptr(), B, F, nb, n_mels, rt.stream())
    return out


def spec_augment_(mel, lens, mF, Fp, mT, pS, sid):
    B, NM, F = mel.shape
    lens_ = None if lens is None else lens.to(device=mel.device, dtype=torch.int64).contiguous()
    lib.specaugment(mel.data_ptr(), _p(lens_), B, NM, F, mF, Fp, mT, pS, rt.rng_state(mel.device).data_ptr(), sid, rt.stream())
    return mel


def len_affine(lengths, sub, div, add):
    """floor((lengths - sub) / div) + add on an int64 length vector: one launch on the device (three ATen launches otherwise), plain torch on the host"""
    if torch.is_tensor(lengths) and lengths.is_cuda and lengths.dtype == torch.int64 and lengths.is_contiguous() and lengths.numel() > 0:
        out = torch.empty_like(lengths)
        lib.len_affine(lengths.data_ptr(), out.data_ptr(), lengths.numel(), sub, div, add, rt.stream())
        return out
    return torch.div(lengths - sub, div, rounding_mode="floor") + add


def argmax_rows(logits):
    lead, V = logits.shape[:-1], logits.shape[-1]
    lg = _f32c(logits)
    out = torch.empty(lead, dtype=torch.int64, device=lg.device)
    lib.argmax_rows(lg.data_ptr(), out.data_ptr(), out.numel(), V, rt.stream())
    return out
