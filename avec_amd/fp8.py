"""fp8 (OCP e4m3) operands for the forward Linear products -- BASELINE config 5's "fp8 GEMMs" (FFN, Q|K|V, output and pointwise-convolution projections).

Opt-in (`avec_amd.fp8.enable(True)` or AVEC_FP8=1) on top of the bf16 compute dtype; the backward pass, the residual stream, norms, softmax and losses are
unchanged (bf16 operands / fp32 accumulation as before), so the saved activations stay bf16 and only the forward products read e4m3:

    activation (bf16) --avec_fp8_quantize--> e4m3 [M][K]  \
                                                            avec_gemm_nt_fp8 (v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate, same fused epilogue)
    fp32 master weight --avec_fp8_weights_refresh--> e4m3 /

Scaling is per tensor and "current": max|x| of the very tensor being quantized (an absolute-maximum pass in front of the quantizer), weights once per
optimizer step for all matrices in two launches.  Eligible: arena-managed Linear weights (and fused Q|K|V groups) whose input width is a multiple of 8
(rows are zero padded to 16-byte chunks: 360 -> 368; the 180-channel audio stage keeps bf16).  See DESIGN.md section 15 for why this is a coverage path, not a faster one, on this model."""
import ctypes
import os

import torch

from . import runtime as rt
from .lib import lib, Fp8Item

_STATE = {"on": None}


def enable(flag=True):
    _STATE["on"] = bool(flag)


def enabled():
    if _STATE["on"] is None:
        _STATE["on"] = os.environ.get("AVEC_FP8", "0") == "1"
    return _STATE["on"] and rt.compute_dtype() == "bf16"


class _Entry:
    __slots__ = ("wq", "K", "Kp", "N", "w_amax", "a_amax")


class Fp8Weights:
    """e4m3 shadows of one ParamArena's Linear weights + the amax slots (first half: weights, second half: the activation each weight consumes)."""

    def __init__(self, arena):
        dev = arena.master.device
        items, self.entries = [], {}
        off_of = {id(p): o for p, o in zip(arena.params, arena.offsets)}
        done = set()
        total = 0
        for p in arena.params:
            sh = getattr(p, "_avec_shadow", None)
            if sh is None or id(p) in done or sh.Tm != 1 or sh.Cp != sh.C or sh.C % 8 != 0:      # (activation rows must be 16-byte aligned in bf16)
                continue
            grp = sh.group
            members = list(grp.weights) if grp is not None else [p]
            n = sum(w.numel() for w in members)
            o = off_of[id(members[0])]
            if any(off_of[id(w)] != o + i * members[0].numel() for i, w in enumerate(members)) or (o * 4) % 16 != 0:
                continue
            for w in members:
                done.add(id(w))
            e = _Entry()
            e.K, e.N = sh.C, n // sh.C
            e.Kp = (sh.C + 15) // 16 * 16                     # e4m3 rows are zero padded to whole 16-byte chunks (360 -> 368)
            items.append((o, total, n, members, e))
            total += e.N * e.Kp
        self.n = len(items)
        self.wq = torch.zeros(max(total, 16), dtype=torch.uint8, device=dev)
        self.amax = torch.zeros(2 * max(self.n, 1), dtype=torch.float32, device=dev)
        tab = (Fp8Item * max(self.n, 1))()
        for i, (o, qo, n, members, e) in enumerate(items):
            tab[i].src = arena.master.data_ptr() + 4 * o
            tab[i].dst = self.wq.data_ptr() + qo
            tab[i].n, tab[i].slot, tab[i].K, tab[i].Kp = n, i, e.K, e.Kp
            e.wq = self.wq.data_ptr() + qo
            e.w_amax = self.amax.data_ptr() + 4 * i
            e.a_amax = self.amax.data_ptr() + 4 * (self.n + i)
            for w in members:
                self.entries[id(w)] = e
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        self.blocks = 64

    def refresh(self):
        if self.n:
            self.amax[:self.n].zero_()
            lib.fp8_weights_refresh(self.table.data_ptr(), self.n, self.blocks, self.amax.data_ptr(), rt.stream())

    def begin_pass(self):
        if self.n:
            self.amax[self.n:].zero_()


def state_of(arena):
    st = getattr(arena, "_fp8", None)
    if st is None:
        st = arena._fp8 = Fp8Weights(arena)
        st.refresh()
    return st


def entry(weight, K):
    """the e4m3 shadow of `weight` (or of the fused group it leads) when fp8 is on and the product is eligible, else None"""
    if not enabled():
        return None
    sh = getattr(weight, "_avec_shadow", None)
    if sh is None or sh.arena is None:
        return None
    sh.arena.ensure_fresh()
    e = state_of(sh.arena).entries.get(id(weight))
    return e if (e is not None and e.K == K) else None
