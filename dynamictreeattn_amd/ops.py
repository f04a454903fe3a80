"""torch-facing operators over the C ABI (include/dta.h).  PyTorch supplies device memory, streams
and autograd plumbing only; all arithmetic of these ops runs in the HIP kernels."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import packing
from ._lib import check, lib, ptr

_DT = {torch.bfloat16: 0, torch.float16: 1}


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional live timing of the attention launches with HIP events recorded on the launch stream
    (bench.py's roofline leg).  Disabled unless `KernelTimer.active` is set to an instance."""
    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.spans = {"fwd": [], "bwd_dq": [], "bwd_dkv": []}

    def span(self, name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans[name].append((a, b))
        return a, b

    def totals_ms(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.spans.items()}


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dynamictreeattn_amd ops run on the MI355X only (tensor is on %s); there is no CPU path" % t.device)


@dataclass
class TreeAttnMeta:
    """Device-resident visibility metadata of one packed trie (or None fields for the stack form)."""
    T: int
    subtree_end: Optional[torch.Tensor]      # int32 [T]
    run_ptr: Optional[torch.Tensor]          # int32 [nqt+1]
    runs: Optional[torch.Tensor]             # int32 [nruns,4]
    ktile_qend: Optional[torch.Tensor]       # int32 [nkt]
    q_offset: int = 0
    pairs: int = 0                           # number of visible (query,key) pairs (for FLOP accounting)


def ktile_qend_from(subtree_end: torch.Tensor, tile: int = packing.KTILE) -> torch.Tensor:
    T = subtree_end.numel()
    nkt = (T + tile - 1) // tile
    pad = nkt * tile - T
    se = torch.nn.functional.pad(subtree_end, (0, pad), value=0) if pad else subtree_end
    return se.view(nkt, tile).amax(dim=1).to(torch.int32).contiguous()


def _strides(t):
    if t.stride(-1) != 1:
        raise ValueError("head_dim must be contiguous")
    return t.stride(0), t.stride(1)


def attn_fwd_raw(q, k, v, meta: TreeAttnMeta, scale: float):
    _require_cuda(q, k, v)
    Tq, Hq, D = q.shape
    Tk, Hkv, _ = k.shape
    if k.stride() != v.stride():
        v = v.contiguous(); k = k.contiguous()
    out = torch.empty((Tq, Hq, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((Tq, Hq), dtype=torch.float32, device=q.device)
    (qs, qh), (ks, kh), (os_, oh) = _strides(q), _strides(k), _strides(out)
    tm = KernelTimer.active
    if tm is not None:
        ev = tm.span("fwd"); ev[0].record()
    st = lib().dta_tree_attn_fwd_ex(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(meta.subtree_end), ptr(meta.run_ptr), ptr(meta.runs),
                                    Tq, Tk, meta.q_offset, Hq, Hkv, D, qs, qh, ks, kh, os_, oh, float(scale), _DT[q.dtype], _stream())
    if tm is not None:
        ev[1].record()
    check(st, "dta_tree_attn_fwd")
    return out, lse, k, v


def attn_bwd_raw(q, k, v, out, dout, lse, meta: TreeAttnMeta, scale: float, dk=None, dv=None, accumulate=False):
    Tq, Hq, D = q.shape
    Tk, Hkv, _ = k.shape
    if dout.stride() != out.stride():
        dout = dout.contiguous(); out = out.contiguous()
    dq = torch.empty((Tq, Hq, D), dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty((Tk, Hkv, D), dtype=q.dtype, device=q.device); dv = torch.empty_like(dk)
    delta = torch.empty((Tq, Hq), dtype=torch.float32, device=q.device)
    (qs, qh), (ks, kh), (os_, oh), (dqs, dqh), (dks, dkh) = _strides(q), _strides(k), _strides(out), _strides(dq), _strides(dk)
    def launch(which):
        return lib().dta_tree_attn_bwd_ex(ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                                          ptr(meta.subtree_end), ptr(meta.run_ptr), ptr(meta.runs), ptr(meta.ktile_qend),
                                          Tq, Tk, meta.q_offset, Hq, Hkv, D, qs, qh, ks, kh, os_, oh, dqs, dqh, dks, dkh,
                                          float(scale), _DT[q.dtype], 1 if accumulate else 0, which, _stream())
    tm = KernelTimer.active
    if tm is None:
        check(launch(3), "dta_tree_attn_bwd")
    else:
        for which, name in ((1, "bwd_dq"), (2, "bwd_dkv")):
            a, b = tm.span(name); a.record(); st = launch(which); b.record()
            check(st, "dta_tree_attn_bwd")
    return dq, dk, dv


class _TreeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, meta: TreeAttnMeta, scale: float):
        out, lse, k, v = attn_fwd_raw(q, k, v, meta, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta, ctx.scale = meta, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = attn_bwd_raw(q, k, v, out, dout, lse, ctx.meta, ctx.scale)
        return dq, dk, dv, None, None


def tree_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, meta: TreeAttnMeta, scale: Optional[float] = None) -> torch.Tensor:
    """q [T,Hq,128], k/v [T,Hkv,128] packed in DFS pre-order -> out [T,Hq,128].  Differentiable."""
    if q.dtype not in _DT:
        raise TypeError("tree_attention supports bf16 / f16 (got %s)" % q.dtype)
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    return _TreeAttention.apply(q, k, v, meta, scale)


def stack_meta(start: int) -> TreeAttnMeta:
    """Rectangular-causal stack form: query i sits at stack position start+i (tte:171-186)."""
    return TreeAttnMeta(T=0, subtree_end=None, run_ptr=None, runs=None, ktile_qend=None, q_offset=start)


def meta_from_plan(plan: packing.SegmentPlan, subtree_end: torch.Tensor, device) -> TreeAttnMeta:
    rp, runs = packing.plan_qtile_runs(plan)
    return TreeAttnMeta(T=plan.T, subtree_end=subtree_end,
                        run_ptr=torch.from_numpy(rp).to(device), runs=torch.from_numpy(runs).to(device).contiguous(),
                        ktile_qend=ktile_qend_from(subtree_end))
