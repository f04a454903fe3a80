"""Data parallelism over tries: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm).  The reference only plans DP offline (data_parallel.py writes `{name}_bin{k}.pt`, the
bins then run one after another on one GPU, exp/exp_dp.py:43-49, and "DP time" is simulated as the max
over bins, exp/calc_time.py:24-42).  Here every rank computes the same partition, runs its bin, and the
parameter gradients are summed across ranks — the total loss is a plain sum over sequences
(tree_training_engine.py:396-398), so the sum of per-bin gradients is the full-batch gradient.

Every rank issues the SAME sequence of collectives of the SAME sizes whatever its bin holds: the bucket layout
is fixed over all `requires_grad` parameters, a parameter without a gradient on this rank contributes zeros,
and a rank whose bin is empty (K > number of leaves) skips the compute but still takes part in every reduce.
"""
from __future__ import annotations

import types
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import data_parallel as _lb
from .tree_time_model import TreeTimeModel


def my_bin(token_seqs: Sequence[torch.Tensor], rank: int, world: int, mode: str = "backward", block_size: Optional[int] = 2048,
           time_model: Optional[TreeTimeModel] = None, method: str = "LB_by_DFS_and_TM") -> List[int]:
    """Original sequence ids this rank processes (deterministic: every rank computes the same bins)."""
    if world == 1:
        return list(range(len(token_seqs)))
    return all_bins(token_seqs, world, mode, block_size, time_model, method)[rank]


def all_bins(token_seqs: Sequence[torch.Tensor], world: int, mode: str = "backward", block_size: Optional[int] = 2048,
             time_model: Optional[TreeTimeModel] = None, method: str = "LB_by_DFS_and_TM") -> List[List[int]]:
    """The K = `world` bins of data_parallel.py's balancers (public `LB_by_*` entries), in rank order."""
    tm = time_model or TreeTimeModel()
    args = types.SimpleNamespace(K=world, mode=mode, block_size=block_size if mode == "backward" else None)
    if method == "LB_by_n_tokens":
        return _lb.LB_by_n_tokens(token_seqs, world)
    return getattr(_lb, method)(token_seqs, tm, args)


def _buckets(params, bucket_bytes: int):
    """Parameters in reverse registration order (the order their gradients become ready: last layer first, the
    embedding last) cut into buckets of one dtype and at most `bucket_bytes` — a function of the parameter list only."""
    out, cur, n = [], [], 0
    for p in reversed([p for p in params if p.requires_grad]):
        b = p.numel() * p.element_size()
        if cur and (n + b > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
            out.append(cur); cur, n = [], 0
        cur.append(p); n += b
    if cur:
        out.append(cur)
    return out


def allreduce_grads(params, group=None, bucket_bytes: int = 1 << 29) -> None:
    """ONE logical all-reduce(SUM) of every parameter gradient — no averaging — for callers that hold no `GradReducer`:
    flat buckets over ALL `requires_grad` parameters (zeros where this rank has no gradient, which then becomes a real
    gradient tensor), so ranks with different bins still issue identical collectives."""
    for bucket in _buckets(list(params), bucket_bytes):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        o = 0
        for p in bucket:
            g = flat[o:o + p.numel()].view_as(p); o += p.numel()
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)


_ACTIVE: set = set()          # reducers between start() and finish()


def defer_active_reducers() -> None:
    """Called by the block-wise engine before its first backward: see GradReducer.defer."""
    for r in list(_ACTIVE):
        r.defer()


class GradReducer:
    """The ONE logical all-reduce(SUM) of the step, IN PLACE and overlapped with the backward pass (SURVEY §8e:
    "bucketed and overlapped").

    * One persistent flat buffer per bucket; every `param.grad` is a VIEW into it (``zero_grad()`` zero-fills the buffers and
      re-attaches the views — use it instead of ``model.zero_grad(set_to_none=True)``).  autograd accumulates into the views,
      ``dist.all_reduce`` runs on the flat buffer itself: no `torch.cat`, no copy back (round 1 moved 1.2 GB each way).
    * Buckets follow the order gradients become ready; a post-accumulate-grad hook counts a bucket down and, once it is complete,
      enqueues its asynchronous all-reduce — strictly in bucket order, so every rank issues the collectives in the same sequence
      whatever the readiness order inside a layer.  ``finish()`` launches whatever did not complete (parameters without a gradient
      this step; every bucket when the rank's bin was empty and no backward ran) and waits.

        red = GradReducer(model.parameters())
        red.zero_grad(); red.start(); engine.backward(...); red.finish()
    """

    def __init__(self, params, group=None, bucket_bytes: int = 1 << 28, overlap: bool = True):
        self.group, self.overlap = group, overlap
        self.buckets = _buckets(list(params), bucket_bytes)
        self.flats = [torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device) for b in self.buckets]
        self._views = []
        for b, flat in zip(self.buckets, self.flats):
            o = 0
            for p in b:
                self._views.append((p, flat[o:o + p.numel()].view_as(p))); o += p.numel()
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for p in b] if overlap else []
        self._active = False
        self._deferred = False
        self._works: list = []
        self.zero_grad()

    def zero_grad(self) -> None:
        for flat in self.flats:
            flat.zero_()
        for p, v in self._views:
            p.grad = v

    def defer(self) -> None:
        """Stop launching from the hooks for the rest of this step (everything goes out in `finish()`): the block-wise engine
        accumulates into every parameter once per BLOCK, so "this parameter's gradient arrived" no longer means "final"."""
        self._deferred = True

    def start(self) -> None:
        self._deferred = False
        _ACTIVE.add(self)
        for p, v in self._views:           # a `zero_grad(set_to_none=True)` or an optimizer may have replaced the views
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                if p.grad is not None:
                    v.copy_(p.grad)
                else:
                    v.zero_()              # "no gradient" is zero: the buffer still holds the previous step's reduced sums
                p.grad = v
        self._left = [len(b) for b in self.buckets]
        self._next = 0                    # next bucket index to launch
        self._works = []
        self._active = True

    def _on_grad(self, p) -> None:
        if not self._active or self._deferred:
            return
        self._left[self._bucket_of[id(p)]] -= 1
        while self._next < len(self.buckets) and self._left[self._next] <= 0:
            self._launch(self._next); self._next += 1

    def _launch(self, i: int) -> None:
        self._works.append(dist.all_reduce(self.flats[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        if not self._active:
            self.start()
        self._active = False
        _ACTIVE.discard(self)
        while self._next < len(self.buckets):
            self._launch(self._next); self._next += 1
        for w in self._works:
            w.wait()
        self._works = []

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


OverlappedGradAllReduce = GradReducer      # round-1 name
