"""Data parallelism over tries: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on ROCm).  The reference only plans DP offline (data_parallel.py writes `{name}_bin{k}.pt`, the
bins then run one after another on one GPU, exp/exp_dp.py:43-49, and "DP time" is simulated as the max
over bins, exp/calc_time.py:24-42).  Here every rank computes the same partition, runs its bin, and the
parameter gradients are summed across ranks — the total loss is a plain sum over sequences
(tree_training_engine.py:396-398), so the sum of per-bin gradients is the full-batch gradient."""
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
    tm = time_model or TreeTimeModel()
    args = types.SimpleNamespace(K=world, mode=mode, block_size=block_size if mode == "backward" else None)
    if method == "LB_by_n_tokens":
        bins = _lb.LB_by_n_tokens(token_seqs, world)
    else:
        bins = getattr(_lb, method)(token_seqs, tm, args)
    return bins[rank]


def allreduce_grads(params, group=None, bucket_bytes: int = 1 << 29) -> None:
    """ONE logical all-reduce(SUM) of every parameter gradient — no averaging — issued as a few flat
    buckets (default 512 MB) so that RCCL drives all xGMI links with large messages."""
    grads = [p.grad for p in params if p.grad is not None]
    i = 0
    while i < len(grads):
        chunk, n = [], 0
        while i < len(grads) and (not chunk or (n + grads[i].numel() * grads[i].element_size() <= bucket_bytes and grads[i].dtype == chunk[0].dtype)):
            chunk.append(grads[i]); n += grads[i].numel() * grads[i].element_size(); i += 1
        flat = torch.cat([g.reshape(-1) for g in chunk])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        o = 0
        for g in chunk:
            g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()


class OverlappedGradAllReduce:
    """The same ONE logical all-reduce(SUM) as `allreduce_grads`, overlapped with the backward pass (SURVEY §8e:
    "bucketed and overlapped").  Parameters are cut into flat buckets in the order their gradients become ready
    (reverse registration order: last layer first, the embedding last); a post-accumulate-grad hook counts a bucket
    down and, once it is complete, enqueues its asynchronous all-reduce — strictly in bucket order, so every rank
    issues the collectives in the same sequence whatever the readiness order inside a layer.  `finish()` launches
    whatever did not complete (parameters without a gradient this step), waits, and copies the sums back.

        red = OverlappedGradAllReduce(model.parameters())
        red.start(); engine.backward(...); red.finish()
    """

    def __init__(self, params, group=None, bucket_bytes: int = 1 << 28):
        self.group = group
        ps = [p for p in params if p.requires_grad]
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, n = [], 0
        for p in reversed(ps):
            b = p.numel() * p.element_size()
            if cur and (n + b > bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(cur); cur, n = [], 0
            cur.append(p); n += b
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in ps]
        self._active = False

    def start(self) -> None:
        self._left = [len(b) for b in self.buckets]
        self._next = 0                    # next bucket index to launch
        self._works = []
        self._active = True

    def _on_grad(self, p) -> None:
        if not self._active:
            return
        self._left[self._bucket_of[id(p)]] -= 1
        while self._next < len(self.buckets) and self._left[self._next] <= 0:
            self._launch(self._next); self._next += 1

    def _launch(self, i: int) -> None:
        grads = [p.grad for p in self.buckets[i] if p.grad is not None]
        if not grads:
            self._works.append(None); return
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, flat, grads))

    def finish(self) -> None:
        self._active = False
        while self._next < len(self.buckets):
            self._launch(self._next); self._next += 1
        for w in self._works:
            if w is None:
                continue
            work, flat, grads = w
            work.wait()
            o = 0
            for g in grads:
                g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()
        self._works = []

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
