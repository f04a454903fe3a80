"""Log-probabilities / entropy from logits — drop-in for the reference's ``vocab_parallel``
(vocab_parallel.py:399-467): ``gather_logprobs`` / ``gather_logprobs_entropy`` with the same
arguments, fp32 results, and an optional vocab-sharded (TP) path.

CUDA logits (bf16 / f16 / f32) go through the HIP kernels ``dta_logprob_entropy_fwd/bwd`` (or
``dta_logprob_entropy_shard_stats`` with a ``tp_group``) behind autograd Functions (ops.logprob_entropy): one read of
the logits per direction, no fp32 copy of them, so ``chunk_size`` has nothing left to bound there and is ignored; if the
HIP library is missing these calls raise.  CPU logits (the reference also runs on CPU: SURVEY §6; the gloo tests) take
the torch restatement below, which chunks rows as the reference does.

Differences in mechanism, not in results:
* the vocab-parallel branch packs its SUM reductions into ONE all-reduce after the MAX all-reduce (2 latency-bound
  RCCL calls instead of 3-4 per chunk: vocab_parallel.py:134,142,156 / 264,273,291,298) and saves no [rows, V/tp]
  tensor for backward (recomputed from the logits);
* the reference's shape quirk is kept: ``labels`` may hold fewer rows than ``logits`` (the engine
  passes B logits rows and B-1 labels, tte:190-193); logprobs then cover the first rows, entropy all.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import distributed as dist


def _rows(logits: torch.Tensor, labels: torch.Tensor):
    V = logits.shape[-1]
    lead = logits.shape[:-1]
    if labels.shape != lead:
        if not (labels.dim() == len(lead) and labels.shape[:-1] == lead[:-1] and labels.shape[-1] <= lead[-1]
                and all(s == 1 for s in lead[:-1])):
            raise ValueError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
    return logits.reshape(-1, V), labels.reshape(-1)


def _lse_entropy_chunk(x: torch.Tensor, want_entropy: bool):
    xf = x.float()
    lse = torch.logsumexp(xf, dim=-1)
    if not want_entropy:
        return xf, lse, None
    p = torch.exp(xf - lse[:, None])
    ent = lse - (p * xf).sum(-1)
    return xf, lse, ent


def _local(logits2d, labels1d, temperature, chunk, want_entropy):
    R = logits2d.shape[0]
    n_lab = labels1d.shape[0]
    lps, ents = [], []
    for a in range(0, R, chunk):
        b = min(a + chunk, R)
        x = logits2d[a:b]
        if temperature != 1.0:
            x = x.float() / temperature
        xf, lse, ent = _lse_entropy_chunk(x, want_entropy)
        nb = max(0, min(b, n_lab) - a)
        if nb > 0:
            lps.append(xf[:nb].gather(-1, labels1d[a:a + nb, None]).squeeze(-1) - lse[:nb])
        if want_entropy:
            ents.append(ent)
    lp = torch.cat(lps) if lps else logits2d.new_zeros(0, dtype=torch.float32)
    return lp, (torch.cat(ents) if want_entropy else None)


class _ShardedLogProbEntropy(torch.autograd.Function):
    """Vocab-sharded log-prob (+entropy).  Shard = contiguous slice rank·V/tp (vocab_parallel.py:128-130)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, labels: torch.Tensor, group, want_entropy: bool):
        rank = dist.get_rank(group)
        Vp = x.shape[-1]
        lo = rank * Vp
        x2 = x.reshape(-1, Vp)
        lab = labels.reshape(-1)
        mx = x2.max(dim=-1).values
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        e = torch.exp(x2 - mx[:, None])
        mine = (lab >= lo) & (lab < lo + Vp)
        idx = torch.where(mine, lab - lo, torch.zeros_like(lab))
        picked = torch.where(mine, x2.gather(-1, idx[:, None]).squeeze(-1) - mx, torch.zeros_like(mx))
        packed = torch.stack([e.sum(-1), picked, (e * x2).sum(-1) if want_entropy else torch.zeros_like(mx)], dim=-1)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)     # ONE packed SUM
        z, picked, ex = packed.unbind(-1)
        logz = mx + torch.log(z)
        lp = picked - torch.log(z)
        mean_x = ex / z
        ctx.save_for_backward(x2, logz, mean_x, mine, idx)
        ctx.shape, ctx.want_entropy = labels.shape, want_entropy
        ent = (logz - mean_x).reshape(labels.shape) if want_entropy else None
        return lp.reshape(labels.shape), ent

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        x2, logz, mean_x, mine, idx = ctx.saved_tensors
        p = torch.exp(x2 - logz[:, None])
        g = -p * g_lp.reshape(-1, 1)
        if ctx.want_entropy and g_ent is not None:
            g = g + g_ent.reshape(-1, 1) * p * (mean_x[:, None] - x2)
        g.scatter_add_(-1, idx[:, None], (g_lp.reshape(-1) * mine)[:, None].to(g.dtype))
        return g.reshape(ctx.shape + (x2.shape[-1],)), None, None, None


def _sharded(logits, labels, temperature, group, chunk, want_entropy):
    x = logits.float() / temperature if temperature != 1.0 else logits.float()
    R = x.shape[0]
    lps, ents = [], []
    for a in range(0, R, chunk):                  # chunks along dim 0, as the reference (vocab_parallel.py:39-59)
        lp, ent = _ShardedLogProbEntropy.apply(x[a:a + chunk], labels[a:a + chunk], group, want_entropy)
        lps.append(lp); ents.append(ent)
    return torch.cat(lps), (torch.cat(ents) if want_entropy else None)


def _hip(logits, labels, temperature, tp_group, want_entropy):
    """CUDA path: flatten, pad the labels of the shape quirk with -1 (= no label), run the HIP operator."""
    from . import ops
    if tp_group is None:
        x2, l1 = _rows(logits, labels)
    else:
        x2, l1 = logits.reshape(-1, logits.shape[-1]), labels.reshape(-1)
    R, n_lab = x2.shape[0], l1.shape[0]
    lab = l1.to(torch.long)
    if n_lab < R:
        lab = torch.cat([lab, lab.new_full((R - n_lab,), -1)])
    lp, ent = ops.logprob_entropy(x2, lab.contiguous(), temperature, want_entropy, tp_group)
    lp = lp[:n_lab].reshape(labels.shape)
    return lp, (ent.reshape(logits.shape[:-1]) if want_entropy else None)


def gather_logprobs(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                    tp_group: Optional["dist.ProcessGroup"] = None, chunk_size: int = 1024) -> torch.Tensor:
    sharded = tp_group is not None and dist.get_world_size(tp_group) > 1
    if logits.is_cuda:
        return _hip(logits, labels, temperature, tp_group if sharded else None, False)[0]
    if sharded:
        return _sharded(logits, labels, temperature, tp_group, chunk_size, False)[0]
    assert logits.shape[0] > 0, "Input logits must have at least one element"
    x2, l1 = _rows(logits, labels)
    lp, _ = _local(x2, l1, temperature, chunk_size, False)
    return lp.reshape(labels.shape)


def gather_logprobs_entropy(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                            tp_group: Optional["dist.ProcessGroup"] = None, chunk_size: int = 1024) -> Tuple[torch.Tensor, torch.Tensor]:
    sharded = tp_group is not None and dist.get_world_size(tp_group) > 1
    if logits.is_cuda:
        return _hip(logits, labels, temperature, tp_group if sharded else None, True)
    if sharded:
        return _sharded(logits, labels, temperature, tp_group, chunk_size, True)
    assert logits.shape[0] > 0, "Input logits must have at least one element"
    x2, l1 = _rows(logits, labels)
    lp, ent = _local(x2, l1, temperature, chunk_size, True)
    return lp.reshape(labels.shape), ent.reshape(logits.shape[:-1])
