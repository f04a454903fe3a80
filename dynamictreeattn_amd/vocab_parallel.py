"""Log-probabilities / entropy from logits — drop-in for the reference's ``vocab_parallel``
(vocab_parallel.py:399-467): ``gather_logprobs`` / ``gather_logprobs_entropy`` with the same
arguments, fp32 results, and an optional vocab-sharded (TP) path.

Logits (bf16 / f16 / f32 on the MI355X; other float dtypes are up-cast with ``.float()`` as the reference does) go through the
HIP kernels ``dta_logprob_entropy_fwd/bwd`` (or ``dta_logprob_entropy_shard_stats`` with a ``tp_group``) behind autograd
Functions (ops.logprob_entropy): one read of the logits per direction, no fp32 copy of them, so ``chunk_size`` has nothing left
to bound and is ignored.  There is no CPU path in the product: host logits raise like every other operator of this package (the
torch restatement the gloo tests run on CPU lives in tests/hostmirror.py and is patched over ``_device_logprob_entropy`` there).

Differences in mechanism, not in results:
* the vocab-parallel branch packs its SUM reductions into ONE all-reduce after the MAX all-reduce (2 latency-bound
  RCCL calls for ALL rows instead of 3-4 per 1024-row chunk: vocab_parallel.py:134,142,156 / 264,273,291,298) and saves no
  [rows, V/tp] tensor for backward (recomputed from the logits);
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


def _device_logprob_entropy(logits2d, labels1d, temperature, want_entropy, tp_group):
    """(logprob [R], entropy [R] | None): the HIP operator.  Raises for host tensors (ops._on) and when the library is missing."""
    from . import ops
    if logits2d.dtype not in ops._DT_LOGITS:
        if not logits2d.is_floating_point():
            raise TypeError(f"logits must be floating point (got {logits2d.dtype})")
        logits2d = logits2d.float()               # e.g. float64: the reference computes in fp32 whatever comes in (vocab_parallel.py:14,22)
    return ops.logprob_entropy(logits2d, labels1d, temperature, want_entropy, tp_group)


def _run(logits, labels, temperature, tp_group, want_entropy):
    """Flatten, pad the labels of the shape quirk with -1 (= no label), run the operator."""
    if tp_group is None:
        x2, l1 = _rows(logits, labels)
    else:
        x2, l1 = logits.reshape(-1, logits.shape[-1]), labels.reshape(-1)
    R, n_lab = x2.shape[0], l1.shape[0]
    lab = l1.to(torch.long)
    if n_lab < R:
        lab = torch.cat([lab, lab.new_full((R - n_lab,), -1)])
    lp, ent = _device_logprob_entropy(x2, lab.contiguous(), temperature, want_entropy, tp_group)
    lp = lp[:n_lab].reshape(labels.shape)
    return lp, (ent.reshape(logits.shape[:-1]) if want_entropy else None)


def gather_logprobs(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                    tp_group: Optional["dist.ProcessGroup"] = None, chunk_size: int = 1024) -> torch.Tensor:
    sharded = tp_group is not None and dist.get_world_size(tp_group) > 1
    if not sharded:
        assert logits.shape[0] > 0, "Input logits must have at least one element"        # vocab_parallel.py:46
    return _run(logits, labels, temperature, tp_group if sharded else None, False)[0]


def gather_logprobs_entropy(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0,
                            tp_group: Optional["dist.ProcessGroup"] = None, chunk_size: int = 1024) -> Tuple[torch.Tensor, torch.Tensor]:
    sharded = tp_group is not None and dist.get_world_size(tp_group) > 1
    if not sharded:
        assert logits.shape[0] > 0, "Input logits must have at least one element"
    return _run(logits, labels, temperature, tp_group if sharded else None, True)
