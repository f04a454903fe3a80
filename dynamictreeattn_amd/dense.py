"""Dense baseline: one pass per ORIGINAL sequence, nothing shared (dense.py:7-53) — the
correctness counterpart of the tree engine.  Same model code and the same HIP attention operator in
its pure-causal stack form (q_offset 0, no subtree bound)."""
from __future__ import annotations

from typing import List

import torch

from . import ops
from .model import head_weight, packed_hidden_states
from .tree_training_engine import packed_logprob_entropy


def _one(model, ids: torch.Tensor, want_entropy: bool, checkpoint_layers: bool):
    n = ids.numel()
    depth = torch.arange(n, device=ids.device, dtype=torch.int32)
    h = packed_hidden_states(model, ids, depth, ops.stack_meta(0), checkpoint_layers)
    parent = depth.to(torch.long) - 1
    return packed_logprob_entropy(h, head_weight(model), ids, parent, want_entropy)


@torch.no_grad()
def forward(model, token_seqs: List[torch.LongTensor], use_tqdm: bool = False) -> List[torch.Tensor]:
    out = []
    with ops.weight_cache():
        for ids in token_seqs:
            lp, _ = _one(model, ids.to(model.device), False, False)
            out.append(lp[1:])
    return out


def backward(model, token_seqs: List[torch.LongTensor], attachs, loss_fn, act_ckpt: bool = False, use_tqdm: bool = False) -> float:
    total = 0.0
    with ops.weight_cache():
        for ids, att in zip(token_seqs, attachs):
            lp, ent = _one(model, ids.to(model.device), True, act_ckpt)
            loss = loss_fn(lp[1:], ent, att)
            loss.backward()
            total += loss.item()
    return total
