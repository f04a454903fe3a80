"""TokenTrie — drop-in for the reference's ``token_trie.TokenTrie`` (token_trie.py:51-109).

Host work: tagging the attach dicts and a stable lexicographic sort by big-endian byte keys
(memcmp order == token order; a proper prefix sorts first; duplicates keep input order — the
semantics of ``pairs.sort(key=lambda x: x[0].tolist())``, token_trie.py:68-71).
Device work (HIP, `dta_lcp_adjacent` + `dta_leafize`): adjacent LCPs with the sortedness check
and leafization as a stream compaction, on the token buffer that is uploaded once and then reused
by the engine for the packed gather (no per-leaf H2D as in tree_training_engine.py:583).

There is no CPU implementation in the product: without the HIP library / a GPU the constructor
raises.  (tests monkeypatch `_device_trie_arrays` with the oracle to exercise the host logic.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .trie import CompressedTrie, _get_stats

__all__ = ["TokenTrie"]


def _sort_keys(inputs: Sequence[torch.Tensor]) -> List[bytes]:
    keys = []
    for t in inputs:
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        if a.size and int(a.min()) < 0:
            raise ValueError("token ids must be non-negative")
        keys.append(a.astype(">u8", copy=False).tobytes())
    return keys


class _DeviceTokens:
    """All sequences of a batch concatenated in HBM (int64) + where each one starts."""

    def __init__(self, inputs: Sequence[torch.Tensor], device):
        lens = [int(t.numel()) for t in inputs]
        starts = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(lens, out=starts[1:])
        flat = torch.cat([t.reshape(-1).to(torch.long) for t in inputs]) if inputs else torch.zeros(0, dtype=torch.long)
        self.tokens = flat.to(device, non_blocking=True)
        self.starts = starts[:-1].copy()      # per ORIGINAL sequence id
        self.lens = np.asarray(lens, np.int32)
        self.device = device


def _device_trie_arrays(dev: _DeviceTokens, order: Sequence[int], leafize: bool):
    """LCP of adjacent sequences taken in `order` (and, if asked, the leafization) on the GPU.
    Returns (lcp list, unsorted pair count, leaf positions or None)."""
    from ._lib import check, lib, ptr
    S = len(order)
    d = dev.device
    idx = np.asarray(order, np.int64)
    starts = torch.from_numpy(dev.starts[idx]).to(d)
    lens = torch.from_numpy(dev.lens[idx]).to(d)
    out = torch.zeros(max(S - 1, 1) + 1, dtype=torch.int32, device=d)        # [lcp..., unsorted]
    stream = torch.cuda.current_stream().cuda_stream
    check(lib().dta_lcp_adjacent(ptr(dev.tokens), ptr(starts), ptr(lens), S, ptr(out), out[-1:].data_ptr(), stream), "dta_lcp_adjacent")
    if not leafize:
        host = out.cpu().numpy()
        return host[:S - 1].tolist(), int(host[-1]), None
    res = torch.empty(3 * S + 1, dtype=torch.int32, device=d)                # leaf_pos | leaf_lcp | seq_leaf | M
    check(lib().dta_leafize(ptr(lens), ptr(out), S, res[:S].data_ptr(), res[S:2 * S].data_ptr(), res[2 * S:3 * S].data_ptr(),
                            res[3 * S:].data_ptr(), stream), "dta_leafize")
    host, r = out.cpu().numpy(), res.cpu().numpy()
    M = int(r[3 * S])
    return r[S:S + M - 1].tolist(), int(host[-1]), r[:M].tolist()


class TokenTrie:
    def __init__(self, inputs: List[torch.LongTensor], attachs: Optional[List[dict]] = None, sorted: bool = False, device=None):
        if attachs is not None:
            assert len(inputs) == len(attachs), "Length of inputs and attachs must match."
        else:
            attachs = [{} for _ in range(len(inputs))]
        for sid, att in enumerate(attachs):                 # mutates the caller's dicts, as the reference does
            att["_sequence_batch_id"] = sid
        S = len(inputs)
        if sorted:
            order = list(range(S))
        else:
            keys = _sort_keys(inputs)
            order = list(range(S))
            order.sort(key=keys.__getitem__)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("TokenTrie needs the MI355X: LCP / leafization run in HIP kernels and there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self._dev = _DeviceTokens(inputs, device)
        leaf_lcp, unsorted, leaf_pos = _device_trie_arrays(self._dev, order, True)
        if unsorted:
            raise ValueError("Input_ids not sorted in lexicographic order.")
        self.inputs, self.attach_lists, self._leaf_src = [], [], []
        first = 0
        for pos in leaf_pos:
            self.inputs.append(inputs[order[pos]])
            self._leaf_src.append(order[pos])               # original id whose tokens are the leaf's tokens
            self.attach_lists.append([(attachs[order[k]], int(self._dev.lens[order[k]])) for k in range(first, pos + 1)])
            first = pos + 1
        self.lcp_lens = [int(c) for c in leaf_lcp]
        self.lens = [int(self._dev.lens[s]) for s in self._leaf_src]
        self.n_sequences = S
        self.n_tokens = int(self._dev.lens.sum())

    def get_stats(self, mode: str, block_size: Optional[int] = None):
        stats = _get_stats(self.lens, self.lcp_lens, mode, block_size)
        stats["n_sequences"] = self.n_sequences
        stats["n_tokens"] = self.n_tokens
        return stats

    def permute(self, order):
        self.inputs = [self.inputs[i] for i in order]
        self.attach_lists = [self.attach_lists[i] for i in order]
        self.lens = [self.lens[i] for i in order]
        self._leaf_src = [self._leaf_src[i] for i in order]
        # LCPs are recomputed from the tokens (token_trie.py:94) — on the device-resident copy
        self.lcp_lens, _, _ = _device_trie_arrays(self._dev, self._leaf_src, False) if len(order) > 1 else ([], 0, None)

    def forward_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_forward()[0])

    def backward_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_backward()[0])

    def random_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_random())
