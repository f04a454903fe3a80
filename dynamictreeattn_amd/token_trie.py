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

import contextlib
import os
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


_TRIE_STREAMS: dict = {}


def _trie_stream(device):
    st = _TRIE_STREAMS.get(device.index)
    if st is None:
        st = _TRIE_STREAMS[device.index] = torch.cuda.Stream(device=device, priority=-1)
    return st


class _DeviceTokens:
    """All sequences of a batch concatenated in HBM (int64) + where each one starts."""

    def __init__(self, inputs: Sequence[torch.Tensor], device):
        lens = [int(t.numel()) for t in inputs]
        starts = np.zeros(len(lens) + 1, np.int64)
        np.cumsum(lens, out=starts[1:])
        from ._staging import upload
        flat = torch.cat([t.reshape(-1).to(torch.long) for t in inputs]) if inputs else torch.zeros(0, dtype=torch.long)
        if flat.is_cuda:                      # token ids that already live on a GPU (the reference takes tensors on any device)
            self.tokens = flat.to(device)
        else:
            self.tokens = upload([flat.numpy()], device, np.int64)[0]      # one asynchronous copy out of page-locked staging
        self.starts = starts[:-1].copy()      # per ORIGINAL sequence id
        self.lens = np.asarray(lens, np.int32)
        self.device = device


def _device_trie_arrays(dev: _DeviceTokens, order: Sequence[int], leafize: bool):
    """LCP of adjacent sequences taken in `order` (and, if asked, the leafization) on the GPU.
    Returns (lcp list, unsorted pair count, leaf positions or None).  ONE device->host read per call: both kernels write
    into one int32 buffer [lcp (S-1) | unsorted | leaf_pos (S) | leaf_lcp (S) | seq_leaf (S) | M]."""
    from ._lib import check, lib, ptr
    S = len(order)
    d = dev.device
    idx = np.asarray(order, np.int64)
    from ._staging import upload
    starts, lens64 = upload([dev.starts[idx], dev.lens[idx]], d, np.int64)                                  # one asynchronous H2D: starts | lens
    lens = lens64.to(torch.int32)
    n_lcp = max(S - 1, 1)
    buf = torch.zeros(n_lcp + 1 + (3 * S + 1 if leafize else 0), dtype=torch.int32, device=d)
    with torch.cuda.device(d):
        stream = torch.cuda.current_stream(d).cuda_stream
        check(lib().dta_lcp_adjacent(ptr(dev.tokens), ptr(starts), ptr(lens), S, ptr(buf), buf[n_lcp:].data_ptr(), stream), "dta_lcp_adjacent")
        if leafize:
            o = n_lcp + 1
            check(lib().dta_leafize(ptr(lens), ptr(buf), S, buf[o:].data_ptr(), buf[o + S:].data_ptr(), buf[o + 2 * S:].data_ptr(),
                                    buf[o + 3 * S:].data_ptr(), stream), "dta_leafize")
    host = buf.cpu().numpy()
    if not leafize:
        return host[:S - 1].tolist(), int(host[n_lcp]), None
    r = host[n_lcp + 1:]
    M = int(r[3 * S])
    return r[S:S + M - 1].tolist(), int(host[n_lcp]), r[:M].tolist()


def _range_min_table(a: Sequence[int]):
    """Sparse table for range-minimum queries over `a` (host; M leaves)."""
    t = [np.asarray(a, np.int64)]
    k = 1
    while 2 * k <= len(a):
        prev = t[-1]
        t.append(np.minimum(prev[:-k], prev[k:]))
        k *= 2
    return t


def _range_min(t, lo: int, hi: int) -> int:
    """min(a[lo:hi]), hi > lo."""
    j = (hi - lo).bit_length() - 1
    return int(min(t[j][lo], t[j][hi - (1 << j)]))


class TokenTrie:
    def __init__(self, inputs: List[torch.LongTensor], attachs: Optional[List[dict]] = None, sorted: bool = False, device=None):
        if attachs is not None:
            assert len(inputs) == len(attachs), "Length of inputs and attachs must match."
        else:
            attachs = [{} for _ in range(len(inputs))]
        for sid, att in enumerate(attachs):                 # mutates the caller's dicts, as the reference does
            att["_sequence_batch_id"] = sid
        S = len(inputs)
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("TokenTrie needs the MI355X: LCP / leafization run in HIP kernels and there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if S == 0:                                          # an empty batch is an empty trie (token_trie.py:52-82 with no inputs)
            self._dev = _DeviceTokens([], device)
            self.inputs, self.attach_lists, self._leaf_src, self.lcp_lens, self.lens = [], [], [], [], []
            self._sorted_pos, self._sorted_lcp = [], []
            self.n_sequences, self.n_tokens = 0, 0
            return
        if sorted:
            order = list(range(S))
        else:
            keys = _sort_keys(inputs)
            order = list(range(S))
            order.sort(key=keys.__getitem__)
        # Host-resident token lists are uploaded and indexed on a SIDE stream of high priority: the device->host read of the LCP/leaf table
        # below then waits for these two small kernels only, not for whatever the caller has queued on its own stream (an engine call
        # returns with its backward still running - tree_training_engine.py `_backward` - and the next trie is built under it).
        side = None
        if device.type == "cuda" and os.environ.get("DTA_TRIE_STREAM", "1") != "0" and not any(isinstance(t, torch.Tensor) and t.is_cuda for t in inputs):
            side = _trie_stream(device)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            self._dev = _DeviceTokens(inputs, device)
            leaf_lcp, unsorted, leaf_pos = _device_trie_arrays(self._dev, order, True)
        if side is not None:                                # the caller's stream consumes the token buffer (packing, embedding lookup)
            main = torch.cuda.current_stream(device)
            main.wait_stream(side)
            self._dev.tokens.record_stream(main)
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
        # the leaves are in lexicographic order here: remember it, so that the LCP of ANY two leaves later is a range minimum
        self._sorted_pos = list(range(len(self.lens)))      # current leaf -> position in the sorted leaf order
        self._sorted_lcp = list(self.lcp_lens)              # LCP of sorted neighbours
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
        self._sorted_pos = [self._sorted_pos[i] for i in order]
        # The reference recomputes the adjacent LCPs from the tokens (token_trie.py:94).  For leaves of ONE sorted batch the
        # LCP of two leaves is the minimum of the sorted-neighbour LCPs between them, so the new values follow on the host
        # from the construction-time kernel output: no launch and no device->host read per permutation.
        # `lcp_from_tokens()` is the kernel form of the same values (tests compare the two).
        if len(order) > 1:
            t = _range_min_table(self._sorted_lcp)
            sp = self._sorted_pos
            self.lcp_lens = [_range_min(t, min(sp[i], sp[i + 1]), max(sp[i], sp[i + 1])) if sp[i] != sp[i + 1] else int(self.lens[i])
                             for i in range(len(order) - 1)]
        else:
            self.lcp_lens = []

    def lcp_from_tokens(self) -> List[int]:
        """Adjacent LCPs of the leaves in their CURRENT order, recomputed from the device-resident tokens by
        `dta_lcp_adjacent` (what token_trie.py:94 does with `_lcp_torch`)."""
        if len(self._leaf_src) < 2:
            return []
        return _device_trie_arrays(self._dev, self._leaf_src, False)[0]

    def forward_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_forward()[0])

    def backward_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_backward()[0])

    def random_permute(self):
        self.permute(CompressedTrie(self.lens, self.lcp_lens).get_order_random())
