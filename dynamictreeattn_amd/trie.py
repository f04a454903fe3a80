"""Radix trie over sorted leaves — host-side planning (integer, O(M log M), microseconds).

Drop-in surface of the reference's ``trie.py``: ``CompressedTrie(lens, lcp_lens)`` with ``.nodes``
(``depth, seq_id, chain_tail_depth, child_ids``; ids in creation order, root 0),
``get_order_forward/backward/random``, ``get_lens``; module functions ``_get_stats``,
``_get_subtrie``.  Reference: trie.py:7-56 (stats), 58-133 (build), 135-238 (orders), 240-242.

Iterative throughout (the reference recurses: Python's recursion limit caps its trie height).
"""
from __future__ import annotations

import random as _random
from typing import Iterable, List, Optional, Sequence

__all__ = ["CTNode", "CompressedTrie", "_get_stats", "_get_subtrie", "pop_block_starts"]


class CTNode:
    __slots__ = ("depth", "seq_id", "chain_tail_depth", "child_ids")

    def __init__(self, depth: int = 0, seq_id: int = -1):
        self.depth, self.seq_id, self.chain_tail_depth, self.child_ids = depth, seq_id, 0, []

    def __repr__(self):
        return f"CTNode(depth={self.depth}, seq_id={self.seq_id}, chain_tail_depth={self.chain_tail_depth}, child_ids={self.child_ids})"


def _ceil_div(a: int, b: int) -> int:
    return -(-a // b)


def pop_block_starts(start: int, end: int, block_size: Optional[int]) -> List[int]:
    """Start offsets of the equal blocks a pop of [start,end) is cut into, last block first
    (tree_training_engine.py:503-511; also trie.py:38-43 and tte:36-44)."""
    span = end - start
    if span <= 0:
        return []
    if block_size is None:
        return [start]
    nb = _ceil_div(span, block_size)
    bs = _ceil_div(span, nb)
    return [max(end - (b + 1) * bs, start) for b in range(nb)]


def _get_stats(lens: Sequence[int], lcp_lens: Sequence[int], mode: str, block_size: Optional[int] = None) -> dict:
    """Closed-form cost statistics (trie.py:7-56).  For the last leaf in backward mode the reference
    reads one past the end of ``lcp_lens`` (IndexError) whenever that leaf's pop fits one block; the
    value it stands for is the last leaf's pop start, 0, which is what is used here."""
    if mode not in ("forward", "backward"):
        raise ValueError(f"Unsupported mode: {mode}")
    M = len(lens)
    prev = [0] + list(lcp_lens)
    stats = {
        "n_leaf_sequences": M,
        "n_tree_tokens": sum(lens) - sum(lcp_lens),
        "sum_prefix_len": 0,
        "sum_depth": sum((prev[i] + lens[i] - 1) * (lens[i] - prev[i]) // 2 for i in range(M)),
    }
    if mode == "forward":
        stats["sum_prefix_len"] = sum(lcp_lens)
        return stats
    f1 = 0
    for i in range(M):
        pop_to = lcp_lens[i] if i < M - 1 else 0
        span = lens[i] - pop_to
        if block_size is None or span <= block_size:
            stats["sum_prefix_len"] += pop_to
            f1_end = pop_to
        else:
            starts = pop_block_starts(pop_to, lens[i], block_size)
            stats["sum_prefix_len"] += sum(starts)
            f1_end = starts[0]                      # = end - actual block size: only the last block is left for the pop
        f1 += max(f1_end - prev[i], 0)
    stats["n_f1_tokens"] = f1
    return stats


class CompressedTrie:
    def __init__(self, lens: Sequence[int], lcp_lens: Sequence[int]):
        if len(lcp_lens) != len(lens) - 1:
            raise ValueError("len(lcp_lens) must be len(lens) - 1")
        self.nodes: List[CTNode] = [CTNode(0, -1)]
        self.order = self.lens = self.lcp_lens = None
        self._grow(lens, lcp_lens)

    # -- construction: monotone stack of open nodes (trie.py:93-133) ---------------------------
    def _grow(self, lens, lcps):
        nodes = self.nodes
        open_ids = [0]
        for sid, n in enumerate(lens):
            cut = lcps[sid - 1] if sid else 0
            if len(open_ids) > 1:
                # close everything strictly deeper than the branching depth
                while nodes[open_ids[-2]].depth > cut:
                    closed = open_ids.pop()
                    nodes[open_ids[-1]].child_ids.append(closed)
                closed = open_ids.pop()
                if nodes[open_ids[-1]].depth < cut:
                    nodes.append(CTNode(cut, -1)); open_ids.append(len(nodes) - 1)
                nodes[open_ids[-1]].child_ids.append(closed)
            elif nodes[open_ids[-1]].depth < cut:
                nodes.append(CTNode(cut, -1)); open_ids.append(len(nodes) - 1)
            nodes.append(CTNode(n, sid)); open_ids.append(len(nodes) - 1)
        while len(open_ids) > 1:
            closed = open_ids.pop()
            nodes[open_ids[-1]].child_ids.append(closed)

    # -- child rankings (trie.py:190-214) -----------------------------------------------------------
    def _rank(self, nid: int, how: str, seed=None) -> List[int]:
        nodes = self.nodes
        kids = nodes[nid].child_ids
        if how == "forward":
            return sorted(kids, key=lambda c: nodes[c].chain_tail_depth)
        if how == "backward":
            return sorted(kids, key=lambda c: (bool(nodes[c].child_ids), nodes[c].chain_tail_depth))
        kids = list(kids)
        (_random.Random(seed) if seed is not None else _random).shuffle(kids)
        return kids

    def _chain_tails(self, how: str, seed=None):                       # trie.py:135-148
        nodes = self.nodes
        work = [(0, 0)]
        while work:
            nid, phase = work.pop()
            nd = nodes[nid]
            if nd.seq_id != -1:
                nd.chain_tail_depth = nd.depth
            elif phase == 0:
                work.append((nid, 1))
                # children in creation order (the stack pops from the end): a random ranking then draws from the RNG in the
                # reference's post-order, so `get_order_random(seed=None)` under a seeded global RNG gives the reference's order
                work.extend((c, 0) for c in reversed(nd.child_ids))
            else:
                nd.chain_tail_depth = nodes[self._rank(nid, how, seed)[0]].chain_tail_depth

    def _traverse(self, how: str, seed=None):                          # trie.py:171-188, 216-223
        self._chain_tails(how, seed)
        nodes = self.nodes
        self.order, self.lens, self.lcp_lens = [], [], []
        lca = 0
        work = [[0, self._rank(0, how, seed), 0]]
        while work:
            top = work[-1]
            nid, kids, pos = top
            if pos == len(kids):
                work.pop(); continue
            top[2] = pos + 1
            lca = min(lca, nodes[nid].depth)
            child = nodes[kids[pos]]
            if child.seq_id != -1:
                self.order.append(child.seq_id); self.lens.append(child.depth); self.lcp_lens.append(lca)
                lca = child.depth
            else:
                work.append([kids[pos], self._rank(kids[pos], how, seed), 0])

    def get_order_forward(self):
        self._traverse("forward")
        return self.order, self.lens, self.lcp_lens[1:]

    def get_order_backward(self):
        self._traverse("backward")
        return self.order[::-1], self.lens[::-1], self.lcp_lens[1:][::-1]

    def get_order_random(self, seed: Optional[int] = None):
        self._traverse("random", seed)
        return self.order

    def get_lens(self, seq_set: Iterable[int]):                         # trie.py:150-169
        keep = seq_set if isinstance(seq_set, (set, frozenset)) else set(seq_set)
        nodes = self.nodes
        lens, lcps = [], []
        lca = 0
        work = [[0, 0]]
        while work:
            top = work[-1]
            nid, pos = top
            kids = nodes[nid].child_ids
            if pos == len(kids):
                work.pop(); continue
            top[1] = pos + 1
            lca = min(lca, nodes[nid].depth)
            child = nodes[kids[pos]]
            if child.seq_id == -1:
                work.append([kids[pos], 0])
            elif child.seq_id in keep:
                lens.append(child.depth); lcps.append(lca)
                lca = child.depth
        self.lens, self.lcp_lens = lens, lcps
        return lens, lcps[1:]


def _get_subtrie(trie: CompressedTrie, seq_set) -> CompressedTrie:
    lens, lcps = trie.get_lens(seq_set)
    return CompressedTrie(lens, lcps)
