"""Host-side planning of the packed (DFS pre-order) trie layout.

Given the leaves of a TokenTrie in their current DFS order — ``lens[M]`` and ``lcp_lens[M-1]`` —
leaf i contributes the *segment* of its tokens at depths ``[lcp[i-1], len[i])``; segments are laid
back to back, so a token's packed index is ``seg_off[i] + depth - lcp[i-1]`` and its RoPE position
is its depth (the stack position of tree_training_engine.py:166, 293).

This module only produces O(M·height) integer tables on the host; the per-token arrays
(token id, depth, parent, subtree_end) are expanded on the GPU by ``dta_preorder_meta``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

QTILE = 128
KTILE = 128      # = DTA_KTILE (dta.h)


@dataclass
class SegmentPlan:
    M: int
    T: int
    seg_off: np.ndarray          # int32 [M+1]
    seg_depth0: np.ndarray       # int32 [M]   depth of the segment's first token (= lcp with the previous leaf)
    parent_of_seg: np.ndarray    # int32 [M]   packed index of the token above the segment's first token (-1 at the root)
    brk_ptr: np.ndarray          # int32 [M+1]
    brk_depth: np.ndarray        # int32 [nbrk] ascending per segment; first entry == seg_depth0
    brk_end: np.ndarray          # int32 [nbrk] subtree_end of the segment's tokens at depth >= brk_depth (until the next break)
    path_runs: List[List[Tuple[int, int]]]   # per segment: packed [begin,end) runs of its ancestors, root first


def plan_segments(lens: Sequence[int], lcp_lens: Sequence[int]) -> SegmentPlan:
    M = len(lens)
    if len(lcp_lens) != M - 1:
        raise ValueError("lcp_lens must have len(lens)-1 entries")
    d0 = [0] + [int(c) for c in lcp_lens]
    seg_off = [0] * (M + 1)
    for i in range(M):
        if not (0 <= d0[i] <= lens[i]) or (i > 0 and d0[i] > lens[i - 1]):
            raise ValueError("lcp_lens inconsistent with lens")
        seg_off[i + 1] = seg_off[i] + (lens[i] - d0[i])
    T = seg_off[M]
    live: List[List[int]] = []            # [segment, depth_lo, depth_hi) intervals on the current root->leaf path
    brk: List[List[Tuple[int, int]]] = [[] for _ in range(M)]
    parent = [-1] * M
    path_runs: List[List[Tuple[int, int]]] = []
    for i in range(M):
        cut, here = d0[i], seg_off[i]
        while live and live[-1][1] >= cut:            # whole interval diverges: its tokens end their subtree here
            j, lo, _ = live.pop()
            brk[j].append((lo, here))
        if live and live[-1][2] > cut:                # partial: depths >= cut end here
            live[-1][2] = cut
            brk[live[-1][0]].append((cut, here))
        runs = []
        for j, lo, hi in live:
            b, e = seg_off[j] + lo - d0[j], seg_off[j] + hi - d0[j]
            if runs and runs[-1][1] == b:
                runs[-1] = (runs[-1][0], e)
            else:
                runs.append((b, e))
        path_runs.append(runs)
        if cut > 0:
            j, lo, hi = live[-1]
            parent[i] = seg_off[j] + (cut - 1) - d0[j]
        if lens[i] > cut:
            live.append([i, cut, lens[i]])
    for j, lo, _ in live:
        brk[j].append((lo, T))
    ptr = [0]
    bd: List[int] = []
    be: List[int] = []
    for j in range(M):
        ent = brk[j][::-1]                            # closing order is deepest first
        # merge duplicates of the same depth (a truncated-to-empty interval closes twice at one depth)
        clean: List[Tuple[int, int]] = []
        for dep, end in ent:
            if clean and clean[-1][0] == dep:
                continue
            clean.append((dep, end))
        if not clean:                                 # empty segment (leaf equal to a prefix of the previous): keep one dummy
            clean = [(d0[j], seg_off[j])]
        bd += [c[0] for c in clean]; be += [c[1] for c in clean]
        ptr.append(len(bd))
    return SegmentPlan(M, T, np.asarray(seg_off, np.int32), np.asarray(d0, np.int32), np.asarray(parent, np.int32),
                       np.asarray(ptr, np.int32), np.asarray(bd, np.int32), np.asarray(be, np.int32), path_runs)


def pad_plan(plan: SegmentPlan, multiple: int = 256) -> SegmentPlan:
    """Append one detached chain of filler tokens (depths 0.., no ancestors, no descendants, in no leaf's path) so
    that the packed length is a multiple of `multiple`: hipBLASLt's projection GEMMs lose 5-15 % on a ragged row
    count (scripts/gemm_shapes.py).  Real tokens are untouched — no real token's subtree reaches past the old T —
    and the filler rows receive zero gradient.  `n_real` keeps the number of real segments."""
    pad = (-plan.T) % multiple
    if pad == 0:
        out = SegmentPlan(plan.M, plan.T, plan.seg_off, plan.seg_depth0, plan.parent_of_seg, plan.brk_ptr, plan.brk_depth,
                          plan.brk_end, plan.path_runs)
        out.n_real = plan.M
        return out
    T2 = plan.T + pad
    i32 = np.int32
    out = SegmentPlan(plan.M + 1, T2,
                      np.concatenate([plan.seg_off, np.asarray([T2], i32)]),
                      np.concatenate([plan.seg_depth0, np.asarray([0], i32)]),
                      np.concatenate([plan.parent_of_seg, np.asarray([-1], i32)]),
                      np.concatenate([plan.brk_ptr, np.asarray([plan.brk_ptr[-1] + 1], i32)]),
                      np.concatenate([plan.brk_depth, np.asarray([0], i32)]),
                      np.concatenate([plan.brk_end, np.asarray([T2], i32)]),
                      plan.path_runs + [[]])
    out.n_real = plan.M
    return out


def ktile_qend_host(plan: SegmentPlan, tile: int = KTILE) -> np.ndarray:
    """ktile_qend[j] = max subtree_end over the keys of key tile j, from the plan's closing tables alone (host,
    O(#breaks + #tiles)): subtree_end is piecewise constant over the packed intervals [seg_off[i] + brk_depth[b] -
    seg_depth0[i], next break), so a tile's maximum is the maximum over the intervals that touch it.  Lets the dK/dV
    work units be planned without reading subtree_end back from the device."""
    T = plan.T
    nkt = (T + tile - 1) // tile
    out = np.zeros(nkt, np.int64)
    if T == 0:
        return out.astype(np.int32)
    seg = np.repeat(np.arange(plan.M, dtype=np.int64), np.diff(plan.brk_ptr))
    start = plan.seg_off[seg].astype(np.int64) + plan.brk_depth.astype(np.int64) - plan.seg_depth0[seg].astype(np.int64)
    nxt = np.empty_like(start)
    nxt[:-1] = start[1:]
    last_of_seg = plan.brk_ptr[1:].astype(np.int64) - 1
    nxt[last_of_seg] = plan.seg_off[1:].astype(np.int64)
    keep = nxt > start                                  # empty segments keep a dummy break
    start, nxt, val = start[keep], nxt[keep], plan.brk_end.astype(np.int64)[keep]
    t0, t1 = start // tile, (nxt - 1) // tile
    np.maximum.at(out, t0, val)
    np.maximum.at(out, t1, val)
    inner = t1 - t0 - 1                                 # tiles wholly inside one interval belong to it alone
    has = inner > 0
    if has.any():
        cnt = inner[has]
        first = (t0[has] + 1)
        idx = np.repeat(first - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
        out[idx] = np.maximum(out[idx], np.repeat(val[has], cnt))
    return out.astype(np.int32)


def _merge(ivs):
    """Sorted union of half-open intervals."""
    out = []
    for b, e in sorted(x for x in ivs if x[1] > x[0]):
        if out and b <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], e))
        else:
            out.append((b, e))
    return out


def _intersect(a, b):
    a, b = _merge(a), _merge(b)
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            out.append((lo, hi))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def _subtract(a, b):
    """a minus b (both merged, sorted)."""
    out = []
    for lo, hi in a:
        cur = lo
        for b0, b1 in b:
            if b1 <= cur or b0 >= hi:
                continue
            if b0 > cur:
                out.append((cur, b0))
            cur = max(cur, b1)
        if cur < hi:
            out.append((cur, hi))
    return out


def plan_qtile_runs(plan: SegmentPlan, tile: int = QTILE):
    """Key runs each query tile of `tile` packed rows must visit: (run_ptr int32 [nqt+1],
    runs int32 [nruns,4] = {key_begin, key_end, needs_mask, 0}).  A run is flagged maskless only when
    every key in it is an ancestor of every row of the tile."""
    T = plan.T
    nqt = (T + tile - 1) // tile
    seg_off = plan.seg_off
    starts = np.arange(nqt, dtype=np.int64) * tile
    ends = np.minimum(starts + tile, T)
    first = np.searchsorted(seg_off, starts, side="right") - 1
    last = np.searchsorted(seg_off, ends - 1, side="right") - 1
    run_ptr = [0]
    out: List[Tuple[int, int, int, int]] = []
    for qt in range(nqt):
        q0, q1, i0, i1 = int(starts[qt]), int(ends[qt]), int(first[qt]), int(last[qt])
        if i0 == i1:
            for b, e in plan.path_runs[i0]:
                out.append((b, e, 0, 0))
            s = int(seg_off[i0])
            if s < q0:
                if out and len(out) > run_ptr[-1] and out[-1][1] == s and out[-1][2] == 0:
                    out[-1] = (out[-1][0], q0, 0, 0)
                else:
                    out.append((s, q0, 0, 0))
            out.append((q0, q1, 1, 0))
        else:
            # the tile straddles segments i0..i1: keys that are ancestors of EVERY row of the tile (the intersection of the
            # segments' ancestor sets, below the tile) need no mask; the rest of the union does
            common = None
            ivs = []
            for i in range(i0, i1 + 1):
                s, e = int(seg_off[i]), min(q1, int(seg_off[i + 1]))
                if e <= s:
                    continue                                  # empty segment: no rows here
                anc = list(plan.path_runs[i])
                if s < q0:
                    anc.append((s, q0))                       # the segment's own rows above the tile
                common = anc if common is None else _intersect(common, anc)
                ivs.extend(plan.path_runs[i]); ivs.append((s, e))
            common = _merge([(b, min(e, q0)) for b, e in (common or []) if b < q0])
            for b, e in common:
                out.append((b, e, 0, 0))
            for b, e in _subtract(_merge(ivs), common):
                out.append((b, e, 1, 0))
        run_ptr.append(len(out))
    return np.asarray(run_ptr, np.int32), np.asarray(out, np.int32).reshape(-1, 4)


def plan_dkv_units(ktile_qend: np.ndarray, Tk: int, Tq: int, q_offset: int = 0, Hkv: int = 8, tile: int = KTILE,
                   waves_per_cu: int = 6, n_cu: int = 256, min_tiles: int = 8):
    """Balanced work units of the dK/dV sweep.  Key tile j must visit the 64-row query tiles of
    [max(key0, q_offset), ktile_qend[j]); root-side key tiles are seen by every query below them, so
    their sweep is hundreds of tiles long while leaf-side tiles need a handful.  Tiles longer than the
    cap are cut into equal units; units of a cut tile write fp32 slabs that a finalize launch sums in
    unit order.  Returns (units int32 [n,4] = {key tile, q_begin, q_end, slab|-1} heaviest first,
    splits int32 [m,4] = {key tile, first slab, n slabs, 0}, n_slabs)."""
    nkt = ktile_qend.shape[0]
    k0 = np.arange(nkt, dtype=np.int64) * tile
    qbeg = np.maximum(k0, q_offset)
    qend = np.minimum(ktile_qend.astype(np.int64), q_offset + Tq)
    ntile = np.maximum(0, -(-(qend - qbeg) // 64))
    total = int(ntile.sum())
    target = max(1, (waves_per_cu * n_cu) // max(Hkv, 1))
    cap = max(min_tiles, -(-total // target))
    nsplit = np.maximum(1, -(-ntile // cap))
    per = -(-ntile // nsplit)                                    # query tiles per unit
    kt_of = np.repeat(np.arange(nkt, dtype=np.int64), nsplit)
    first_unit = np.cumsum(nsplit) - nsplit
    j = np.arange(kt_of.shape[0], dtype=np.int64) - first_unit[kt_of]
    ub = qbeg[kt_of] + j * per[kt_of] * 64
    ue = np.minimum(qend[kt_of], ub + per[kt_of] * 64)
    is_split = nsplit[kt_of] > 1
    slab = np.where(is_split, np.cumsum(is_split) - 1, -1)
    units = np.stack([kt_of, ub, np.maximum(ue, ub), slab], axis=1)
    order = np.argsort(-(units[:, 2] - units[:, 1]), kind="stable")       # heaviest first: short tail
    split_kt = np.flatnonzero(nsplit > 1)
    first_slab = (np.cumsum(np.where(nsplit > 1, nsplit, 0)) - np.where(nsplit > 1, nsplit, 0))[split_kt]
    splits = np.stack([split_kt, first_slab, nsplit[split_kt], np.zeros_like(split_kt)], axis=1) if split_kt.size else np.zeros((0, 4), np.int64)
    return units[order].astype(np.int32), splits.astype(np.int32), int(is_split.sum())
