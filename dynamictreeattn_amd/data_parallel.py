"""Partition one batch of sequences into K data-parallel bins (data_parallel.py:8-107).

``LB_by_n_tokens`` — longest-processing-time by token count; ``LB_by_TM`` — greedy by predicted
tree time; ``LB_by_DFS_and_TM`` — contiguous ranges of leaves in sorted order, min-max predicted
time by a nested bisection.  All return ``List[List[int]]`` of ORIGINAL sequence ids.
In the reference the bins are written to disk and run one after another on one GPU
(exp/exp_dp.py:43-49); here each rank of a torch.distributed job takes one bin (see dp.py).
"""
from __future__ import annotations

from typing import List, Optional

from .token_trie import TokenTrie
from .trie import CompressedTrie, _get_stats, _get_subtrie


def LB_by_n_tokens(token_seqs, K):
    load = [0] * K
    bins = [[] for _ in range(K)]
    for i in sorted(range(len(token_seqs)), key=lambda j: -len(token_seqs[j])):
        b = min(range(K), key=load.__getitem__)
        bins[b].append(i)
        load[b] += len(token_seqs[i])
    return bins


def pred_time(compressed_trie, time_model, mode: str, block_size: Optional[int] = None) -> float:
    if mode == "forward":
        _, lens, lcps = compressed_trie.get_order_forward()
    elif mode == "backward":
        _, lens, lcps = compressed_trie.get_order_backward()
    else:
        raise ValueError(f"Unsupported mode: {mode}")
    return time_model.pred(_get_stats(lens, lcps, mode, block_size))


def get_original_bins(token_trie: TokenTrie, leaf_bins: List[List[int]]) -> List[List[int]]:
    return [[att["_sequence_batch_id"] for leaf in lb for att, _ in token_trie.attach_lists[leaf]] for lb in leaf_bins]


def _leaf_bins_by_TM(lens, lcps, time_model, K, mode, block_size):
    trie = CompressedTrie(lens, lcps)
    leaf_bins = [[] for _ in range(K)]
    cost = [0.0] * K
    for leaf in range(len(lens)):
        b = min(range(K), key=cost.__getitem__)
        leaf_bins[b].append(leaf)
        cost[b] = pred_time(_get_subtrie(trie, leaf_bins[b]), time_model, mode, block_size)
    return leaf_bins


def _split_under_limit(trie, n_leaves, K, mode, block_size, lo_bound, hi_bound, time_model, limit):
    """Greedy left-to-right cut of the sorted leaves into ranges whose predicted time stays <= limit;
    returns the range starts (more than K of them = the limit is too small).  data_parallel.py:58-79"""
    cuts = []
    start = 0
    while start < n_leaves:
        cuts.append(start)
        if len(cuts) > K:
            break
        lo = max(lo_bound[len(cuts)] - 1, start)
        hi = hi_bound[len(cuts)] - 1
        while lo < hi:
            mid = (lo + hi + 1) // 2
            t = pred_time(_get_subtrie(trie, set(range(start, mid + 1))), time_model, mode, block_size)
            if t <= limit:
                lo = mid
            else:
                hi = mid - 1
        start = lo + 1
    return cuts


def _leaf_bins_by_DFS_and_TM(lens, lcps, time_model, K, mode, block_size):
    n = len(lens)
    trie = CompressedTrie(lens, lcps)
    hi = float(pred_time(trie, time_model, mode, block_size))
    lo = hi / K
    eps = hi * 1e-4
    lo_bound = [0] * (K + 1)
    hi_bound = [n] * (K + 1)
    while hi - lo > eps:
        mid = (lo + hi) / 2.0
        cuts = _split_under_limit(trie, n, K, mode, block_size, lo_bound, hi_bound, time_model, mid)
        if len(cuts) <= K:
            hi = mid
            hi_bound[:len(cuts)] = cuts
        else:
            lo = mid + eps
            lo_bound = cuts[:K + 1]
    return [list(range(hi_bound[i], hi_bound[i + 1])) for i in range(K)]


def LB_by_TM(token_seqs, time_model, args):
    tt = TokenTrie(token_seqs)
    return get_original_bins(tt, _leaf_bins_by_TM(tt.lens, tt.lcp_lens, time_model, args.K, args.mode, args.block_size))


def LB_by_DFS_and_TM(token_seqs, time_model, args):
    tt = TokenTrie(token_seqs)
    return get_original_bins(tt, _leaf_bins_by_DFS_and_TM(tt.lens, tt.lcp_lens, time_model, args.K, args.mode, args.block_size))


def eval(token_seqs, bins, time_model, args):
    total = worst = 0.0
    for b in bins:
        if not b:
            continue
        tt = TokenTrie([token_seqs[i] for i in b])
        t = pred_time(CompressedTrie(tt.lens, tt.lcp_lens), time_model, args.mode, args.block_size)
        total += t
        worst = max(worst, t)
    return total, worst


def partition_folder(data_folder: str, out_folder: Optional[str], method: str, K: int, mode: str = "forward", block_size: Optional[int] = None,
                     time_model=None, stats_file: Optional[str] = None, evaluate: bool = False):
    """The body of the reference's CLI (data_parallel.py:153-204): every `<name>.pt` batch of `data_folder` is split into K bins by
    `method` and written as `{name}_bin{k}.pt` (lists of LongTensors, input order inside a bin = the balancer's) into `out_folder`.
    `stats_file`: a stats JSONL (run_all.py:162-165) that trains the time model first; `evaluate`: also return the summed
    predicted (total, max-bin) times.  Returns {"bins": {name: bins}, "total_time": .., "max_time": ..}."""
    import types

    from . import io
    from .tree_time_model import TreeTimeModel
    if mode == "forward":
        block_size = None                                        # data_parallel.py:155-156
    tm_eval = TreeTimeModel()
    if stats_file is not None:
        tm_eval.add_data(io.read_stats(stats_file))
    tm = time_model if time_model is not None else tm_eval
    args = types.SimpleNamespace(K=K, mode=mode, block_size=block_size)
    out = {"bins": {}, "total_time": 0.0, "max_time": 0.0}
    for name, inputs in io.load_folder(data_folder):
        if method == "LB_by_n_tokens":
            bins = LB_by_n_tokens(inputs, K)
        elif method == "LB_by_TM":
            bins = LB_by_TM(inputs, tm, args)
        elif method == "LB_by_DFS_and_TM":
            bins = LB_by_DFS_and_TM(inputs, tm, args)
        else:
            raise ValueError(f"Unsupported method: {method}")
        out["bins"][name] = bins
        if stats_file is not None and evaluate:
            total, worst = eval(inputs, bins, tm_eval, args)
            out["total_time"] += total; out["max_time"] += worst
        if out_folder is not None:
            io.save_bins(out_folder, name, inputs, bins)
    return out
