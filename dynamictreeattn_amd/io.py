"""Interchange formats of the reference's experiment scripts, so that batches, DP bins,
stats and gradient dumps written by either side can be read by the other:

* batches: ``torch.save(List[LongTensor])`` `.pt` files (run.py:134-137, run_all.py:21-31);
* DP bins: ``{name}_bin{k}.pt`` (data_parallel.py:195-199);
* stats: one JSON object per line with the keys of trie.py:18-53 + ``n_sequences, n_tokens, loss, time,
  name`` (run_all.py:162-165) — what ``TreeTimeModel.add_data`` and exp/calc_time.py consume;
* gradient dumps ``{param_name: tensor}`` (run.py:154-161) and the comparison table of
  exp/compare_grads.py:42-76 (|Δg|/|g| per parameter, sorted descending)."""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional

import torch


def load_batch(path: str) -> List[torch.Tensor]:
    data = torch.load(path, map_location="cpu", weights_only=True)
    return [t.to(torch.long) for t in data]


def load_folder(folder: str):
    return [(f[:-3], load_batch(os.path.join(folder, f))) for f in sorted(os.listdir(folder)) if f.endswith(".pt")]


def save_bins(out_folder: str, name: str, inputs: List[torch.Tensor], bins: List[List[int]]) -> List[str]:
    os.makedirs(out_folder, exist_ok=True)
    paths = []
    for k, b in enumerate(bins):
        p = os.path.join(out_folder, f"{name}_bin{k}.pt")
        torch.save([inputs[i] for i in b], p)
        paths.append(p)
    return paths


def write_stats(path: str, rows: Iterable[dict]) -> None:
    with open(path, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


def read_stats(path: str) -> List[dict]:
    with open(path) as f:
        return [json.loads(line) for line in f if line.strip()]


def dp_throughput(rows: List[dict]) -> float:
    """Σ n_tokens / Σ_calls max_bins(time) with `name` = `<call>_bin<k>` (exp/calc_time.py:24-42)."""
    per_call: Dict[str, float] = {}
    tokens = 0
    for r in rows:
        call = r["name"].rsplit("_bin", 1)[0] if "_bin" in r["name"] else r["name"]
        per_call[call] = max(per_call.get(call, 0.0), r["time"])
        tokens += r["n_tokens"]
    return tokens / sum(per_call.values())


def save_gradients(model, path: str) -> None:
    torch.save({n: (p.grad.detach().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}, path)


def compare_grads(base: Dict[str, Optional[torch.Tensor]], exp: Dict[str, Optional[torch.Tensor]], eps: float = 1e-6):
    """[(name, |Δg|/|g|, |g_base|, |g_exp|)] sorted by the ratio, descending (exp/compare_grads.py:42-62)."""
    rows = []
    for n in sorted(set(base) & set(exp)):
        if base[n] is None or exp[n] is None:
            continue
        g0, g1 = base[n].float(), exp[n].float()
        rows.append((n, float(torch.norm(g1 - g0) / (torch.norm(g0) + eps)), float(torch.norm(g0)), float(torch.norm(g1))))
    rows.sort(key=lambda r: r[1], reverse=True)
    return rows


def format_compare_table(rows, n_common: Optional[int] = None, missing_in_exp: Iterable[str] = (), missing_in_base: Iterable[str] = ()) -> str:
    """The text exp/compare_grads.py:26-76 writes: optional missing-key warnings, "Comparing N common parameters" (N counts the
    common KEYS, entries that are None on either side included), header, rule, one row per compared parameter."""
    lines = []
    missing_in_exp, missing_in_base = sorted(missing_in_exp), sorted(missing_in_base)
    if missing_in_exp:
        lines.append(f"[Warning] Missing in exp-grad ({len(missing_in_exp)}):")
        lines += [f"  {k}" for k in missing_in_exp]
    if missing_in_base:
        lines.append(f"[Warning] Missing in baseline-grad ({len(missing_in_base)}):")
        lines += [f"  {k}" for k in missing_in_base]
    head = f"{'Parameter':60s} {'|Δg|/|g|':>12s} {'|g_baseline|':>12s} {'|g_exp|':>12s}"
    lines += [f"\nComparing {len(rows) if n_common is None else n_common} common parameters\n", head, "-" * 104]
    lines += [f"{n:60s} {r:12.4e} {a:12.4e} {b:12.4e}" for n, r, a, b in rows]
    return "\n".join(lines)


def compare_grads_report(base: Dict[str, Optional[torch.Tensor]], exp: Dict[str, Optional[torch.Tensor]]) -> str:
    """exp/compare_grads.py as one call on two loaded dumps."""
    return format_compare_table(compare_grads(base, exp), len(set(base) & set(exp)), set(base) - set(exp), set(exp) - set(base))


def parse_compare_table(text: str):
    """Rows [(name, ratio, |g_base|, |g_exp|)] of a table in exp/compare_grads.py's format, e.g. the reference's recorded
    grad/Qwen3-0.6B-TB-vs-DB-bf16.txt."""
    rows, body = [], False
    for ln in text.split("\n"):
        if ln.startswith("-" * 20):
            body = True
        elif body and ln.strip():
            name, ratio, a, b = ln.split()
            rows.append((name, float(ratio), float(a), float(b)))
    return rows


STATS_KEYS = {                                   # run.py:57-60 / 110-112 + run_all.py:60 / 98 (the order json.dumps writes them in)
    "tree_forward": ("n_leaf_sequences", "n_tree_tokens", "sum_prefix_len", "sum_depth", "n_sequences", "n_tokens", "time", "name"),
    "tree_backward": ("n_leaf_sequences", "n_tree_tokens", "sum_prefix_len", "sum_depth", "n_f1_tokens", "n_sequences", "n_tokens",
                      "loss", "time", "name"),
}


def stats_row(trie, run: str, seconds: float, name: str, loss: Optional[float] = None, block_size: Optional[int] = None) -> dict:
    """One stats row of `run` ∈ {tree_forward, tree_backward} for a TokenTrie in the order it was run in — the dict run.py
    returns and run_all.py:162-165 writes as a JSONL line; `TreeTimeModel.add_data` and exp/calc_time.py read these keys."""
    if run == "tree_forward":
        st = trie.get_stats(mode="forward")
    elif run == "tree_backward":
        st = trie.get_stats(mode="backward", block_size=block_size)
        st["loss"] = float(loss if loss is not None else 0.0)
    else:
        raise ValueError(f"Unsupported run: {run}")
    st["time"] = float(seconds)
    st["name"] = name
    assert tuple(st) == STATS_KEYS[run], tuple(st)
    return st
