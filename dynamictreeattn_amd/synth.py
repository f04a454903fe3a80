"""Seeded synthetic token batches (the tau2-16k dataset and the model weights of the reference
are absent: SURVEY.md §8c/§8d).  Pure torch-CPU; shared by bench.py, the tests and
scripts/make_golden.py so that every side regenerates identical inputs from a recorded seed.
(The test-case tables — tiny model configs, trie / balancer / engine case lists — live in tests/cases.py.)

Generators return ``List[List[int]]`` (``as_tensors`` converts to the ``List[LongTensor]`` the
reference's ``.pt`` batches hold, run.py:134-137).
"""
from __future__ import annotations

from typing import Dict, List

import torch

QWEN3_0P6B = dict(vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                  num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                  rope_theta=1000000.0)
QWEN3_4B = dict(vocab_size=151936, hidden_size=2560, intermediate_size=9728, num_hidden_layers=36,
                num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                rope_theta=1000000.0)

# the larger models of exp/exp_dp.py:5-10 (geometry only; weights are random-initialised wherever these are used)
QWEN3_8B = dict(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=36,
                num_attention_heads=32, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                rope_theta=1000000.0)
QWEN3_14B = dict(vocab_size=151936, hidden_size=5120, intermediate_size=17408, num_hidden_layers=40,
                 num_attention_heads=40, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                 rope_theta=1000000.0)


def as_tensors(seqs: List[List[int]]) -> List[torch.Tensor]:
    return [torch.tensor(s, dtype=torch.long) for s in seqs]


def _ri(g, lo, hi, n):
    return torch.randint(lo, hi, (n,), generator=g)


def config1(seed: int = 0, V: int = 151936, prompt: int = 256, rollouts: int = 4, gen: int = 256):
    """BASELINE config 1: one prompt, `rollouts` continuations."""
    g = torch.Generator().manual_seed(seed)
    p = _ri(g, 0, V, prompt)
    return [torch.cat([p, _ri(g, 0, V, gen)]).tolist() for _ in range(rollouts)]


def tau2(seed: int = 0, V: int = 151936, G: int = 8, sys_len: int = 2000, turns: int = 6,
         lo: int = 200, hi: int = 900, cap: int = 16384):
    """tau2-16k-shaped call: G rollouts share a system prompt; every turn's cumulative context is
    its own training sequence (SURVEY §8d / Appendix A draw order)."""
    g = torch.Generator().manual_seed(seed)
    sysp = _ri(g, 0, V, sys_len)
    out = []
    for _ in range(G):
        cur = sysp
        for _ in range(turns):
            n = int(_ri(g, lo, hi, 1))
            cur = torch.cat([cur, _ri(g, 0, V, n)])[:cap]
            out.append(cur.tolist())
    return out


def wide(seed: int = 1, V: int = 151936, root: int = 1024, branches: int = 64, depth: int = 16384):
    """BASELINE config 5: `branches` sequences of length `depth` sharing `root` tokens."""
    g = torch.Generator().manual_seed(seed)
    r = _ri(g, 0, V, root)
    return [torch.cat([r, _ri(g, 0, V, depth - root)]).tolist() for _ in range(branches)]


def random_tree(seed: int, n_seq: int, max_len: int, alphabet: int = 3, V: int = 512, p_stop: float = 0.08,
                dup: int = 0, shuffle: bool = True):
    """Small random batch with heavy prefix sharing: tokens drawn from a tiny alphabet near the
    root (so branches collide) and from [0,V) deeper down; optional duplicates."""
    g = torch.Generator().manual_seed(seed)
    seqs = []
    for _ in range(n_seq):
        n = int(_ri(g, 1, max_len + 1, 1))
        head = min(n, int(_ri(g, 1, 8, 1)))
        s = _ri(g, 0, alphabet, head).tolist() + _ri(g, 0, V, n - head).tolist()
        if seqs and float(torch.rand(1, generator=g)) < 0.35:      # extend an existing sequence's prefix
            base = seqs[int(_ri(g, 0, len(seqs), 1))]
            cut = int(_ri(g, 1, len(base) + 1, 1))
            s = base[:cut] + (s[: max(0, n - cut)] if float(torch.rand(1, generator=g)) > p_stop * 4 else [])
        seqs.append(s[:max_len] if s else [0])
    for _ in range(dup):
        seqs.append(list(seqs[int(_ri(g, 0, len(seqs), 1))]))
    if shuffle:
        perm = torch.randperm(len(seqs), generator=g).tolist()
        seqs = [seqs[i] for i in perm]
    return seqs


def chain(seed: int, n_seq: int, step: int, V: int = 512):
    """Every sequence is a proper prefix of the next (leafization folds them into one leaf)."""
    g = torch.Generator().manual_seed(seed)
    full = _ri(g, 0, V, n_seq * step).tolist()
    seqs = [full[: (i + 1) * step] for i in range(n_seq)]
    perm = torch.randperm(n_seq, generator=g).tolist()
    return [seqs[i] for i in perm]


def make_case(case: dict):
    k = case["kind"]
    if k == "config1":
        return config1(case.get("seed", 0), case.get("V", 151936), case.get("prompt", 256), case.get("rollouts", 4), case.get("gen", 256))
    if k == "tau2":
        return tau2(case.get("seed", 0), case.get("V", 151936), case.get("G", 8), case.get("sys_len", 2000),
                    case.get("turns", 6), case.get("lo", 200), case.get("hi", 900), case.get("cap", 16384))
    if k == "wide":
        return wide(case.get("seed", 1), case.get("V", 151936), case.get("root", 1024), case.get("branches", 64), case.get("depth", 16384))
    if k == "random_tree":
        return random_tree(case["seed"], case["n_seq"], case["max_len"], case.get("alphabet", 3), case.get("V", 512),
                           dup=case.get("dup", 0))
    if k == "chain":
        return chain(case["seed"], case["n_seq"], case["step"], case.get("V", 512))
    if k == "literal":
        return [list(s) for s in case["seqs"]]
    raise ValueError(k)
