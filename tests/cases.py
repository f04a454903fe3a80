"""Test-case tables shared by the tests, __graft_entry__.smoke() and scripts/make_golden.py: tiny model configurations and the
seeded trie / balancer / engine case lists (the generators themselves are in dynamictreeattn_amd/synth.py)."""
from __future__ import annotations

from typing import Dict, List

import torch

TINY_CFGS: Dict[str, dict] = {
    "d16": dict(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                num_attention_heads=4, num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-6, rope_theta=1000000.0),
    "d128": dict(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                 num_attention_heads=4, num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-6, rope_theta=1000000.0),
}



def trie_cases() -> List[dict]:
    cases = [
        {"kind": "literal", "seqs": [[1, 2, 3]]},                                   # single sequence
        {"kind": "literal", "seqs": [[1, 2, 3], [4, 5], [6]]},                      # empty LCPs
        {"kind": "literal", "seqs": [[1, 2, 3], [1, 2, 3], [1, 2, 3]]},             # duplicates only
        {"kind": "literal", "seqs": [[7], [7, 8], [7, 8, 9], [7, 8, 9, 10]]},       # pure chain
        {"kind": "literal", "seqs": [[1, 2, 3, 4, 5], [1, 2, 3, 9], [1, 2, 8], [1, 7]]},   # trie.py smoke shape
        {"kind": "literal", "seqs": [[2, 1], [1, 2], [1], [1, 2, 3], [2], [1, 2]]},  # prefixes + dups, unsorted
        {"kind": "chain", "seed": 3, "n_seq": 6, "step": 5},
        {"kind": "config1", "seed": 0, "V": 1000, "prompt": 24, "rollouts": 4, "gen": 16},
        {"kind": "tau2", "seed": 0},                                                # full-size SURVEY call
        {"kind": "tau2", "seed": 1},
        {"kind": "tau2", "seed": 5, "G": 16, "turns": 12, "lo": 400, "hi": 1800},   # heavier variant, hits the cap
        {"kind": "tau2", "seed": 2, "V": 50, "G": 5, "sys_len": 30, "turns": 4, "lo": 3, "hi": 12, "cap": 60},
        {"kind": "wide", "seed": 1, "V": 1000, "root": 16, "branches": 12, "depth": 80},
    ]
    for s in range(40):
        cases.append({"kind": "random_tree", "seed": 100 + s, "n_seq": 3 + (s * 7) % 40, "max_len": 4 + (s * 5) % 60,
                      "alphabet": 2 + s % 3, "dup": s % 4})
    return cases


def balancer_cases() -> List[dict]:
    return [
        {"kind": "tau2", "seed": 0},
        {"kind": "tau2", "seed": 3, "G": 16, "turns": 4},
        {"kind": "random_tree", "seed": 11, "n_seq": 40, "max_len": 200, "alphabet": 2, "dup": 3},
        {"kind": "random_tree", "seed": 12, "n_seq": 17, "max_len": 90, "alphabet": 3, "dup": 0},
        {"kind": "wide", "seed": 1, "V": 1000, "root": 64, "branches": 24, "depth": 400},
    ]


def engine_cases() -> Dict[str, dict]:
    return {
        "d16_tree": {"cfg": "d16", "wseed": 0,
                     "data": {"kind": "random_tree", "seed": 21, "n_seq": 7, "max_len": 40, "alphabet": 2, "dup": 1}},
        "d128_minitau": {"cfg": "d128", "wseed": 1,
                         "data": {"kind": "tau2", "seed": 4, "V": 512, "G": 3, "sys_len": 20, "turns": 3, "lo": 5, "hi": 15, "cap": 64}},
        "d128_tree": {"cfg": "d128", "wseed": 2,
                      "data": {"kind": "random_tree", "seed": 22, "n_seq": 9, "max_len": 150, "alphabet": 2, "dup": 2}},
    }


def attn_inputs(start: int, B: int, Hq: int = 16, Hkv: int = 8, D: int = 128, seed: int = 0):
    """Seeded fp32 q [1,Hq,B,D], k/v [1,Hkv,start+B,D], dout [1,B,Hq,D] for the attention operator."""
    g = torch.Generator().manual_seed(seed * 7919 + start * 31 + B)
    q = torch.randn(1, Hq, B, D, generator=g)
    k = torch.randn(1, Hkv, start + B, D, generator=g)
    v = torch.randn(1, Hkv, start + B, D, generator=g)
    do = torch.randn(1, B, Hq, D, generator=g)
    return q, k, v, do
