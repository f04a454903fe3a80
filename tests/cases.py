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



# Qwen2-style geometry: q/k/v biases, no per-head q/k RMSNorm, head_dim = hidden / heads = 128 (INTEGRATION.md claims Qwen2 support)
QWEN2_TINY = dict(vocab_size=512, hidden_size=256, intermediate_size=320, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=128, rms_norm_eps=1e-6, rope_theta=1000000.0)


def qwen2_weights(cfg: dict = None, seed: int = 5, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights with the HF Qwen2 parameter names (tied head)."""
    cfg = cfg or QWEN2_TINY
    g = torch.Generator().manual_seed(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    Hq, Hkv, D = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    w = {"model.embed_tokens.weight": torch.randn(V, H, generator=g) * std}
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        for nm, o in (("q", Hq * D), ("k", Hkv * D), ("v", Hkv * D)):
            w[p + f"self_attn.{nm}_proj.weight"] = torch.randn(o, H, generator=g) * std
            w[p + f"self_attn.{nm}_proj.bias"] = torch.randn(o, generator=g) * 0.1
        w[p + "self_attn.o_proj.weight"] = torch.randn(H, Hq * D, generator=g) * std
        w[p + "mlp.gate_proj.weight"] = torch.randn(I, H, generator=g) * std
        w[p + "mlp.up_proj.weight"] = torch.randn(I, H, generator=g) * std
        w[p + "mlp.down_proj.weight"] = torch.randn(H, I, generator=g) * std
        w[p + "input_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        w[p + "post_attention_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    w["model.norm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    return w


def hf_qwen2(weights=None, cfg: dict = None):
    """transformers.Qwen2ForCausalLM of QWEN2_TINY with the seeded weights (fp32, eager attention)."""
    import transformers
    cfg = cfg or QWEN2_TINY
    c = transformers.Qwen2Config(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                                 num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                                 num_key_value_heads=cfg["num_key_value_heads"], tie_word_embeddings=True, max_position_embeddings=40960,
                                 rms_norm_eps=cfg["rms_norm_eps"], rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"]})
    c._attn_implementation = "eager"
    m = transformers.Qwen2ForCausalLM(c)
    w = weights or qwen2_weights(cfg)
    missing, unexpected = m.load_state_dict({**w, "lm_head.weight": w["model.embed_tokens.weight"]}, strict=False)
    assert not unexpected and all("rotary" in k for k in missing), (missing, unexpected)
    return m.float().train()


def hf_qwen3(cfg: dict, weights: Dict[str, torch.Tensor]):
    """transformers.Qwen3ForCausalLM of a TINY_CFGS geometry with the given named weights (fp32, eager attention)."""
    import transformers
    c = transformers.Qwen3Config(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                                 num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                                 num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"], tie_word_embeddings=True,
                                 max_position_embeddings=40960, rms_norm_eps=cfg["rms_norm_eps"],
                                 rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"]})
    c._attn_implementation = "eager"
    m = transformers.Qwen3ForCausalLM(c)
    missing, unexpected = m.load_state_dict({**weights, "lm_head.weight": weights["model.embed_tokens.weight"]}, strict=False)
    assert not unexpected and all("rotary" in k for k in missing), (missing, unexpected)
    return m.float().train()


QWEN2_DATA = {"kind": "tau2", "seed": 6, "V": 512, "G": 3, "sys_len": 20, "turns": 3, "lo": 5, "hi": 15, "cap": 64}


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
