"""Qwen3-family decoder over a *packed trie* (one pass over all tree tokens).

The reference drives an unmodified HuggingFace causal LM once per trie segment
(tree_training_engine.py:182-186, 248-252, 351-353).  Here the whole trie is one batch of T packed
tokens: position = trie depth, attention = `ops.tree_attention` (HIP), GEMMs = hipBLASLt through
torch.  `Qwen3TreeLM` mirrors the HF module/parameter tree (``model.embed_tokens.weight``,
``model.layers.N.self_attn.q_proj.weight`` …, tied head) so that

* gradients compare name by name with grad/Qwen3-0.6B-TB-vs-DB-bf16.txt (310 tensors), and
* `packed_hidden_states` also accepts a HuggingFace Qwen2/Qwen3 ``*ForCausalLM`` by duck typing —
  its own ``nn.Parameter`` objects are used, so ``param.grad`` lands where the training loop expects.
"""
from __future__ import annotations

import os

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

FUSE_PROJECTIONS = os.environ.get("DTA_FUSE_PROJ", "1") != "0"      # diagnostic A/B switch


class _Norm(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))


class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        H, Hq, Hkv, D = c.hidden_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = _Lin(H, Hq * D), _Lin(H, Hkv * D), _Lin(H, Hkv * D), _Lin(Hq * D, H)
        self.q_norm, self.k_norm = _Norm(D), _Norm(D)


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gate_proj, self.up_proj, self.down_proj = _Lin(c.hidden_size, c.intermediate_size), _Lin(c.hidden_size, c.intermediate_size), _Lin(c.intermediate_size, c.hidden_size)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn, self.mlp = _Attn(c), _MLP(c)
        self.input_layernorm, self.post_attention_layernorm = _Norm(c.hidden_size), _Norm(c.hidden_size)


class _Body(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embed_tokens = nn.Embedding(c.vocab_size, c.hidden_size)
        self.layers = nn.ModuleList(_Layer(c) for _ in range(c.num_hidden_layers))
        self.norm = _Norm(c.hidden_size)


def make_config(d: dict) -> SimpleNamespace:
    c = SimpleNamespace(**d)
    if not hasattr(c, "head_dim"):
        c.head_dim = c.hidden_size // c.num_attention_heads
    c.rms_norm_eps = getattr(c, "rms_norm_eps", 1e-6)
    c.rope_theta = getattr(c, "rope_theta", 1e6)
    return c


class Qwen3TreeLM(nn.Module):
    """Parameter container with the HF names; tied LM head (no lm_head parameter)."""

    def __init__(self, config):
        super().__init__()
        self.config = make_config(config) if isinstance(config, dict) else config
        self.model = _Body(self.config)

    @property
    def device(self):
        return self.model.embed_tokens.weight.device

    @torch.no_grad()
    def load_named(self, weights: dict):
        own = dict(self.named_parameters())
        for k, p in own.items():
            p.copy_(weights[k])
        return self

    @classmethod
    def from_named(cls, config, weights: dict, device, dtype):
        m = cls(config)
        m.load_named(weights)
        return m.to(device=device, dtype=dtype)


def _cfg_of(model):
    c = model.config
    D = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    theta = getattr(c, "rope_theta", None)
    if theta is None:
        rp = getattr(c, "rope_parameters", None) or {}
        theta = rp.get("rope_theta", 1e6) if isinstance(rp, dict) else 1e6
    return c.num_attention_heads, c.num_key_value_heads, D, float(getattr(c, "rms_norm_eps", 1e-6)), float(theta)


def _layer_forward(layer, res, delta, cos_sin, attn, Hq, Hkv, D, eps):
    """One decoder layer over the packed rows.  The hidden state enters as (residual stream, pending update)
    so that each residual add is fused into the RMSNorm that follows it.  hipBLASLt GEMMs through torch;
    everything between them is a HIP kernel of this package."""
    T = res.shape[0]
    a = layer.self_attn
    res, h = ops.add_rms_norm(res, delta, layer.input_layernorm.weight, eps)
    bq, bk, bv = (getattr(m_, "bias", None) for m_ in (a.q_proj, a.k_proj, a.v_proj))
    if FUSE_PROJECTIONS:
        # one projection GEMM for q,k,v (and one for gate,up below): the weights stay separate parameters with
        # their HF names; stacking them is three plain copies whose backward hands out gradient row slices
        qkv = ops.linear(h, ops.stack_rows(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight),
                       torch.cat([bq, bk, bv]) if bq is not None else None).view(T, Hq + 2 * Hkv, D)
        qn = getattr(a, "q_norm", None); kn = getattr(a, "k_norm", None)
        q, k, v = ops.qkv_prep(qkv, qn.weight if qn is not None else None, kn.weight if kn is not None else None, cos_sin, eps, Hq, Hkv)
    else:
        q = ops.linear(h, a.q_proj.weight, bq).view(T, Hq, D)
        k = ops.linear(h, a.k_proj.weight, bk).view(T, Hkv, D)
        v = ops.linear(h, a.v_proj.weight, bv).view(T, Hkv, D)
        qn = getattr(a, "q_norm", None); kn = getattr(a, "k_norm", None)
        q = ops.qk_norm_rope(q, qn.weight if qn is not None else None, cos_sin, eps)
        k = ops.qk_norm_rope(k, kn.weight if kn is not None else None, cos_sin, eps)
    o = attn(q, k, v)                   # ops.tree_attention over a packed trie, or ops.stack_attention over the KV stack
    attn_out = ops.linear(o.reshape(T, Hq * D), a.o_proj.weight)
    res, h = ops.add_rms_norm(res, attn_out, layer.post_attention_layernorm.weight, eps)
    m = layer.mlp
    if FUSE_PROJECTIONS:
        act = ops.swiglu_fused(ops.linear(h, ops.stack_rows(m.gate_proj.weight, m.up_proj.weight)))
    else:
        act = ops.swiglu(ops.linear(h, m.gate_proj.weight), ops.linear(h, m.up_proj.weight))
    return res, ops.linear(act, m.down_proj.weight)


class _LayerRecompute(torch.autograd.Function):
    """Per-layer activation recomputation (for tries whose activations do not fit): the forward runs the layer
    without a graph and keeps only its two inputs — plus, when `keep_attn`, the attention output and lse, so that the
    recomputation in the backward skips the forward attention kernel (a fifth of the attention time at depth).  The
    backward re-runs the layer with a graph and back-propagates through it; parameter gradients accumulate into
    `.grad` directly (nested autograd), input gradients are returned."""

    @staticmethod
    def forward(ctx, fn, keep_attn, res, delta):
        ctx.fn, ctx.has_delta = fn, delta is not None
        items = []
        with torch.no_grad():
            if keep_attn:
                with ops.AttentionTape("record", items):
                    out = fn(res, delta)
            else:
                out = fn(res, delta)
        ctx.items = items
        ctx.save_for_backward(res, delta if delta is not None else res)
        return out

    @staticmethod
    def backward(ctx, g_res, g_delta):
        res, delta = ctx.saved_tensors
        r = res.detach().requires_grad_(True)
        d = delta.detach().requires_grad_(True) if ctx.has_delta else None
        with torch.enable_grad():
            if ctx.items:
                with ops.AttentionTape("replay", ctx.items):
                    o_res, o_delta = ctx.fn(r, d)
            else:
                o_res, o_delta = ctx.fn(r, d)
        ctx.items = None
        torch.autograd.backward((o_res, o_delta), (g_res, g_delta))
        return None, None, r.grad, (d.grad if d is not None else None)


def packed_hidden_states(model, tokens: torch.Tensor, depth: torch.Tensor, meta, checkpoint_layers: bool = False,
                         attn_keep_bytes: int = 0, attn_of_layer=None, full_layers=0, kept_out=None, embed=None) -> torch.Tensor:
    """Final-norm hidden states [T, hidden] of the packed tokens.  `model` is a Qwen3TreeLM or an HF
    Qwen2/Qwen3 *ForCausalLM (duck-typed).  `checkpoint_layers`: recompute each layer in the backward, except the first
    `full_layers`, which keep their activations like the plain pass — an int, or a plan `bytes kept by layer 0 -> number of
    layers` that is asked once layer 0 has run in full and its footprint has been measured (the number lands in `kept_out`);
    `attn_keep_bytes`: HBM budget for attention outputs kept across that recomputation (layers are served first to last).
    `attn_of_layer(l)` -> callable (q, k, v) -> o replaces the packed tree attention (the block-wise engine passes the
    stack form bound to layer l's KV stack; `meta` is unused then); `embed(tokens)` replaces the plain embedding lookup (the
    block-wise engine routes the rows' gradients into its fp32 sink instead of a dense [vocab, hidden] gradient per block)."""
    Hq, Hkv, D, eps, theta = _cfg_of(model)
    body = model.model
    res, delta = (embed(tokens) if embed is not None else F.embedding(tokens, body.embed_tokens.weight)), None
    cos_sin = ops.rope_cos_sin(depth, D, theta)
    per_layer = tokens.shape[0] * Hq * (D * res.element_size() + 4)             # out + lse of one layer
    n_full = full_layers if isinstance(full_layers, int) else 1
    for li, layer in enumerate(body.layers):
        attn = attn_of_layer(li) if attn_of_layer is not None else (lambda q, k, v: ops.tree_attention(q, k, v, meta))
        if li == 0 and callable(full_layers) and checkpoint_layers and torch.is_grad_enabled() and res.is_cuda:
            m0 = torch.cuda.memory_allocated(res.device)
            res, delta = _layer_forward(layer, res, delta, cos_sin, attn, Hq, Hkv, D, eps)
            n_full = int(full_layers(torch.cuda.memory_allocated(res.device) - m0))
            continue
        if checkpoint_layers and li >= n_full and torch.is_grad_enabled():
            keep = attn_of_layer is None and attn_keep_bytes >= per_layer
            if keep:
                attn_keep_bytes -= per_layer
            fn = (lambda layer_, attn_: lambda r_, d_: _layer_forward(layer_, r_, d_, cos_sin, attn_, Hq, Hkv, D, eps))(layer, attn)
            res, delta = _LayerRecompute.apply(fn, keep, res, delta)
        else:
            res, delta = _layer_forward(layer, res, delta, cos_sin, attn, Hq, Hkv, D, eps)
    if kept_out is not None:
        kept_out.append(n_full if checkpoint_layers else len(body.layers))
    return ops.add_rms_norm(res, delta, body.norm.weight, eps)[1]


def head_weight(model) -> torch.Tensor:
    lm = getattr(model, "lm_head", None)
    return lm.weight if lm is not None else model.model.embed_tokens.weight
