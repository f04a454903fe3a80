"""HuggingFace `AttentionInterface` plug-in: lets an UNMODIFIED transformers causal LM — and hence the
reference's own `tree_training_engine.py` with its `DynamicCache` — run its attention on the HIP kernels.

The reference selects the attention backend with ``attn_implementation`` (run.py:170,199-204; default
``flash_attention_3``, CUDA-only).  After ``register()`` the name ``"dta_mi355x"`` is valid there.  The
backend receives ``query [1,Hq,B,D]`` and ``key/value [1,Hkv,S,D]`` already concatenated with the
prefix (hf: models/qwen3/modeling_qwen3.py:262-280) — exactly the stack form of the kernels with
``q_offset = S - B`` (rectangular causal; tree_training_engine.py:171-186), read in place through strides.
"""
from __future__ import annotations

from . import ops

NAME = "dta_mi355x"


def dta_attention_forward(module, query, key, value, attention_mask=None, dropout: float = 0.0, scaling=None,
                          sliding_window=None, **kwargs):
    if query.shape[0] != 1:
        raise ValueError("dta_mi355x attention handles one sequence per call (batch 1), as the reference engine issues them")
    if sliding_window is not None or dropout:
        raise ValueError("dta_mi355x attention: sliding window / dropout are not part of this path")
    _, Hq, B, D = query.shape
    S = key.shape[2]
    q = query[0].transpose(0, 1)          # [B, Hq, D] view
    k = key[0].transpose(0, 1)            # [S, Hkv, D] view
    v = value[0].transpose(0, 1)
    scale = D ** -0.5 if scaling is None else scaling
    out = ops.tree_attention(q, k, v, ops.stack_meta(S - B), scale)      # [B, Hq, D] contiguous
    return out.unsqueeze(0), None


def register() -> str:
    """Registers the backend (and a mask function that builds no [B,S] mask tensor) with transformers."""
    from transformers import AttentionInterface
    from transformers.masking_utils import AttentionMaskInterface, flash_attention_mask
    AttentionInterface.register(NAME, dta_attention_forward)
    AttentionMaskInterface.register(NAME, flash_attention_mask)
    return NAME
