"""ORACLE — test infrastructure only (import rule: oracle/trie_oracle.py header).

fp32 reference arithmetic for the two attention operators of the hot path:

* ``rect_causal_attention`` — what the reference's model call computes per segment: B new queries at
  stack positions start..start+B-1 against keys 0..start+B-1, query i seeing keys <= start+i
  (tree_training_engine.py:171-186, 339-353 via the third-party transformers attention,
  un-pinned; pinned on the installed 5.15.0 eager path by tests/golden/attn_cases.pt).
* ``tree_attention`` — the same contraction over a DFS-pre-order packed trie: query t sees key s
  iff s is an ancestor-or-self of t, i.e. ``s <= t < subtree_end[s]``.  This is the packed (M2)
  form of "each node attends to its ancestor path" (SURVEY §7); it has no counterpart function in
  the reference and is checked against ``rect_causal_attention`` path by path in the tests.
"""
import torch


def rect_causal_attention(q, k, v, start, scale=None):
    """q [Hq,B,D], k/v [Hkv,start+B,D] -> out [B,Hq,D]; fp32 softmax; GQA by head repetition."""
    Hq, B, D = q.shape
    Hkv = k.shape[0]
    scale = D ** -0.5 if scale is None else scale
    rep = Hq // Hkv
    kk = k.repeat_interleave(rep, dim=0); vv = v.repeat_interleave(rep, dim=0)
    s = torch.matmul(q.float(), kk.float().transpose(1, 2)) * scale
    qpos = torch.arange(start, start + B)[:, None]; kpos = torch.arange(start + B)[None, :]
    s = s.masked_fill(~(kpos <= qpos), float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vv.float()).transpose(0, 1).to(q.dtype)


def tree_mask(subtree_end: torch.Tensor) -> torch.Tensor:
    """[T,T] bool, mask[t,s] = s <= t < subtree_end[s]."""
    T = subtree_end.shape[0]
    idx = torch.arange(T)
    return (idx[None, :] <= idx[:, None]) & (idx[:, None] < subtree_end[None, :])


def tree_attention(q, k, v, subtree_end, scale=None):
    """q [T,Hq,D], k/v [T,Hkv,D] packed in DFS pre-order -> (out [T,Hq,D], lse [T,Hq] natural log)."""
    T, Hq, D = q.shape
    Hkv = k.shape[1]
    scale = D ** -0.5 if scale is None else scale
    rep = Hq // Hkv
    qq = q.float().transpose(0, 1)
    kk = k.float().transpose(0, 1).repeat_interleave(rep, dim=0)
    vv = v.float().transpose(0, 1).repeat_interleave(rep, dim=0)
    s = torch.matmul(qq, kk.transpose(1, 2)) * scale
    s = s.masked_fill(~tree_mask(subtree_end)[None], float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    return torch.matmul(p, vv).transpose(0, 1).to(q.dtype), lse.transpose(0, 1)
