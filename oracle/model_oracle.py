"""ORACLE — test infrastructure only (see oracle/trie_oracle.py header for the import rule).

Plain-PyTorch CPU restatement of the floating-point side of the hot path:

* ``qwen3_segment`` — the Qwen3 decoder arithmetic the reference reaches through the
  third-party ``transformers`` package (un-pinned by the reference; checked here against the
  installed 5.15.0 ``Qwen3ForCausalLM`` by tests/golden/engine_*.pt):  RMSNorm, per-head q/k
  RMSNorm, RoPE at absolute positions ``start + i``, GQA attention of the new tokens against
  prefix+new keys with an offset causal mask and an fp32 softmax, SwiGLU MLP, tied LM head.
  Call sites in the reference: tree_training_engine.py:182-186, 248-252, 351-353; dense.py:15,39.
* ``logprobs_of`` / ``logprobs_entropy_of`` — vocab_parallel.py:13-27.
* ``StackEngineOracle`` — the DFS push / block-wise pop schedule of
  tree_training_engine.py:146-616 (forward and backward).
* ``dense_forward`` / ``dense_backward`` — dense.py:7-53.

Weights are a ``{hf_parameter_name: tensor}`` dict so gradients compare name by name with
grad/Qwen3-0.6B-TB-vs-DB-bf16.txt rows.
"""
from __future__ import annotations

import math
from bisect import bisect_left, bisect_right
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .trie_oracle import block_starts, fork_positions

QWEN3_0P6B = dict(vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=28,
                  num_attention_heads=16, num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-6,
                  rope_theta=1000000.0)


def init_weights(cfg: dict, seed: int = 0, dtype=torch.float32, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Random-init weights with the HF parameter names (tied head: no lm_head entry)."""
    g = torch.Generator().manual_seed(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    Hq, Hkv, D = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    w = {"model.embed_tokens.weight": torch.randn(V, H, generator=g) * std}
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        w[p + "self_attn.q_proj.weight"] = torch.randn(Hq * D, H, generator=g) * std
        w[p + "self_attn.k_proj.weight"] = torch.randn(Hkv * D, H, generator=g) * std
        w[p + "self_attn.v_proj.weight"] = torch.randn(Hkv * D, H, generator=g) * std
        w[p + "self_attn.o_proj.weight"] = torch.randn(H, Hq * D, generator=g) * std
        w[p + "self_attn.q_norm.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
        w[p + "self_attn.k_norm.weight"] = 1.0 + 0.1 * torch.randn(D, generator=g)
        w[p + "mlp.gate_proj.weight"] = torch.randn(I, H, generator=g) * std
        w[p + "mlp.up_proj.weight"] = torch.randn(I, H, generator=g) * std
        w[p + "mlp.down_proj.weight"] = torch.randn(H, I, generator=g) * std
        w[p + "input_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        w[p + "post_attention_layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    w["model.norm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    return {k: v.to(dtype) for k, v in w.items()}


def _rms(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def _rope_tables(pos: torch.Tensor, D: int, theta: float, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = pos.float()[:, None] * inv[None, :]
    ang = torch.cat([ang, ang], dim=-1)
    return ang.cos().to(dtype), ang.sin().to(dtype)


def _rot(x, cos, sin):
    h = x.shape[-1] // 2
    return x * cos + torch.cat([-x[..., h:], x[..., :h]], dim=-1) * sin


def qwen3_segment(cfg: dict, w: Dict[str, torch.Tensor], tokens: torch.Tensor, start: int,
                  past_k: Optional[Sequence[torch.Tensor]] = None, past_v: Optional[Sequence[torch.Tensor]] = None):
    """Run ``tokens`` (1-D, B new tokens occupying positions start..start+B-1) through the decoder.

    ``past_k[l]``/``past_v[l]``: ``[Hkv, start, D]`` (post-norm, post-RoPE keys, as a KV cache holds
    them).  Returns ``(logits [B,V], new_k list of [Hkv,B,D], new_v list)``.
    """
    H = cfg["hidden_size"]; Hq = cfg["num_attention_heads"]; Hkv = cfg["num_key_value_heads"]; D = cfg["head_dim"]
    eps = cfg.get("rms_norm_eps", 1e-6); L = cfg["num_hidden_layers"]
    B = tokens.shape[0]
    x = F.embedding(tokens, w["model.embed_tokens.weight"])
    dt = x.dtype
    pos = torch.arange(start, start + B)
    cos, sin = _rope_tables(pos, D, cfg.get("rope_theta", 1e6), dt)
    # offset causal mask: query i sees keys 0..start+i
    kpos = torch.arange(start + B)
    allowed = kpos[None, :] <= pos[:, None]
    bias = torch.zeros(B, start + B, dtype=dt).masked_fill(~allowed, torch.finfo(dt).min)
    scale = D ** -0.5
    new_k, new_v = [], []
    for l in range(L):
        p = f"model.layers.{l}."
        h = _rms(x, w[p + "input_layernorm.weight"], eps)
        q = F.linear(h, w[p + "self_attn.q_proj.weight"]).view(B, Hq, D)
        k = F.linear(h, w[p + "self_attn.k_proj.weight"]).view(B, Hkv, D)
        v = F.linear(h, w[p + "self_attn.v_proj.weight"]).view(B, Hkv, D)
        q = _rot(_rms(q, w[p + "self_attn.q_norm.weight"], eps).transpose(0, 1), cos, sin)   # [Hq,B,D]
        k = _rot(_rms(k, w[p + "self_attn.k_norm.weight"], eps).transpose(0, 1), cos, sin)   # [Hkv,B,D]
        v = v.transpose(0, 1)
        new_k.append(k); new_v.append(v)
        kk = k if start == 0 else torch.cat([past_k[l], k], dim=1)
        vv = v if start == 0 else torch.cat([past_v[l], v], dim=1)
        rep = Hq // Hkv
        kk = kk.repeat_interleave(rep, dim=0); vv = vv.repeat_interleave(rep, dim=0)
        att = torch.matmul(q, kk.transpose(1, 2)) * scale + bias
        att = torch.softmax(att, dim=-1, dtype=torch.float32).to(dt)
        o = torch.matmul(att, vv).transpose(0, 1).reshape(B, Hq * D)
        x = x + F.linear(o, w[p + "self_attn.o_proj.weight"])
        h = _rms(x, w[p + "post_attention_layernorm.weight"], eps)
        m = F.silu(F.linear(h, w[p + "mlp.gate_proj.weight"])) * F.linear(h, w[p + "mlp.up_proj.weight"])
        x = x + F.linear(m, w[p + "mlp.down_proj.weight"])
    x = _rms(x, w["model.norm.weight"], eps)
    return F.linear(x, w["model.embed_tokens.weight"]), new_k, new_v


# ------------------------------------------------------------------------------------------
def logprobs_of(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """[vocab_parallel.py:13-18]  logits [R,V] (any dtype), labels [R'] with R' <= R (torch.gather
    allows the shorter index; the engine relies on it, tte:190-193)."""
    lp = F.log_softmax(logits.float() / temperature, dim=-1)
    return lp.gather(-1, labels.unsqueeze(-1)).squeeze(-1)


def logprobs_entropy_of(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0):
    """[vocab_parallel.py:21-27]  entropy keeps all R rows, logprobs only len(labels)."""
    lp = F.log_softmax(logits.float() / temperature, dim=-1)
    ent = -(lp.exp() * lp).sum(-1)
    return lp.gather(-1, labels.unsqueeze(-1)).squeeze(-1), ent


def default_loss(logprob, entropy, attachment):                       # run.py:149-152
    return attachment["w_logprobs"] * logprob.mean() + attachment["w_entropy"] * entropy.mean()


# ------------------------------------------------------------------------------------------
class StackEngineOracle:
    """The reference's stack engine on CPU tensors.  [tree_training_engine.py:52-616]

    ``w`` leaves must have ``requires_grad`` for ``backward``; their ``.grad`` accumulate across
    pops exactly as ``param.grad`` does in the reference.
    """

    def __init__(self, cfg: dict, w: Dict[str, torch.Tensor], max_seq_len: int, dtype=torch.float32):
        self.cfg, self.w, self.cap, self.dt = cfg, w, max_seq_len, dtype
        L, Hkv, D = cfg["num_hidden_layers"], cfg["num_key_value_heads"], cfg["head_dim"]
        self.top = 0
        self.tok = torch.zeros(max_seq_len, dtype=torch.long)
        self.lp = torch.zeros(max_seq_len); self.ent = torch.zeros(max_seq_len)          # f32 (tte:84,88)
        self.g_lp = torch.zeros(max_seq_len, dtype=dtype); self.g_ent = torch.zeros(max_seq_len, dtype=dtype)
        self.K = torch.zeros(L, Hkv, max_seq_len, D, dtype=dtype); self.V = torch.zeros_like(self.K)
        self.gK = torch.zeros_like(self.K); self.gV = torch.zeros_like(self.K)
        self.fork: Dict[int, torch.Tensor] = {}; self.g_fork: Dict[int, torch.Tensor] = {}
        self.forks: List[int] = []
        self.live: List[Tuple[dict, int]] = []
        self.L = L

    def _forks_in(self, a: int, b: int):                              # tte:135-144
        return self.forks[bisect_left(self.forks, a):bisect_right(self.forks, b - 1)]

    def _past(self, start: int, grad: bool):
        ks = [self.K[l, :, :start].detach().requires_grad_(grad) for l in range(self.L)]
        vs = [self.V[l, :, :start].detach().requires_grad_(grad) for l in range(self.L)]
        return ks, vs

    def _patch_prev_logprob(self, start: int, first_token: int):      # tte:205-209, 307-311
        if start > 0:
            row = F.log_softmax(self.fork[start - 1].float(), dim=-1)
            self.lp[start - 1] = row[first_token].item()

    # ---- inference ----------------------------------------------------------------------
    @torch.no_grad()
    def push_forward_only(self, new_tokens, attach_list, returns):    # tte:146-230
        B = new_tokens.numel()
        assert self.top + B <= self.cap, f"Exceeds max_seq_len: cur_len={self.top}, new_tokens={B}, max={self.cap}"
        a, b = self.top, self.top + B
        ks, vs = self._past(a, False)
        logits, nk, nv = qwen3_segment(self.cfg, self.w, new_tokens, a, ks, vs)
        self.tok[a:b] = new_tokens
        self.lp[a:b - 1] = logprobs_of(logits, new_tokens[1:])
        self._patch_prev_logprob(a, int(new_tokens[0]))
        for l in range(self.L):
            self.K[l, :, a:b] = nk[l]; self.V[l, :, a:b] = nv[l]
        for i in self._forks_in(a, b):
            self.fork[i] = logits[i - a].detach().clone()
        for att, n in attach_list:
            returns[att["_sequence_batch_id"]] = self.lp[:n - 1].clone()
        self.top = b

    @torch.no_grad()
    def forward(self, trie) -> List[torch.Tensor]:                    # tte:515-553
        returns = [None] * trie.n_sequences
        self.forks = fork_positions(None, trie.lcp_lens, None)
        for i, ids in enumerate(trie.inputs):
            ids = torch.as_tensor(ids)
            if i > 0:
                self.top = trie.lcp_lens[i - 1]
            self.push_forward_only(ids[self.top:], trie.attach_lists[i], returns)
        self.top = 0
        self.fork = {}
        return returns

    # ---- training -----------------------------------------------------------------------
    @torch.no_grad()
    def _fill(self, a: int, b: int):                                  # tte:232-272
        ks, vs = self._past(a, False)
        logits, nk, nv = qwen3_segment(self.cfg, self.w, self.tok[a:b], a, ks, vs)
        lp, ent = logprobs_entropy_of(logits, self.tok[a + 1:b])
        self.lp[a:b - 1] = lp; self.ent[a:b] = ent
        for l in range(self.L):
            self.K[l, :, a:b] = nk[l]; self.V[l, :, a:b] = nv[l]
        for i in self._forks_in(a, b):
            self.fork[i] = logits[i - a].detach().clone()

    @torch.no_grad()
    def push(self, new_tokens, attach_list, keep_until: int):         # tte:274-313
        B = new_tokens.numel()
        assert self.top + B <= self.cap, f"Exceeds max_seq_len: cur_len={self.top}, new_tokens={B}, max={self.cap}"
        a, b = self.top, self.top + B
        self.live.extend(attach_list)
        self.tok[a:b] = new_tokens
        if a < keep_until:
            self._fill(a, keep_until)
        self._patch_prev_logprob(a, int(new_tokens[0]))
        self.top = b

    def pop(self, start: int, loss_fn: Callable) -> float:            # tte:315-487
        assert 0 <= start < self.top, f"Invalid start={start}, cur_len={self.top}"
        end = self.top
        toks = self.tok[start:end]
        ks, vs = self._past(start, True)
        logits, nk, nv = qwen3_segment(self.cfg, self.w, toks, start, ks, vs)
        suf_lp, suf_ent = logprobs_entropy_of(logits, toks[1:])
        mid_logits = mid_lp = None
        if start > 0:                                                  # tte:369-372
            mid_logits = self.fork[start - 1].float().detach().requires_grad_(True)
            mid_lp = F.log_softmax(mid_logits, dim=-1)[int(self.tok[start])].unsqueeze(0)
        ending = [(a, n) for a, n in self.live if start < n <= end]    # tte:379
        loss = None; pre_ent = pre_lp = None
        if ending:                                                     # tte:381-398
            if start > 0:
                pre_ent = self.ent[:start].detach().requires_grad_(True)
                ent_all = torch.cat([pre_ent, suf_ent])
                if start > 1:
                    pre_lp = self.lp[:start - 1].detach().requires_grad_(True)
                    lp_all = torch.cat([pre_lp, mid_lp, suf_lp])
                else:
                    lp_all = torch.cat([mid_lp, suf_lp])
            else:
                ent_all, lp_all = suf_ent, suf_lp
            loss = 0.0
            for att, n in ending:
                loss = loss + loss_fn(lp_all[:n - 1], ent_all[:n], att)
        roots, seeds = [], []                                          # tte:404-440
        if ending:
            roots.append(loss); seeds.append(torch.tensor(1.0, dtype=loss.dtype))
        for l in range(self.L):
            roots += [nk[l], nv[l]]
            seeds += [self.gK[l, :, start:end], self.gV[l, :, start:end]]
        roots += [suf_lp, suf_ent]
        seeds += [self.g_lp[start:end - 1], self.g_ent[start:end]]
        if start > 0:
            roots.append(mid_lp); seeds.append(self.g_lp[start - 1].unsqueeze(0))
        for i in self._forks_in(start, end):
            if i in self.g_fork:
                roots.append(logits[i - start]); seeds.append(self.g_fork[i])
        torch.autograd.backward(roots, seeds)
        for l in range(self.L):                                        # tte:447-451
            if ks[l].grad is not None:
                self.gK[l, :, :start] += ks[l].grad
            if vs[l].grad is not None:
                self.gV[l, :, :start] += vs[l].grad
        if start > 0:                                                  # tte:453-465
            if mid_logits.grad is not None:
                if (start - 1) in self.g_fork:
                    self.g_fork[start - 1] += mid_logits.grad
                else:
                    self.g_fork[start - 1] = mid_logits.grad.clone()
            if ending:
                if pre_ent.grad is not None:
                    self.g_ent[:start] += pre_ent.grad
                if start > 1 and pre_lp.grad is not None:
                    self.g_lp[:start - 1] += pre_lp.grad
        self.live = [(a, n) for a, n in self.live if n <= start]       # tte:471-485
        self.gK[:, :, start:end].zero_(); self.gV[:, :, start:end].zero_()
        self.g_lp[(0 if start == 0 else start - 1):end - 1].zero_()
        self.g_ent[start:end].zero_()
        for i in self._forks_in(start, end):
            self.fork.pop(i, None); self.g_fork.pop(i, None)
        self.top = start
        return float(loss.item()) if ending else 0.0

    def pop_blocks(self, start: int, block_size: int, loss_fn) -> float:   # tte:489-513
        total = 0.0
        for s in block_starts(start, self.top, block_size):
            total += self.pop(s, loss_fn)
        return total

    def backward(self, trie, loss_fn, block_size: int, cut_f1_tail: bool = True) -> float:   # tte:555-616
        total = 0.0
        lens = [len(x) for x in trie.inputs]
        self.forks = fork_positions(lens, trie.lcp_lens, block_size)
        M = len(trie.inputs)
        for i in range(M):
            ids = torch.as_tensor(trie.inputs[i])
            if i > 0 and trie.lcp_lens[i - 1] < self.top:
                total += self.pop_blocks(trie.lcp_lens[i - 1], block_size, loss_fn)
            new = ids[self.top:]
            nxt = trie.lcp_lens[i] if i < M - 1 else 0
            B = new.numel()
            span = self.top + B - nxt
            if span > block_size:                                      # tte:601-606
                nb = math.ceil(span / block_size); bs = math.ceil(span / nb)
                keep = max(self.top + B - bs, nxt)
            else:
                keep = nxt
            if not cut_f1_tail:
                keep = self.top + B
            self.push(new, trie.attach_lists[i], keep)
        if self.top > 0:
            total += self.pop_blocks(0, block_size, loss_fn)
        return total


# ------------------------------------------------------------------------------------------
@torch.no_grad()
def dense_forward(cfg, w, seqs) -> List[torch.Tensor]:                 # dense.py:7-24
    out = []
    for s in seqs:
        s = torch.as_tensor(s)
        logits, _, _ = qwen3_segment(cfg, w, s, 0)
        out.append(logprobs_of(logits, s[1:]))
    return out


def dense_backward(cfg, w, seqs, attachs, loss_fn) -> float:           # dense.py:26-53
    total = 0.0
    for s, att in zip(seqs, attachs):
        s = torch.as_tensor(s)
        logits, _, _ = qwen3_segment(cfg, w, s, 0)
        lp, ent = logprobs_entropy_of(logits, s[1:])
        loss = loss_fn(lp, ent, att)
        loss.backward()
        total += float(loss.item())
    return total


def grad_ratio(g_base: torch.Tensor, g_exp: torch.Tensor, eps: float = 1e-6) -> float:
    """|Δg|/|g| as in exp/compare_grads.py:42-57."""
    g0, g1 = g_base.float(), g_exp.float()
    return float(torch.norm(g1 - g0) / (torch.norm(g0) + eps))
