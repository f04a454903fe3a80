"""CPU stand-ins for the product's three device steps, for `-m "not gpu"` tests of the HOST logic
only (sorting, leaf grouping, permutation, packing plan, loss assembly, balancers).  They are
installed by monkeypatching from tests; the product never imports this module or the oracle."""
import numpy as np
import torch

from oracle import attn_oracle, trie_oracle


def expand_plan_host(plan):
    """numpy mirror of dta_preorder_meta's index arithmetic -> (depth, parent, subtree_end)."""
    T, M = plan.T, plan.M
    seg = np.repeat(np.arange(M, dtype=np.int32), np.diff(plan.seg_off))
    j = np.arange(T, dtype=np.int32) - plan.seg_off[seg]
    depth = (plan.seg_depth0[seg] + j).astype(np.int32)
    parent = np.where(j > 0, np.arange(T, dtype=np.int32) - 1, plan.parent_of_seg[seg]).astype(np.int32)
    se = np.empty(T, np.int32)
    for i in range(M):
        a, b = plan.brk_ptr[i], plan.brk_ptr[i + 1]
        s, e = plan.seg_off[i], plan.seg_off[i + 1]
        if e > s:
            k = np.searchsorted(plan.brk_depth[a:b], depth[s:e], side="right") - 1
            se[s:e] = plan.brk_end[a:b][k]
    return seg, depth, parent, se


def _cpu_trie_arrays(dev, order, leafize):
    toks = dev.tokens.numpy()
    seqs = [toks[dev.starts[i]:dev.starts[i] + dev.lens[i]] for i in order]
    lcp, unsorted = [], 0
    for a, b in zip(seqs[:-1], seqs[1:]):
        c = trie_oracle.lcp_pair(a, b)
        unsorted += int(c < min(len(a), len(b)) and a[c] > b[c])
        lcp.append(c)
    if not leafize:
        return lcp, unsorted, None
    S = len(seqs)
    keep = [i for i in range(S) if i == S - 1 or lcp[i] < min(len(seqs[i]), len(seqs[i + 1]))]
    return [lcp[i] for i in keep[:-1]], unsorted, keep


def _cpu_expand(self, tokens, leaf_off, seg_off, seg_d0, par_seg, brk_ptr, brk_depth, brk_end, M, T):
    seg, depth, parent, se = expand_plan_host(self.plan)
    self.depth.copy_(torch.from_numpy(depth)); self.parent.copy_(torch.from_numpy(parent)); self.subtree_end.copy_(torch.from_numpy(se))
    src = leaf_off.numpy()[seg] + depth
    self.tokens.copy_(tokens[torch.from_numpy(src)])


def _cpu_attention(q, k, v, meta, scale=None):
    T = q.shape[0]
    if meta.subtree_end is None:
        se = torch.full((k.shape[0],), meta.q_offset + T, dtype=torch.long)
        assert meta.q_offset == 0
    else:
        se = meta.subtree_end.long()
    return attn_oracle.tree_attention(q, k, v, se, scale)[0]


def _cpu_lm_head_rows(h, W, next_tok, fork_ptr, fork_tok, fork_rows, fork_bounds, want_entropy, chunk, keep_bytes=None):
    """Plain-torch stand-in of ops.lm_head_rows (fp32 log-softmax over the full rows); a label < 0 = no label (0)."""
    lp_all = torch.log_softmax(torch.nn.functional.linear(h, W).float(), dim=-1)
    ent = -(lp_all.exp() * lp_all).sum(-1) if want_entropy else None
    lp_next = lp_all.gather(-1, next_tok.clamp(min=0)[:, None]).squeeze(-1) * (next_tok >= 0)
    lp_fork = lp_all[fork_rows, fork_tok] if fork_rows.numel() else lp_all.new_zeros(0)
    return lp_next, lp_fork, ent


class _CpuStackAttention(torch.autograd.Function):
    """Stand-in of ops._StackAttention with the same side effects: K/V written into the stack rows, dK/dV ADDED into the fp32
    grad stacks for all rows [0, end), and the grad-stack rows of the block returned as the gradient of its own K/V."""

    @staticmethod
    def forward(ctx, q, k_new, v_new, kst, vst, gk, gv, start, scale):
        end = start + q.shape[0]
        kst[start:end].copy_(k_new); vst[start:end].copy_(v_new)
        ctx.save_for_backward(q)
        ctx.stacks, ctx.span, ctx.scale = (kst, vst, gk, gv), (start, end), scale
        return attn_oracle.rect_causal_attention(q.transpose(0, 1), kst[:end].transpose(0, 1), vst[:end].transpose(0, 1), start, scale)

    @staticmethod
    def backward(ctx, dout):
        (q,) = ctx.saved_tensors
        kst, vst, gk, gv = ctx.stacks
        start, end = ctx.span
        with torch.enable_grad():
            q_, k_, v_ = (t.detach().clone().requires_grad_(True) for t in (q, kst[:end], vst[:end]))
            out = attn_oracle.rect_causal_attention(q_.transpose(0, 1), k_.transpose(0, 1), v_.transpose(0, 1), start, ctx.scale)
            dq, dk, dv = torch.autograd.grad(out, (q_, k_, v_), dout)
        gk[:end] += dk.float(); gv[:end] += dv.float()
        return dq, gk[start:end].to(q.dtype), gv[start:end].to(q.dtype), None, None, None, None, None, None


def _cpu_stack_attention(q, k_new, v_new, kst, vst, gk, gv, start, scale=None):
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    return _CpuStackAttention.apply(q, k_new, v_new, kst, vst, gk, gv, start, scale)


def _cpu_rms_norm(x, w, eps):
    xf = x.float()
    return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)


def _cpu_add_rms_norm(x, delta, w, eps):
    if delta is not None:
        x = x + delta
    return x, _cpu_rms_norm(x, w, eps)


def _cpu_qk_norm_rope(x, w, cos_sin, eps):
    if w is not None:
        x = _cpu_rms_norm(x, w, eps)
    half = x.shape[-1] // 2
    cos = torch.cat([cos_sin[:, :half], cos_sin[:, :half]], -1).to(x.dtype)[:, None, :]
    sin = torch.cat([cos_sin[:, half:], cos_sin[:, half:]], -1).to(x.dtype)[:, None, :]
    return x * cos + torch.cat([-x[..., half:], x[..., :half]], -1) * sin


def _cpu_qkv_prep(qkv, wq, wk, cos_sin, eps, Hq, Hkv):
    q, k, v = qkv.split([Hq, Hkv, Hkv], dim=1)
    return _cpu_qk_norm_rope(q, wq, cos_sin, eps), _cpu_qk_norm_rope(k, wk, cos_sin, eps), v


def _cpu_swiglu(g, u):
    return torch.nn.functional.silu(g) * u


def _cpu_swiglu_fused(gu):
    C = gu.shape[-1] // 2
    return torch.nn.functional.silu(gu[..., :C]) * gu[..., C:]


# ------------------------------------------------------------------------------------------------------------------
# vocab_parallel on host tensors (the product raises there): a torch restatement of vocab_parallel.py:13-27 (local) and
# :82-370 (vocabulary-sharded, with the product's packed collectives: ONE MAX + ONE SUM), used by the gloo tests.
# ------------------------------------------------------------------------------------------------------------------
class _CpuShardedLogProbEntropy(torch.autograd.Function):
    """Shard = contiguous slice rank·V/tp (vocab_parallel.py:128-130); labels are shard-local, -1 = not mine / no label."""

    @staticmethod
    def forward(ctx, x2, lab_local, group, want_entropy):
        import torch.distributed as dist
        mx = x2.max(dim=-1).values
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        e = torch.exp(x2 - mx[:, None])
        mine = lab_local >= 0
        idx = lab_local.clamp(min=0)
        picked = torch.where(mine, x2.gather(-1, idx[:, None]).squeeze(-1) - mx, torch.zeros_like(mx))
        packed = torch.stack([e.sum(-1), picked, (e * x2).sum(-1)], dim=-1)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        z, picked, ex = packed.unbind(-1)
        logz = mx + torch.log(z)
        mean_x = ex / z
        ctx.save_for_backward(x2, logz, mean_x, mine, idx)
        ctx.want_entropy = want_entropy
        return picked - torch.log(z), (logz - mean_x)

    @staticmethod
    def backward(ctx, g_lp, g_ent):
        x2, logz, mean_x, mine, idx = ctx.saved_tensors
        p = torch.exp(x2 - logz[:, None])
        g = -p * g_lp.reshape(-1, 1)
        if ctx.want_entropy and g_ent is not None:
            g = g + g_ent.reshape(-1, 1) * p * (mean_x[:, None] - x2)
        g.scatter_add_(-1, idx[:, None], (g_lp.reshape(-1) * mine)[:, None].to(g.dtype))
        return g, None, None, None


def _cpu_logprob_entropy(logits2d, labels1d, temperature, want_entropy, tp_group):
    """Stand-in of vocab_parallel._device_logprob_entropy: labels int64 [R], -1 = no label (log-prob 0)."""
    x = logits2d.float() / temperature if temperature != 1.0 else logits2d.float()
    if tp_group is None:
        lp_all = torch.log_softmax(x, dim=-1)
        lp = lp_all.gather(-1, labels1d.clamp(min=0)[:, None]).squeeze(-1) * (labels1d >= 0)
        return lp, (-(lp_all.exp() * lp_all).sum(-1) if want_entropy else None)
    import torch.distributed as dist
    V = x.shape[1]
    lo = dist.get_rank(tp_group) * V
    lab = torch.where((labels1d >= lo) & (labels1d < lo + V), labels1d - lo, torch.full_like(labels1d, -1))
    lp, ent = _CpuShardedLogProbEntropy.apply(x, lab, tp_group, want_entropy)
    return lp, (ent if want_entropy else None)


def install_cpu_logprob(monkeypatch=None):
    """Patch the host restatement over the product's device entry (spawned gloo workers call this without a monkeypatch)."""
    from dynamictreeattn_amd import vocab_parallel
    if monkeypatch is None:
        vocab_parallel._device_logprob_entropy = _cpu_logprob_entropy
    else:
        monkeypatch.setattr(vocab_parallel, "_device_logprob_entropy", _cpu_logprob_entropy)


def install(monkeypatch):
    from dynamictreeattn_amd import ops, token_trie, tree_training_engine
    monkeypatch.setattr(token_trie, "_device_trie_arrays", _cpu_trie_arrays)
    monkeypatch.setattr(tree_training_engine._PackedTrie, "_expand", _cpu_expand)
    monkeypatch.setattr(ops, "tree_attention", _cpu_attention)
    monkeypatch.setattr(ops, "stack_attention", _cpu_stack_attention)
    monkeypatch.setattr(ops, "lm_head_rows", _cpu_lm_head_rows)
    monkeypatch.setattr(ops, "rms_norm", _cpu_rms_norm)
    monkeypatch.setattr(ops, "add_rms_norm", _cpu_add_rms_norm)
    monkeypatch.setattr(ops, "qk_norm_rope", _cpu_qk_norm_rope)
    monkeypatch.setattr(ops, "qkv_prep", _cpu_qkv_prep)
    monkeypatch.setattr(ops, "swiglu", _cpu_swiglu)
    monkeypatch.setattr(ops, "swiglu_fused", _cpu_swiglu_fused)
