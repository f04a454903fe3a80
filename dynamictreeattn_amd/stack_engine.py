"""Block-wise (memory-bounded) backward over the KV stack — the reference's push / pop schedule
(tree_training_engine.py:232-313 build_cache/push, 315-487 pop, 489-513 pop_byblock, 555-616 backward) on the HIP kernels.

The packed one-pass engine keeps every tree token's activations (or, with per-layer recomputation, two hidden-size rows
per token and layer) alive; on deep-and-wide tries with the larger models that exceeds even 288 GB, where the reference's
`block_size` keeps fitting: its state is the KV stack of the CURRENT root-to-leaf path (tte:108-131) plus the activations
of one block.  This module restores that bound:

* state: K/V stacks `[max_len, Hkv, 128]` per layer (model dtype) and fp32 grad-KV stacks, the path's token ids, its
  log-prob / entropy values and the gradient side channels `g_lp`, `g_ent` (tte:77-131);
* the walk is the reference's: for each leaf in DFS order pop the diverged tail of the previous leaf back to their LCP, then
  push the new leaf's tail — no-grad, and only the part that outlives the next pop (`cut_f1_tail`, tte:596-609);
* a pop re-forwards rows [s, e) WITH autograd in blocks from the end (tte:503-511): `ops.stack_attention` reads the prefix
  K/V in place and adds the block's dK/dV into the grad stacks for all rows [0, e) (no `DynamicCache`, no prefix-sized
  `.grad` tensors, no `+=` over the prefix: tte:339-346, 447-451); losses of the sequences that end inside the block are
  formed over (prefix values as leaves) ‖ (in-graph block values) exactly as tte:379-398;
* the off-by-one — the log-prob of the token at depth d is a function of the row at depth d-1 (tte:205-209, 307-311,
  369-372) — needs no stored fork-position logits (V-sized rows, tte:92-96): when row d leaves the stack its accumulated
  `g_lp[d]` is queued on row d-1 as a PENDING PICK (token, gradient); when row d-1 is popped, the LM-head kernels pick that
  token's log-prob from the row's fresh logits and back-propagate the queued gradient (the fork CSR of
  `dta_logprob_entropy_fwd/bwd`).  Values needed at push time come from the stored final hidden row (hidden-sized).

Results equal the packed engine's (and the reference's) up to summation order; `block_size` is the row budget of a block.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .model import _cfg_of, head_weight, packed_hidden_states
from .tree_training_engine import sum_loss_terms
from .trie import pop_block_starts


class _GradSink:
    """Parameter gradients of a walk, summed over its blocks in PERSISTENT fp32 buffers.

    The reference's `autograd.backward` per popped block runs `AccumulateGrad` over every parameter in model dtype (tte:440): the
    sum over the 64 blocks of a config-5 call is rounded to bf16 / f16 after every add, and the embedding's gradient is materialised
    as a dense, mostly-zero [vocab, hidden] matrix per block.  Here, for the duration of the walk, every `param.grad` is a view of ONE
    flat model-dtype buffer that is zero when a block's backward starts; `absorb()` adds that buffer into ONE flat fp32 buffer and
    clears it - two launches per block whatever the number of parameters (a per-parameter mixed-dtype add cost 5 ms per block in host
    and launch time: 310 tensors) - the embedding rows go straight into the tied weight's fp32 slice by `index_add` on the rows the block
    touched (`embed()`), and `finish()` rounds ONCE into `param.grad` (added to whatever gradient the caller had accumulated before
    the call, as `backward()` must; a `dp.GradReducer` view stays the same tensor).  Inactive - the blocks then accumulate in model
    dtype as the reference does - when the parameters are fp32 already, are not of one dtype on one device, or the buffers
    (6 bytes per parameter) do not fit `budget_bytes`."""

    def __init__(self, model, budget_bytes: Optional[int]):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.embed_weight = model.model.embed_tokens.weight
        p0 = self.params[0] if self.params else None
        n = sum(p.numel() for p in self.params)
        self.active = (p0 is not None and p0.dtype in (torch.bfloat16, torch.float16)
                       and all(p.dtype == p0.dtype and p.device == p0.device for p in self.params)
                       and (budget_bytes is None or n * (4 + p0.element_size()) <= budget_bytes))
        if not self.active:
            return
        self.orig = [p.grad for p in self.params]
        self.flat_lo = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.flat32 = torch.zeros(n, dtype=torch.float32, device=p0.device)
        self._span, o = {}, 0
        for p in self.params:
            self._span[id(p)] = (o, o + p.numel())
            p.grad = self.flat_lo[o:o + p.numel()].view_as(p)
            o += p.numel()

    def _f32(self, p) -> torch.Tensor:
        a, b = self._span[id(p)]
        return self.flat32[a:b].view_as(p)

    def embed(self, tokens: torch.Tensor) -> torch.Tensor:
        w = self.embed_weight
        if not (self.active and torch.is_grad_enabled() and w.requires_grad):
            return F.embedding(tokens, w)
        return _EmbedRows.apply(tokens, w, self._f32(w))

    def absorb(self) -> None:
        if self.active:
            self.flat32.add_(self.flat_lo)
            self.flat_lo.zero_()

    def abort(self) -> None:
        """The walk did not complete (the user's loss_fn raised, out of memory ...): hand the caller's own gradients back untouched."""
        if not self.active:
            return
        for p, o in zip(self.params, self.orig):
            p.grad = o
        self.flat_lo = self.flat32 = self.orig = None
        self.active = False

    def finish(self) -> None:
        if not self.active:
            return
        for p, o in zip(self.params, self.orig):
            g = self._f32(p)
            if o is None:
                p.grad = g.to(p.dtype)
            else:
                o.add_(g)                  # fp32 sum rounded once into the caller's gradient
                p.grad = o
        self.flat_lo = self.flat32 = self.orig = None
        self.active = False


class _EmbedRows(torch.autograd.Function):
    """Embedding lookup whose backward adds the rows' gradients into the sink's fp32 buffer of the (tied) embedding weight -
    B rows touched instead of a dense [vocab, hidden] gradient per block (§6 profile of SURVEY: embedding-grad zero-fill per pop)."""

    @staticmethod
    def forward(ctx, tokens, weight, sink_buf):
        ctx.save_for_backward(tokens)
        ctx.sink = sink_buf
        return F.embedding(tokens, weight)

    @staticmethod
    def backward(ctx, g):
        (tokens,) = ctx.saved_tensors
        ctx.sink.index_add_(0, tokens, g.float())
        return None, None, None


class StackWalk:
    def __init__(self, model, token_trie, device, dtype, block_rows: int, head_chunk: int = 2048, tp_group=None, grad_budget_bytes: Optional[int] = None):
        self.model, self.trie, self.dev, self.dtype = model, token_trie, device, dtype
        self.block = max(int(block_rows), 1)
        self.head_chunk, self.tp_group = head_chunk, tp_group
        Hq, Hkv, D, _, _ = _cfg_of(model)
        L = len(model.model.layers)
        hidden = model.model.embed_tokens.weight.shape[1]
        cap = max(token_trie.lens) if token_trie.lens else 0
        self.cap, self.L = cap, L
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=device)
        self.kst = [z(cap, Hkv, D) for _ in range(L)]
        self.vst = [z(cap, Hkv, D) for _ in range(L)]
        self.gk = [z(cap, Hkv, D, dt=torch.float32) for _ in range(L)]
        self.gv = [z(cap, Hkv, D, dt=torch.float32) for _ in range(L)]
        self.tokens = torch.zeros(cap + 1, dtype=torch.long, device=device)      # +1: the label of the last row is read at cap
        self.lp = z(cap + 1, dt=torch.float32)        # lp[d]  = log p(token at depth d | row d-1), d >= 1
        self.ent = z(cap, dt=torch.float32)           # ent[d] = entropy of the distribution predicted by row d
        self.hfin = z(cap, hidden)                    # final-norm hidden row of depth d (value only)
        self.g_lp = z(cap + 1, dt=torch.float32)
        self.g_ent = z(cap, dt=torch.float32)
        self.pending: Dict[int, List[Tuple[torch.Tensor, torch.Tensor]]] = {}     # row -> [(token [1], gradient [1])]
        self.attachs: List[Tuple[dict, int]] = []
        self.cur = 0
        self.n_blocks = 0
        self.sink = _GradSink(model, grad_budget_bytes)

    # ------------------------------------------------------------------------------------------
    def _attn_of_layer(self, start: int, with_grad: bool):
        def of(l):
            return lambda q, k, v: ops.stack_attention(q, k, v, self.kst[l], self.vst[l], self.gk[l] if with_grad else None,
                                                       self.gv[l] if with_grad else None, start)
        return of

    def _hidden(self, s: int, e: int, with_grad: bool):
        pos = torch.arange(s, e, device=self.dev, dtype=torch.int32)
        return packed_hidden_states(self.model, self.tokens[s:e].clone(), pos, None, False, 0, self._attn_of_layer(s, with_grad),
                                    embed=self.sink.embed)

    def _head(self, h, s: int, e: int, want_forks: bool, last_label: bool = True):
        """(lp_next [B], lp_fork [F], ent [B], g_fork [F]) of rows s..e-1: lp_next[r-s] = log p(tokens[r+1] | row r) with the label of
        the last row masked when row e is not its chain child any more; lp_fork picks the pending tokens of the rows."""
        B = e - s
        W = head_weight(self.model)
        labels = self.tokens[s + 1:e + 1].clone()
        if not last_label:
            labels[-1] = -1                    # no chain child: -1 = no label (log-prob 0, no gradient)
        rows, toks, grads = [], [], []
        if want_forks:
            for r in sorted(k for k in self.pending if s <= k < e):        # the rows that HAVE picks (forks / block seams), not every row
                for tok, g in self.pending[r]:
                    rows.append(r - s); toks.append(tok); grads.append(g)
        F_ = len(rows)
        if F_:
            rows_np = np.asarray(rows, np.int64)                            # ascending by construction
            fork_tok = torch.cat(toks); g_fork = torch.cat(grads)
            ptr = np.zeros(B + 1, np.int32); np.cumsum(np.bincount(rows_np, minlength=B), out=ptr[1:])
            from ._staging import upload                                    # asynchronous, out of page-locked staging: no host sync per block
            (fork_ptr,) = upload([ptr], self.dev, np.int32)
            (fork_rows,) = upload([rows_np], self.dev, np.int64)
            bounds = np.searchsorted(rows_np, np.arange(0, B + self.head_chunk, self.head_chunk)).tolist()
        else:
            fork_tok = self.tokens.new_zeros(0); g_fork = None; fork_ptr = None; fork_rows = self.tokens.new_zeros(0)
            bounds = [0] * (B // self.head_chunk + 2)
        kw = {}
        if self.tp_group is not None:
            import torch.distributed as dist
            tp, rk = dist.get_world_size(self.tp_group), dist.get_rank(self.tp_group)
            Vp = W.shape[0] // tp
            W = W[rk * Vp:(rk + 1) * Vp]
            kw = dict(tp_group=self.tp_group, vocab_offset=rk * Vp)
        lp_next, lp_fork, ent = ops.lm_head_rows(h, W, labels, fork_ptr, fork_tok, fork_rows, bounds, True, self.head_chunk, **kw)
        return lp_next, lp_fork, ent, g_fork

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _push_values(self, s: int, e: int, leaf_len: int):
        """No-grad forward of rows [s, e): fills the K/V stacks and the value buffers (build_cache, tte:232-272)."""
        for a in range(s, e, self.block):
            b = min(a + self.block, e)
            h = self._hidden(a, b, False)
            self.hfin[a:b] = h
            lp_next, _, ent = self._head(h, a, b, False)[:3]
            self.ent[a:b] = ent
            n = min(b, leaf_len - 1) - a                   # rows whose chain child exists on this leaf
            if n > 0:
                self.lp[a + 1:a + 1 + n] = lp_next[:n]

    @torch.no_grad()
    def _link_value(self, d: int):
        """lp[d] for the first token of a pushed tail: picked from the stored hidden row d-1 (tte:205-209, 307-311)."""
        self.lp[d:d + 1] = self._head(self.hfin[d - 1:d], d - 1, d, False)[0]

    def push(self, leaf_tokens: torch.Tensor, attach_list, next_lcp: int):
        """Push the tail of a leaf (rows cur .. len-1).  Only rows that outlive the next pop are forwarded now (cut_f1_tail,
        tte:596-609): with a block-wise pop that is everything below the LAST block of the pop (trie.py:26-45 `n_f1_tokens`)."""
        start, end = self.cur, int(leaf_tokens.numel())
        assert end <= self.cap, f"Exceeds max_seq_len: cur_len={start}, new_tokens={end - start}, max={self.cap}"      # tte:289-291
        self.tokens[start:end] = leaf_tokens[start:end]
        self.attachs.extend(attach_list)
        pop_to = min(next_lcp, end)
        keep_to = pop_to if end - pop_to <= self.block else pop_block_starts(pop_to, end, self.block)[0]
        keep_to = max(start, min(keep_to, end))
        if start > 0:
            self._link_value(start)
        if keep_to > start:
            self._push_values(start, keep_to, end)
        self.cur = end

    # ------------------------------------------------------------------------------------------
    def _pop_block(self, s: int, e: int, loss_fn: Callable):
        """Backward of rows [s, e) (pop, tte:315-487).  Returns the loss of the sequences that end inside, as a 0-d tensor or None."""
        B = e - s
        self.n_blocks += 1
        # the chain child of the last row left the stack before this block (or never existed): its label is masked; its
        # gradient, if any, arrives as a pending pick
        h = self._hidden(s, e, True)
        lp_next, lp_fork, ent, g_fork = self._head(h, s, e, True, last_label=False)
        lp_in = lp_next[:B - 1]                                         # depths s+1 .. e-1
        here = [(att, n) for att, n in self.attachs if s < n <= e]
        roots, grads = [lp_in, ent], [self.g_lp[s + 1:e].clone(), self.g_ent[s:e].clone()]
        if g_fork is not None:
            roots.append(lp_fork); grads.append(g_fork)
        loss = None
        pre_lp = pre_ent = None
        if here:
            pre_lp = self.lp[1:s + 1].detach().clone().requires_grad_(True)       # depths 1 .. s (tte:383-393)
            pre_ent = self.ent[:s].detach().clone().requires_grad_(True)
            lp_vec, ent_vec = torch.cat([pre_lp, lp_in]), torch.cat([pre_ent, ent])
            terms = [loss_fn(lp_vec[:n - 1], ent_vec[:n], att) for att, n in here]
            loss = sum_loss_terms(terms, self.dev)
            if loss.requires_grad:                 # a callback may return constants for the sequences of a block
                roots.append(loss); grads.append(None)
        torch.autograd.backward(roots, grads)
        self.sink.absorb()
        if pre_lp is not None and s > 0:
            if pre_lp.grad is not None:
                self.g_lp[1:s + 1] += pre_lp.grad
            if pre_ent.grad is not None:
                self.g_ent[:s] += pre_ent.grad
        # hand the link of row s to row s-1, then clear what belonged to the popped rows (tte:471-485)
        for r in [k for k in self.pending if s <= k < e]:
            del self.pending[r]
        if s > 0:
            self.pending.setdefault(s - 1, []).append((self.tokens[s:s + 1].clone(), self.g_lp[s:s + 1].clone()))
        self.g_lp[s:e].zero_(); self.g_ent[s:e].zero_()
        for l in range(self.L):
            self.gk[l][s:e].zero_(); self.gv[l][s:e].zero_()
        self.attachs = [(att, n) for att, n in self.attachs if n <= s]
        return loss.detach() if loss is not None else None

    def pop_to(self, start: int, loss_fn: Callable):
        """pop_byblock (tte:489-513): rows [start, cur) leave the stack in equal blocks from the end."""
        total = None
        e = self.cur
        for s in pop_block_starts(start, self.cur, self.block):       # last block first
            l = self._pop_block(s, e, loss_fn)
            if l is not None:
                total = l if total is None else total + l
            e = s
        self.cur = start
        return total

    # ------------------------------------------------------------------------------------------
    def run(self, loss_fn: Callable):
        try:
            return self._run(loss_fn)
        except BaseException:
            self.sink.abort()
            raise

    def _run(self, loss_fn: Callable):
        trie = self.trie
        M = len(trie.lens)
        dev_tokens, starts = trie._dev.tokens, trie._dev.starts
        total = None
        for i in range(M):
            lcp = trie.lcp_lens[i - 1] if i > 0 else 0
            if lcp < self.cur:
                l = self.pop_to(lcp, loss_fn)
                if l is not None:
                    total = l if total is None else total + l
            src = trie._leaf_src[i]
            leaf = dev_tokens[int(starts[src]):int(starts[src]) + trie.lens[i]]
            nxt = trie.lcp_lens[i] if i < M - 1 else 0
            self.push(leaf, trie.attach_lists[i], nxt)
        if self.cur > 0:
            l = self.pop_to(0, loss_fn)
            if l is not None:
                total = l if total is None else total + l
        self.sink.finish()
        return total
