"""TreeTrainingEngine — drop-in for the reference's engine (tree_training_engine.py:52-616), new
mechanism.

The reference walks the trie leaf by leaf over a stack-shaped KV cache: one model call per pushed
segment, and per popped block a re-forward + ``autograd.backward`` with K/V-, logprob-, entropy- and
fork-logit-gradients injected by hand (tte:315-487).  On a 288 GB MI355X the whole trie fits in one
pass, so this engine

1. lays all tree tokens out in DFS pre-order of the (already permuted) leaves — HIP kernel
   ``dta_preorder_meta`` gathers token ids and writes depth / parent / subtree_end per token from
   O(M·height) host tables (packing.py);
2. runs the decoder ONCE over the T packed tokens with position = depth and the HIP tree-attention
   operator (ancestor visibility  s <= t < subtree_end[s]);
3. takes logprob(token | parent) and entropy(node) for every tree node, gathers them along each
   leaf's root path, applies the user's ``loss_fn`` per original sequence exactly as tte:379-398 does
   (``logprobs[:len-1]``, ``entropy[:len]``, attachment) and calls ``backward()`` once.

Same results (sum-of-sequence losses, parameter gradients), none of the side channels
(``grad_kv``, ``grad_logprobs``, ``grad_entropy``, fork logits), no per-segment host syncs.
``block_size`` keeps its meaning as the memory knob: it is the row-chunk of the LM-head/logprob
stage (the [rows, V] logits are never materialised beyond one chunk); ``cut_f1_tail`` has nothing
left to cut (no token is forwarded twice) and is accepted for signature parity.
"""
from __future__ import annotations

import os

from typing import Callable, List, Optional

import numpy as np
import torch

from . import ops, packing
from .model import head_weight, packed_hidden_states
from .trie import pop_block_starts

__all__ = ["TreeTrainingEngine", "_get_forkpos", "packed_logprob_entropy"]


def _get_forkpos(lens, lcp_lens, block_size: Optional[int]) -> list:
    """Sorted unique stack positions whose logits the reference engine must keep: branch points
    (lcp-1) and block boundaries (block start-1).  tree_training_engine.py:12-49.  The packed engine
    does not need them; kept as part of the public surface (stats / tooling)."""
    pos = {c - 1 for c in lcp_lens if c > 0}
    if block_size is not None:
        for i, end in enumerate(lens):
            start = lcp_lens[i] if i < len(lcp_lens) else 0
            pos.update(s - 1 for s in pop_block_starts(start, end, block_size) if s > 0)
    return sorted(pos)


# --------------------------------------------------------------------------------------------------
# LM head + logprob/entropy over packed rows
# --------------------------------------------------------------------------------------------------
def fork_tables_host(fork_child, fork_parent, T: int):
    """Forks (tokens whose parent is not the preceding packed token) sorted by parent row + their CSR over the T rows."""
    fork_child, fork_parent = np.asarray(fork_child, np.int64), np.asarray(fork_parent, np.int64)
    order = np.argsort(fork_parent, kind="stable")
    fc, fp = fork_child[order], fork_parent[order]
    ptr = np.zeros(T + 1, np.int64)
    if fp.size:
        np.cumsum(np.bincount(fp, minlength=T), out=ptr[1:])
    return fc, fp, ptr


def packed_logprob_entropy(h: torch.Tensor, W: torch.Tensor, tokens: torch.Tensor, parent: torch.Tensor,
                           want_entropy: bool, chunk: int = 2048, fork_child: Optional[np.ndarray] = None,
                           fork_parent: Optional[np.ndarray] = None, tp_group=None, fork_dev=None):
    """lp[t] = log softmax(h[parent[t]] Wᵀ)[tokens[t]] (0 for roots), ent[t] = H(softmax(h[t] Wᵀ)); fp32.
    The arithmetic (vocab_parallel.py:13-27; call sites tte:190-193, 256-261, 361-372) runs in
    `ops.lm_head_rows`: hipBLASLt logits GEMM per row chunk + the HIP statistics kernels; at most one
    [chunk, V] block of fp32-free, model-dtype logits per chunk is ever alive.
    `fork_child`/`fork_parent`: host lists of the tokens whose parent is not the preceding packed token (sorted by parent);
    `fork_dev` = their device copies (child int64, parent int64, CSR ptr int32 [T+1]) when the caller uploaded them with the
    plan tables — otherwise they are uploaded here (a blocking copy in the middle of the step)."""
    T = h.shape[0]
    dev = h.device
    if fork_child is None:
        fork_child = np.zeros(0, np.int64); fork_parent = np.zeros(0, np.int64)
    if fork_dev is None:
        fork_child, fork_parent, ptr = fork_tables_host(fork_child, fork_parent, T)
        fc_dev, fp_dev = torch.from_numpy(fork_child).to(dev), torch.from_numpy(fork_parent).to(dev)
        fork_ptr = torch.from_numpy(ptr.astype(np.int32)).to(dev) if fork_child.size else None
    else:
        fc_dev, fp_dev, fork_ptr = fork_dev
    nF = int(np.asarray(fork_child).size)
    ftok = tokens[fc_dev] if nF else tokens.new_zeros(0)
    if not nF:
        fork_ptr = None
    nxt = torch.cat([tokens[1:], tokens.new_zeros(1)])
    bounds = np.searchsorted(np.asarray(fork_parent), np.arange(0, T + chunk, chunk)).tolist()
    if tp_group is not None:
        # vocabulary split across the group (BASELINE config 4; vocab_parallel.py:128-130): this rank multiplies by
        # its contiguous slice of the (tied) head weight only; labels stay global
        import torch.distributed as dist
        tp, rk = dist.get_world_size(tp_group), dist.get_rank(tp_group)
        Vp = W.shape[0] // tp
        assert Vp * tp == W.shape[0], "vocabulary must divide by the tensor-parallel size"
        lp_next, lp_fork, ent = ops.lm_head_rows(h, W[rk * Vp:(rk + 1) * Vp], nxt, fork_ptr, ftok, fp_dev, bounds, want_entropy, chunk,
                                                 tp_group=tp_group, vocab_offset=rk * Vp)
    else:
        lp_next, lp_fork, ent = ops.lm_head_rows(h, W, nxt, fork_ptr, ftok, fp_dev, bounds, want_entropy, chunk)
    chain = torch.zeros(T, dtype=torch.bool, device=dev)           # lp_next[r] = log p(tokens[r+1] | node r)
    chain[1:] = parent[1:] == torch.arange(0, T - 1, device=dev, dtype=parent.dtype)
    lp = torch.cat([lp_next.new_zeros(1), lp_next[:-1]]) * chain
    if nF:
        lp = lp.index_copy(0, fc_dev, lp_fork)
    return lp, ent


def sum_loss_terms(terms, device):
    """Sum of the per-sequence losses (tte:396-398 `loss += loss_fn(...)`).  `loss_fn` returns a 0-d tensor; a plain Python
    number (a constant term for some sequence, which the reference's `0.0 + ...` tolerates) is accepted too.  None if empty."""
    if not terms:
        return None
    ts = [t if isinstance(t, torch.Tensor) else torch.as_tensor(float(t), dtype=torch.float32, device=device) for t in terms]
    return ts[0].reshape(()) if len(ts) == 1 else torch.stack([t.reshape(()).float() for t in ts]).sum()


# --------------------------------------------------------------------------------------------------
class _PackedTrie:
    """Device-resident packed form of a TokenTrie (in its current leaf order)."""

    # packed lengths from PAD_FROM on are rounded up to PAD_TO rows of filler (packing.pad_plan); the env var is a diagnostic A/B switch
    PAD_FROM = int(os.environ.get("DTA_PAD_FROM", 2048))
    PAD_TO = 256

    def __init__(self, trie, device, n_kv_heads: int = 8):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if trie._dev.device != device:
            raise RuntimeError(f"TokenTrie holds its tokens on {trie._dev.device} but the engine runs on {device}: build it with "
                               f"TokenTrie(..., device={str(device)!r}) (or torch.cuda.set_device) so that one GPU owns the whole pass")
        plan = packing.plan_segments(trie.lens, trie.lcp_lens)
        self.n_real_tokens = plan.T
        M_real = plan.M
        leaf_off = [trie._dev.starts[s] for s in trie._leaf_src]
        if plan.T >= self.PAD_FROM and plan.T % self.PAD_TO:
            plan = packing.pad_plan(plan, self.PAD_TO)       # filler ids are read from the head of the token buffer
            leaf_off = leaf_off + [0]
        self.plan = plan
        M, T = plan.M, plan.T
        run_ptr, runs = packing.plan_qtile_runs(plan)
        leaf_off = np.asarray(leaf_off, np.int64)
        # key-tile query ends and the balanced dK/dV work units come from the host plan: no device->host read
        kq = packing.ktile_qend_host(plan)
        units, splits, n_slabs = packing.plan_dkv_units(kq, T, T, 0, n_kv_heads)
        first = plan.seg_off[:-1].astype(np.int64)
        nonempty = np.diff(plan.seg_off) > 0
        par = plan.parent_of_seg.astype(np.int64)
        is_fork = nonempty & (par >= 0) & (par != first - 1)
        self.fork_child, self.fork_parent, fork_ptr = fork_tables_host(first[is_fork], par[is_fork], T)
        # root path of every leaf as packed indices (depth 0 .. len-1)
        paths = []
        for i in range(M_real):
            pieces = [np.arange(b, e, dtype=np.int64) for b, e in plan.path_runs[i]]
            pieces.append(np.arange(plan.seg_off[i], plan.seg_off[i + 1], dtype=np.int64))
            paths.append(np.concatenate(pieces))
        self.path_sizes = [p.size for p in paths]
        self._paths_host = paths
        # ALL tables of the step in two asynchronous uploads out of page-locked staging (one int32, one int64 buffer)
        from ._staging import upload
        (seg_off, seg_d0, par_seg, brk_ptr, brk_depth, brk_end, run_ptr_d, runs_d, kq_d, units_d, splits_d, fptr_d) = upload(
            [plan.seg_off, plan.seg_depth0, plan.parent_of_seg, plan.brk_ptr, plan.brk_depth, plan.brk_end, run_ptr, runs.reshape(-1),
             kq, units.reshape(-1), splits.reshape(-1), fork_ptr], device, np.int32)
        leaf_off_d, self.path_cat, fc_d, fp_d = upload(
            [leaf_off, np.concatenate(paths) if paths else np.zeros(0, np.int64), self.fork_child, self.fork_parent], device, np.int64)
        self.fork_dev = (fc_d, fp_d, fptr_d)
        self.paths = list(torch.split(self.path_cat, self.path_sizes))
        self.tokens = torch.empty(T, dtype=torch.long, device=device)
        meta_i = torch.empty(3, T, dtype=torch.int32, device=device)
        self.depth, self.parent, self.subtree_end = meta_i[0], meta_i[1], meta_i[2]
        self._expand(trie._dev.tokens, leaf_off_d, seg_off, seg_d0, par_seg, brk_ptr, brk_depth, brk_end, M, T)
        self.meta = ops.TreeAttnMeta(T=T, subtree_end=self.subtree_end, run_ptr=run_ptr_d, runs=runs_d.view(-1, 4), ktile_qend=kq_d,
                                     dkv_units=units_d.view(-1, 4), dkv_splits=splits_d.view(-1, 4) if splits.shape[0] else None,
                                     n_slabs=n_slabs)

    def sequence_rows(self, attach_lists):
        """Packed row of every value the loss callbacks see, in callback order (leaf by leaf, the sequences folded onto it in attach-list
        order): (rows of logprobs[:len-1] = depths 1..len-1, rows of entropy[:len] = depths 0..len-1), two int64 device vectors."""
        lp_rows, ent_rows = [], []
        for i, attach_list in enumerate(attach_lists):
            path = self._paths_host[i]
            for _, length in attach_list:
                lp_rows.append(path[1:length]); ent_rows.append(path[:length])
        from ._staging import upload
        empty = np.zeros(0, np.int64)
        return upload([np.concatenate(lp_rows) if lp_rows else empty, np.concatenate(ent_rows) if ent_rows else empty], self.tokens.device, np.int64)

    def _expand(self, tokens, leaf_off, seg_off, seg_d0, par_seg, brk_ptr, brk_depth, brk_end, M, T):
        from ._lib import check, lib, ptr
        with ops._on(tokens, leaf_off, seg_off, self.tokens) as stream:
            st = lib().dta_preorder_meta(ptr(tokens), ptr(leaf_off), ptr(seg_off), ptr(seg_d0), ptr(par_seg), ptr(brk_ptr), ptr(brk_depth),
                                         ptr(brk_end), M, T, ptr(self.tokens), ptr(self.depth), ptr(self.parent), ptr(self.subtree_end), stream)
        check(st, "dta_preorder_meta")


class TreeTrainingEngine:
    def __init__(self, model_config, device, dtype: torch.dtype, max_seq_len: int, forward_only: bool = False):
        if torch.device(device).type == "cuda" and dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise TypeError(f"TreeTrainingEngine supports dtype torch.bfloat16 / torch.float16 (MFMA kernels) and torch.float32 (the reference's "
                            f"--dtype fp32, run.py:122-132: plain-FMA attention kernels, a gradient-check path), got {dtype}")
        self.model = None
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = dtype
        self.max_seq_len = max_seq_len
        self.forward_only = forward_only
        self.n_layers = model_config.num_hidden_layers
        self.n_kv_heads = getattr(model_config, "num_key_value_heads", 8)
        self.cur_len = 0                      # the packed engine is stateless between calls (reference: tte:550, 614-615)
        self.forkpos_list: List[int] = []
        self._loss_host = None                      # page-locked landing place of a call's loss value
        self.returns: List[Optional[torch.Tensor]] = []
        self.last_packed: Optional[_PackedTrie] = None
        self.head_chunk = 2048
        self.checkpoint_layers: Optional[bool] = None    # None = decide from free HBM
        self.mode = "auto"                               # "packed" (one pass), "stack" (block-wise push/pop walk), "auto" (by footprint)
        self.memory_budget_bytes: Optional[int] = None   # activation budget for "auto"; None = 80 % of the free HBM at call time
        self.last_mode: Optional[str] = None             # what the last backward() ran ("packed" / "packed+recompute" / "stack[B]")
        self.attn_keep_fraction = 0.25                   # of free HBM, for attention outputs kept across layer recomputation
        self.partial_recompute = True                    # under recomputation, leading layers keep full activations while HBM allows
        self.tp_group = None                             # set to a process group to split the LM-head vocabulary across it
        self._attn_keep_planned = 0
        self.stack_fp32_grads = True                     # block-wise walk: sum the blocks' parameter gradients in fp32 (False: model dtype, as the reference)

    # ------------------------------------------------------------------------------------------
    def _pack(self, token_trie) -> _PackedTrie:
        longest = max(token_trie.lens) if token_trie.lens else 0
        assert longest <= self.max_seq_len, (                                   # tte:162-164, 289-291
            f"Exceeds max_seq_len: cur_len=0, new_tokens={longest}, max={self.max_seq_len}")
        p0 = next(iter(self.model.parameters()), None) if hasattr(self.model, "parameters") else None
        if p0 is not None and (p0.dtype != self.dtype or p0.device != self.device):
            raise RuntimeError(f"model parameters are {p0.dtype} on {p0.device}; this engine was built for {self.dtype} on {self.device}")
        packed = _PackedTrie(token_trie, self.device, self.n_kv_heads)
        self.last_packed = packed
        return packed

    def _should_checkpoint(self, model, T: int) -> bool:
        if self.checkpoint_layers is not None:
            return self.checkpoint_layers
        c = model.config
        D = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
        per_tok = 2 * (10 * c.hidden_size + 4 * c.intermediate_size + 4 * (c.num_attention_heads + c.num_key_value_heads) * D)
        need = per_tok * self.n_layers * T
        if self.device.type != "cuda":
            return False
        free = self._free_hbm()
        # `per_tok` over-counts by about 1.3x (Qwen3-0.6B, 25.6k rows: 50 GB estimated, 38 GB of activations measured), so 0.75 of
        # the free HBM still leaves the head chunk, the gradient buffers and allocator slack their room
        return need > 0.75 * free

    def _free_hbm(self) -> int:
        """HBM this process can still use (ops.free_hbm: driver-free + the caching allocator's unused reserve, under the
        per-process allocator cap when one is set)."""
        return ops.free_hbm(self.device)

    def _per_token_layer_bytes(self, model) -> int:
        c = model.config
        D = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
        return 2 * (10 * c.hidden_size + 4 * c.intermediate_size + 4 * (c.num_attention_heads + c.num_key_value_heads) * D)

    def _budget(self) -> Optional[int]:
        if self.memory_budget_bytes is not None:
            return int(self.memory_budget_bytes)
        if self.device.type != "cuda":
            return None
        free = self._free_hbm()
        return int(0.8 * free)

    def _stack_block_rows(self, model, T: int, block_size: Optional[int]) -> Optional[int]:
        """None: the packed one-pass engine fits (possibly with per-layer recomputation).  Else the row budget of a block of the
        block-wise walk: at least the caller's `block_size` (the reference's memory knob, tte:489-513), more when HBM allows."""
        if self.mode == "packed":
            return None
        budget = self._budget()
        per_layer = self._per_token_layer_bytes(model)
        c = model.config
        if self.mode != "stack":
            if budget is None:
                return None
            # one pass with per-layer recomputation keeps two hidden rows per token and layer + one layer's activations + the head
            need = T * (2 * 2 * c.hidden_size * self.n_layers + 2 * per_layer)
            if need <= budget:
                return None
        bs = block_size or 2048
        if budget is None:
            return bs
        rows = int(0.5 * budget / max(per_layer * self.n_layers, 1)) // 256 * 256
        return int(max(bs, min(rows, 16384)))

    def _backward_stack(self, model, token_trie, loss_fn, block_rows: int) -> float:
        from . import dp
        from .stack_engine import StackWalk
        dp.defer_active_reducers()          # parameters accumulate once per block here: reduce after the walk, not from the hooks
        # fp32 gradient buffers of the walk: a fifth of the budget at most (Qwen3-0.6B: 2.4 GB; 14B would need 59 GB and may not get it)
        budget = self._budget()
        walk = StackWalk(model, token_trie, self.device, self.dtype, block_rows, self.head_chunk, self.tp_group,
                         grad_budget_bytes=0 if not self.stack_fp32_grads else (None if budget is None else budget // 5))
        total = walk.run(loss_fn)
        self.last_mode = f"stack[{block_rows}]x{walk.n_blocks}"
        self.cur_len = 0
        return float(total.item()) if total is not None else 0.0

    @staticmethod
    def packed_rows(n_tree_tokens: int) -> int:
        """Rows of the packed pass for a trie of `n_tree_tokens` (rounded up to the filler granularity, _PackedTrie)."""
        T = int(n_tree_tokens)
        if T >= _PackedTrie.PAD_FROM and T % _PackedTrie.PAD_TO:
            T += (-T) % _PackedTrie.PAD_TO
        return T

    def warm_gemm_shapes(self, model, row_counts) -> None:
        """Library warm-up, not a step: run the projection GEMMs of ONE decoder layer and the LM-head GEMMs (forward, dgrad, wgrad)
        once per distinct packed row count, on dummy rows and with the attention replaced by the identity.  hipBLASLt picks a
        solution per exact shape and loads its code object on first use — a one-off cost of up to ≈0.2 s per new row count when the
        files are cold (measured: one 452 ms step among 230 ms ones in a fresh process) that a training run amortises to nothing
        and a 6-step benchmark does not."""
        from types import SimpleNamespace
        body = model.model
        one = SimpleNamespace(config=model.config, model=SimpleNamespace(embed_tokens=body.embed_tokens, layers=[body.layers[0]], norm=body.norm))
        ident = lambda l: (lambda q, k, v: q)
        for T in sorted({int(t) for t in row_counts if t > 0}):
            tokens = torch.zeros(T, dtype=torch.long, device=self.device)
            depth = torch.arange(T, device=self.device, dtype=torch.int32) % 64
            parent = torch.arange(-1, T - 1, device=self.device, dtype=torch.int32)
            with torch.enable_grad():
                h = packed_hidden_states(one, tokens, depth, None, False, 0, ident)
                lp, ent = packed_logprob_entropy(h, head_weight(model), tokens, parent, True, self.head_chunk)
                (lp.sum() + ent.sum()).backward()
        for p in model.parameters():
            p.grad = None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _full_layers(self, model, T: int):
        """Under per-layer recomputation: how many (leading) layers can still keep their full activations.  Returned as a PLAN
        `bytes of layer 0 -> number of layers`: the pass runs layer 0 in full, measures what it really kept (the static estimate of
        `_should_checkpoint` is 1.3-1.5x too high), and the plan divides 0.7 of the HBM free at that moment - minus two layers of
        working set for the recomputation and the LM head's logits - by it.  0 when `partial_recompute` is off or recomputation
        was demanded explicitly (`checkpoint_layers = True` means every layer)."""
        if self.device.type != "cuda" or not self.partial_recompute or self.checkpoint_layers is not None:
            return 0
        c = model.config
        head = int(2.0 * T * c.vocab_size * 2 + 2 * 4 * c.vocab_size * c.hidden_size)      # logits (+ slack) and the fp32 head gradient
        esz = torch.empty(0, dtype=self.dtype).element_size()
        D = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
        stash_l = 2 * T * c.hidden_size * esz                                                # (res, delta) kept per RECOMPUTED layer
        attn_l = T * c.num_attention_heads * (D * esz + 4)                                   # (out, lse) kept per recomputed layer ...
        attn_keep = self._attn_keep_planned                                                  # ... within the budget already promised to them
        L = self.n_layers

        def plan(layer0_bytes: int) -> int:
            """Largest n such that n full layers, the stashes and kept attention outputs of the other L - n, two layers of working set and
            the LM head fit 0.7 of the HBM free now (layer 0 is already allocated).  (The first form of this round subtracted the WHOLE
            attention-keep promise - a quarter of the free HBM - and all L stashes whatever n: Qwen3-14B kept 14 of 40 layers at 138 GB
            peak where round 2, which subtracted nothing, kept 27 at 211 GB.)"""
            d0 = max(int(layer0_bytes), 1)
            avail = int(0.7 * self._free_hbm()) - 2 * d0 - head
            for n in range(L, 1, -1):
                rec = L - n
                if (n - 1) * d0 + rec * stash_l + min(attn_keep, rec * attn_l) <= avail:
                    return n
            return 1
        return plan

    def _attn_keep_bytes(self) -> int:
        """HBM budget for attention outputs kept across the per-layer recomputation (model.py:_LayerRecompute): a quarter
        of what is free when the pass starts, so the recomputation skips the forward attention kernel on as many layers
        as that pays for.  `attn_keep_fraction = 0` recomputes everything."""
        self._attn_keep_planned = 0
        if self.device.type != "cuda" or self.attn_keep_fraction <= 0:
            return 0
        free = self._free_hbm()
        self._attn_keep_planned = int(self.attn_keep_fraction * free)
        return self._attn_keep_planned

    def _path_losses(self, packed, token_trie, lp, ent, loss_fn):
        """Σ over original sequences of loss_fn(logprobs[:len-1], entropy[:len], attachment) (tte:379-398).  The root paths of
        ALL leaves are gathered by one indexing kernel per vector; every sequence then gets VIEWS of its leaf's path (a
        sequence folded onto a leaf is a prefix of it), and the user's callback runs per sequence as in the reference.
        The views are autograd LEAVES cut off from the model's graph - as the reference's `logprobs`/`entropy` buffers are
        (tte:383-398: detached storage, `grad_logprobs`/`grad_entropy` beside it): returns (total, [lp_view, ent_view, ...])
        for `_backprop_paths`."""
        lp_all, ent_all = lp.detach()[packed.path_cat], ent.detach()[packed.path_cat]
        terms, leaves, o = [], [], 0
        want = torch.is_grad_enabled() and (lp.requires_grad or ent.requires_grad)
        for i, attach_list in enumerate(token_trie.attach_lists):
            for attachment, length in attach_list:
                a, e = lp_all[o + 1:o + length], ent_all[o:o + length]
                if want:
                    a.requires_grad_(lp.requires_grad); e.requires_grad_(ent.requires_grad)
                terms.append(loss_fn(a, e, attachment)); leaves += [a, e]
            o += packed.path_sizes[i]
        return sum_loss_terms(terms, lp.device), leaves

    @staticmethod
    def _backprop_paths(packed, token_trie, lp, ent, total, leaves):
        """d(total)/d(params): the callbacks' small graph first (its leaves are the per-sequence views), then ONE accumulation of all
        per-sequence gradients onto packed rows (a sorted, deterministic index_put: nested prefixes of a leaf's path add up) and ONE
        backward through head and model - the reference assembles roots and gradients the same way (tte:404-440).  Without this the
        autograd of 2 S slices costs 2 S zero-fills, copies and full-length additions (≈300 launches of 4-5 µs at S = 48)."""
        total.backward()                               # fills .grad of the views (and of anything else the callbacks reached)
        rows = None
        roots, grads = [], []
        for k, src in ((0, lp), (1, ent)):
            gs = [t.grad for t in leaves[k::2]]
            if not src.requires_grad or all(g is None for g in gs):
                continue
            if rows is None:
                rows = packed.sequence_rows(token_trie.attach_lists)
            vals = torch.cat([g if g is not None else torch.zeros_like(t) for g, t in zip(gs, leaves[k::2])])
            acc = torch.zeros_like(src)
            acc.index_put_((rows[k],), vals.to(acc.dtype), accumulate=True)
            roots.append(src); grads.append(acc)
        if roots:
            torch.autograd.backward(roots, grads)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, model, token_trie) -> List[torch.Tensor]:
        """Per-sequence logprobs ``[len_i - 1]`` (fp32), indexed by original sequence id (tte:515-553)."""
        self.model = model
        self.returns = [None] * token_trie.n_sequences
        self.forkpos_list = _get_forkpos(None, token_trie.lcp_lens, None)
        if token_trie.n_sequences == 0:
            return self.returns
        with ops.weight_cache():
            return self._forward(model, token_trie)

    def _forward(self, model, token_trie):
        packed = self._pack(token_trie)
        h = packed_hidden_states(model, packed.tokens, packed.depth, packed.meta, False)
        lp, _ = packed_logprob_entropy(h, head_weight(model), packed.tokens, packed.parent, False, self.head_chunk,
                                       packed.fork_child, packed.fork_parent, self.tp_group, packed.fork_dev)
        for i, attach_list in enumerate(token_trie.attach_lists):
            lp_path = lp[packed.paths[i][1:]]
            for attachment, length in attach_list:
                self.returns[attachment["_sequence_batch_id"]] = lp_path[:length - 1].clone()
        self.cur_len = 0
        return self.returns

    def backward(self, model, token_trie, loss_fn: Callable, block_size: int, cut_f1_tail: bool = True) -> float:
        """Accumulates d(sum of per-sequence losses)/d(params) into ``param.grad``; returns the loss sum
        (tte:555-616)."""
        self.model = model
        lens = [int(ids.size(0)) for ids in token_trie.inputs]
        self.forkpos_list = _get_forkpos(lens, token_trie.lcp_lens, block_size)
        if token_trie.n_sequences == 0:          # an empty bin of a data-parallel step: no loss, no gradient (the caller still reduces)
            return 0.0
        with ops.weight_cache():                 # per-weight copies (stacked rows, transposes) are shared inside this call and dropped after it
            return self._backward(model, token_trie, loss_fn, block_size)

    def _backward(self, model, token_trie, loss_fn, block_size):
        n_tree = sum(token_trie.lens) - sum(token_trie.lcp_lens)
        rows = self._stack_block_rows(model, n_tree, block_size)
        if self.tp_group is not None:
            # every rank of a vocabulary-parallel group runs the same trie and issues collectives per head call: one pass or the
            # block-wise walk, and the walk's block size, must not depend on a rank's own free memory - the tightest rank decides
            import torch.distributed as dist
            v = torch.tensor([rows if rows is not None else (1 << 30)], device=self.device, dtype=torch.int32)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.tp_group)
            rows = None if int(v.item()) >= (1 << 30) else int(v.item())
        if rows is not None:
            longest = max(token_trie.lens)
            assert longest <= self.max_seq_len, f"Exceeds max_seq_len: cur_len=0, new_tokens={longest}, max={self.max_seq_len}"
            return self._backward_stack(model, token_trie, loss_fn, rows)
        packed = self._pack(token_trie)
        chunk = min(self.head_chunk, block_size) if block_size else self.head_chunk
        ckpt = self._should_checkpoint(model, packed.plan.T)
        self.last_mode = "packed+recompute" if ckpt else "packed"
        attn_keep = self._attn_keep_bytes() if ckpt else 0          # before the layer plan: the plan leaves this much alone
        full = self._full_layers(model, packed.plan.T) if ckpt else 0
        kept = []
        h = packed_hidden_states(model, packed.tokens, packed.depth, packed.meta, ckpt, attn_keep, None, full, kept)
        if ckpt and kept:
            self.last_mode = "packed" if kept[0] >= self.n_layers else f"packed+recompute[{self.n_layers - kept[0]}/{self.n_layers}]"
        lp, ent = packed_logprob_entropy(h, head_weight(model), packed.tokens, packed.parent, True, max(chunk, 1),
                                         packed.fork_child, packed.fork_parent, self.tp_group, packed.fork_dev)
        total, leaves = self._path_losses(packed, token_trie, lp, ent, loss_fn)
        if total is None:
            return 0.0
        # the loss value leaves the device as soon as the forward has produced it (an asynchronous copy into page-locked memory) and the
        # call returns when THAT copy has landed - with the backward launched but, on the device, still running: gradients are complete in
        # stream order (an optimizer step, a gradient reduction or the next call simply queue behind them), and the host work of the next
        # call (trie build, packing plan, uploads) overlaps this call's backward instead of leaving the device idle for it
        landed = None
        if total.is_cuda and os.environ.get("DTA_ASYNC_LOSS", "1") != "0":          # (env: diagnostic A/B switch)
            if self._loss_host is None:
                self._loss_host = torch.empty(1, dtype=torch.float32).pin_memory()
            self._loss_host.copy_(total.detach().reshape(1).float(), non_blocking=True)
            landed = torch.cuda.Event(); landed.record()
        if total.requires_grad:                     # (a callback that returns constants only leaves nothing to back-propagate)
            self._backprop_paths(packed, token_trie, lp, ent, total, leaves)
        self.cur_len = 0
        if landed is None:
            return float(total.item())
        landed.synchronize()
        return float(self._loss_host.item())
