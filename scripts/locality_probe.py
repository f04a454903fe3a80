#!/usr/bin/env python3
"""Diagnostic: how much of the dK/dV kernel's time is the Q/dO stream coming from beyond L2?  Same launch, same arithmetic, but every query
row aliases ONE row of memory (token stride 0), so the whole Q/dO stream is an L2 hit.  Results are meaningless; only the time is read.
Caveat found on first use: identical rows also toggle fewer bits in the MFMA operands, and the chip then holds a higher clock —
the dQ kernel, which does not stream Q at all, got 15 % faster too (0.99 -> 0.84 ms; dkv2 1.35 -> 1.25 ms), so the gain is an UPPER bound on locality."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import _PackedTrie
dev = torch.device("cuda:0"); Hq, Hkv, D = 16, 8, 128
seqs = synth.as_tensors(synth.tau2(0)); trie = TokenTrie(seqs); trie.backward_permute(); pk = _PackedTrie(trie, dev)
T = pk.plan.T; scale = D ** -0.5
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(T, H, D, generator=g, device=dev).bfloat16() for H in (Hq, Hkv, Hkv, Hq))
out, lse, _, _ = ops.attn_fwd_raw(q, k, v, pk.meta, scale)
q1 = q[:1].expand(T, Hq, D); do1 = do[:1].expand(T, Hq, D); out1 = out[:1].expand(T, Hq, D)
for name, (qq, oo, dd) in (("real rows", (q, out, do)), ("aliased rows", (q1, out1, do1)), ("real rows", (q, out, do)), ("aliased rows", (q1, out1, do1))):
    for _ in range(10): ops.attn_bwd_raw(qq, k, v, oo, dd, lse, pk.meta, scale)
    tm = ops.KernelTimer(); ops.KernelTimer.active = tm
    for _ in range(20): ops.attn_bwd_raw(qq, k, v, oo, dd, lse, pk.meta, scale)
    ms = tm.totals_ms(); ops.KernelTimer.active = None
    print(name, {n: round(ms[n][0] / ms[n][1], 4) for n in ("bwd_dq", "bwd_dkv")}, "q stride", qq.stride(), flush=True)
