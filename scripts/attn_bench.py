#!/usr/bin/env python3
"""Micro-benchmark of the three tree-attention kernels on one tau2-16k-shaped packed trie
(Qwen3-0.6B head geometry), random data, HIP-event timed.  Usage: python scripts/attn_bench.py [iters] [case]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import _PackedTrie

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
case = sys.argv[2] if len(sys.argv) > 2 else "tau2"
dev = torch.device("cuda:0")
Hq, Hkv, D = 16, 8, 128
seqs = synth.as_tensors(synth.tau2(0) if case == "tau2" else synth.wide(1, branches=16, depth=8192) if case == "wide" else synth.config1(0))
trie = TokenTrie(seqs); trie.backward_permute()
pk = _PackedTrie(trie, dev)
st = trie.get_stats("backward", 2048)
T = pk.plan.T; pairs = st["sum_depth"] + st["n_tree_tokens"]
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, do = (torch.randn(T, H, D, generator=g, device=dev).bfloat16() for H in (Hq, Hkv, Hkv, Hq))
scale = D ** -0.5
out, lse, _, _ = ops.attn_fwd_raw(q, k, v, pk.meta, scale)
res = {}
def timeit(name, fn, coef):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    res[name] = {"ms": round(ms, 4), "TFLOPs": round(coef * Hq * D * pairs / (ms * 1e-3) / 1e12, 1)}
timeit("fwd", lambda: ops.attn_fwd_raw(q, k, v, pk.meta, scale), 4)
tm = ops.KernelTimer(); ops.KernelTimer.active = tm
for _ in range(iters + 1): ops.attn_bwd_raw(q, k, v, out, do, lse, pk.meta, scale)
ms = tm.totals_ms(); ops.KernelTimer.active = None
for name, coef in (("bwd_dq", 2), ("bwd_dkv", 8)):
    t = ms[name][0] / ms[name][1]
    res[name] = {"ms": round(t, 4), "TFLOPs": round(coef * Hq * D * pairs / (t * 1e-3) / 1e12, 1)}
fin = ms["bwd_dkv_finalize"]
res["bwd_dkv_finalize"] = {"ms": round(fin[0] / fin[1], 4) if fin[1] else 0.0}
tot = res["fwd"]["ms"] + res["bwd_dq"]["ms"] + res["bwd_dkv"]["ms"] + res["bwd_dkv_finalize"]["ms"]
res["all_14HqD"] = {"ms": round(tot, 4), "TFLOPs": round(14 * Hq * D * pairs / (tot * 1e-3) / 1e12, 1)}
print(json.dumps({"case": case, "T": T, "pairs": pairs, **res}))
