import sys, json, os
sys.path.insert(0, os.getcwd())
import torch
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import _PackedTrie
dev = torch.device("cuda:0"); Hq, Hkv, D = 16, 8, 128
seqs = synth.as_tensors(synth.tau2(0)); trie = TokenTrie(seqs); trie.backward_permute(); pk = _PackedTrie(trie, dev)
T = pk.plan.T; scale = D ** -0.5
for mode in ("random", "zeros", "random", "zeros"):
    g = torch.Generator(device=dev).manual_seed(0)
    if mode == "random":
        q, k, v, do = (torch.randn(T, H, D, generator=g, device=dev).bfloat16() for H in (Hq, Hkv, Hkv, Hq))
    else:
        q, k, v, do = (torch.zeros(T, H, D, device=dev).bfloat16() for H in (Hq, Hkv, Hkv, Hq))
    out, lse, _, _ = ops.attn_fwd_raw(q, k, v, pk.meta, scale)
    for _ in range(30): ops.attn_bwd_raw(q, k, v, out, do, lse, pk.meta, scale)       # ~0.1 s of load before timing
    tm = ops.KernelTimer(); ops.KernelTimer.active = tm
    for _ in range(20): ops.attn_bwd_raw(q, k, v, out, do, lse, pk.meta, scale)
    ms = tm.totals_ms(); ops.KernelTimer.active = None
    print(mode, {n: round(ms[n][0] / ms[n][1], 4) for n in ("bwd_dq", "bwd_dkv")}, flush=True)
