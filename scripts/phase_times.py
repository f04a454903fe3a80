#!/usr/bin/env python3
"""Diagnostic: wall time of each host/device phase of one bench step (synchronising between phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dynamictreeattn_amd import synth
from dynamictreeattn_amd.model import make_config, packed_hidden_states, head_weight
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine, _PackedTrie, packed_logprob_entropy

dev = torch.device("cuda:0")
cfg = synth.QWEN3_0P6B
model = bench.build_model(cfg, dev, torch.bfloat16)
eng = TreeTrainingEngine(make_config(cfg), dev, torch.bfloat16, 16384)
def sync(): torch.cuda.synchronize(); return time.time()
for it in range(3):
    seqs = synth.as_tensors(synth.tau2(seed=it))
    model.zero_grad(set_to_none=True)
    t0 = sync()
    trie = TokenTrie(seqs, [dict(bench.ATTACH) for _ in seqs]); t1 = sync()
    trie.backward_permute(); t2 = sync()
    pk = _PackedTrie(trie, dev); t3 = sync()
    h = packed_hidden_states(model, pk.tokens, pk.depth, pk.meta, False); t4h = time.time(); t4 = sync()
    lp, ent = packed_logprob_entropy(h, head_weight(model), pk.tokens, pk.parent, True, 2048, pk.fork_child, pk.fork_parent); t5 = sync()
    tot, leaves = eng._path_losses(pk, trie, lp, ent, bench.loss_fn); t6 = sync()
    eng._backprop_paths(pk, trie, lp, ent, tot, leaves); t7h = time.time(); t7 = sync()
    print(f"iter {it}: trie {1e3*(t1-t0):.1f}  permute {1e3*(t2-t1):.1f}  pack {1e3*(t3-t2):.1f}  fwd {1e3*(t4-t3):.1f} (host enqueue {1e3*(t4h-t3):.1f})  "
          f"head {1e3*(t5-t4):.1f}  loss {1e3*(t6-t5):.1f}  bwd {1e3*(t7-t6):.1f} (host {1e3*(t7h-t6):.1f})  total {1e3*(t7-t0):.1f} ms", flush=True)
