#!/usr/bin/env python3
"""BASELINE config 5 at full size: 64-way branching x 16k depth (root 1024 shared), fp16, Qwen3-0.6B dims.
One tree fwd+bwd step; prints sizes, wall time, peak HBM and the achieved attention rate."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd.model import make_config
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine

branches = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = synth.QWEN3_0P6B
model = bench.build_model(cfg, dev, torch.float16)
eng = TreeTrainingEngine(make_config(cfg), dev, torch.float16, 16384)
seqs = synth.as_tensors(synth.wide(seed=1, branches=branches))
# fp16 needs a loss scale at this depth (see tests/test_gpu_engine.py::test_wide_deep_trie_with_layer_checkpointing)
att = [{"w_logprobs": -1024.0, "w_entropy": 102.4} for _ in seqs]
tm = ops.KernelTimer(); ops.KernelTimer.active = tm
torch.cuda.synchronize(); t0 = time.time()
trie = TokenTrie(seqs, att); trie.backward_permute()
t1 = time.time()
loss = eng.backward(model, trie, bench.loss_fn, 2048)
torch.cuda.synchronize(); t2 = time.time()
st = trie.get_stats("backward", 2048)
ms = tm.totals_ms()
pairs = st["sum_depth"] + st["n_tree_tokens"]
attn_ms = sum(v[0] for v in ms.values())
gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters())))
print(json.dumps({"n_sequences": st["n_sequences"], "n_tokens": st["n_tokens"], "n_tree_tokens": st["n_tree_tokens"], "sum_depth": st["sum_depth"],
                  "trie_build_s": round(t1 - t0, 3), "backward_s": round(t2 - t1, 2), "tokens_per_s": round(st["n_tokens"] / (t2 - t0)),
                  "loss": loss, "grad_norm": gn, "finite": bool(torch.isfinite(torch.tensor(gn))),
                  "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                  "attention_s": round(attn_ms / 1e3, 2), "attention_TFLOPs_14HqD": round(14 * 16 * 128 * pairs * 28 / (attn_ms * 1e-3) / 1e12, 1),
                  "layer_checkpointing": eng._should_checkpoint(model, st["n_tree_tokens"])}))
