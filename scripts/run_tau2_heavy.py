#!/usr/bin/env python3
"""The heavier tau2-shaped variant of SURVEY §8(d) (G=16 rollouts x 12 turns, turn lengths 600-1800 so contexts reach the
16 384 cap), Qwen3-0.6B bf16, a few tree fwd+bwd calls: tokens/s and the attention rate at depth."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd.model import make_config
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine

dev = torch.device("cuda:0")
cfg = synth.QWEN3_0P6B
model = bench.build_model(cfg, dev, torch.bfloat16)
eng = TreeTrainingEngine(make_config(cfg), dev, torch.bfloat16, 16384)
tot_tok = tot_tree = tot_pairs = 0
tm = None
for step in range(4):
    seqs = synth.as_tensors(synth.tau2(seed=100 + step, G=16, turns=12, lo=600, hi=1800))
    if step == 1:                                   # step 0 is the warm-up
        tm = ops.KernelTimer(); ops.KernelTimer.active = tm
        torch.cuda.synchronize(); t0 = time.time()
    model.zero_grad(set_to_none=True)
    trie = TokenTrie(seqs, [dict(bench.ATTACH) for _ in seqs]); trie.backward_permute()
    loss = eng.backward(model, trie, bench.loss_fn, 2048)
    if step >= 1:
        st = trie.get_stats("backward", 2048)
        tot_tok += st["n_tokens"]; tot_tree += st["n_tree_tokens"]; tot_pairs += st["sum_depth"] + st["n_tree_tokens"]
torch.cuda.synchronize(); wall = time.time() - t0
attn_ms = sum(v[0] for v in tm.totals_ms().values())
print(json.dumps({"calls": 3, "n_tokens": tot_tok, "n_tree_tokens": tot_tree, "pairs": tot_pairs, "s_per_call": round(wall / 3, 3),
                  "tokens_per_s": round(tot_tok / wall), "tree_tokens_per_s": round(tot_tree / wall), "loss": loss,
                  "attention_share": round(attn_ms / 1e3 / wall, 3),
                  "attention_TFLOPs_14HqD": round(14 * 16 * 128 * tot_pairs * 28 / (attn_ms * 1e-3) / 1e12, 1),
                  "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                  "layer_checkpointing": eng._should_checkpoint(model, tot_tree // 3)}))
