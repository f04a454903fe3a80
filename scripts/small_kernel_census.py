#!/usr/bin/env python3
"""Diagnostic: which torch ops (with input shapes) launch the many small kernels of one engine.backward call.
Prints torch.profiler's per-op table sorted by call count; run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from dynamictreeattn_amd import synth
from dynamictreeattn_amd.model import make_config
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine

dev = torch.device("cuda:0")
cfg = synth.QWEN3_0P6B
model = bench.build_model(cfg, dev, torch.bfloat16)
eng = TreeTrainingEngine(make_config(cfg), dev, torch.bfloat16, 16384)


def step(seed):
    seqs = synth.as_tensors(synth.tau2(seed=seed))
    trie = TokenTrie(seqs, [dict(bench.ATTACH) for _ in seqs])
    trie.backward_permute()
    model.zero_grad(set_to_none=True)
    return eng.backward(model, trie, bench.loss_fn, 2048)


for s in range(2):
    step(s)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print(f"{'op':44s} {'calls':>6s} {'self dev us':>12s} {'avg us':>8s}  shapes")
tot = 0
for e in rows[:70]:
    print(f"{e.key[:44]:44s} {e.count:6d} {e.self_device_time_total:12.0f} {e.self_device_time_total / e.count:8.1f}  {str(e.input_shapes)[:110]}")
small = [e for e in rows if e.self_device_time_total / e.count < 30]
print("small (<30 us avg) ops:", sum(e.count for e in small), "launch groups,", round(sum(e.self_device_time_total for e in small) / 1e3, 2), "ms device time")
