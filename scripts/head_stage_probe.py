#!/usr/bin/env python3
"""f1 experiment: does the LM-head stage get faster when the [rows, V] logits never live in HBM?

The stage = logits GEMM -> dta_logprob_entropy_fwd -> (loss) -> dta_logprob_entropy_bwd in place -> dgrad GEMM + wgrad GEMM, at the
bench's shape (T packed rows x hidden 1024 x V 151 936, bf16).  Variants:
  kept        ONE [T, V] logits buffer (7.7-8.6 GB) written by the GEMM, re-read by the statistics kernel, rewritten as dlogits, read by
              both gradient GEMMs (today's default when it fits a quarter of the free HBM)
  chunk N     rows go N at a time; the chunk's logits ([512, V] bf16 = 155 MB: inside the 256 MB Infinity Cache) are produced, reduced,
              dropped, and RECOMPUTED in the backward, where dlogits feed the two gradient GEMMs chunk by chunk; the weight gradient is
              summed over the chunks in fp32, either by a separate add (+tmp) or inside the GEMM (addmm with an fp32 C/D matrix)
Prints one JSON object: ms per stage (forward, backward, total; mean of `reps` after a warm-up) and the peak HBM allocated."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamictreeattn_amd import ops


def run(T, V, H, chunk, keep, fused, reps=4):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    h = (torch.randn(T, H, generator=g, device=dev) * 0.5).bfloat16().requires_grad_(True)
    W = (torch.randn(V, H, generator=g, device=dev) * 0.02).bfloat16().requires_grad_(True)
    nxt = torch.randint(0, V, (T,), generator=g, device=dev)
    rows = np.sort(np.random.default_rng(0).choice(T, 40, replace=False)).astype(np.int64)
    fork_rows = torch.from_numpy(rows).to(dev); fork_tok = torch.randint(0, V, (40,), generator=g, device=dev)
    ptr = np.zeros(T + 1, np.int32); np.cumsum(np.bincount(rows, minlength=T), out=ptr[1:])
    fork_ptr = torch.from_numpy(ptr).to(dev)
    bounds = np.searchsorted(rows, np.arange(0, T + chunk, chunk)).tolist()
    go = [torch.randn(n, generator=g, device=dev) for n in (T, 40, T)]
    ops.HEAD_WGRAD_FUSED_ACCUMULATE = fused
    tf, tb = [], []
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    for it in range(reps + 1):
        h.grad = None; W.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        a, b, c = ops.lm_head_rows(h, W, nxt, fork_ptr, fork_tok, fork_rows, bounds, True, chunk, (1 << 62) if keep else 0)
        loss = (a * go[0]).sum() + (b * go[1]).sum() + (c * go[2]).sum()
        e[1].record()
        loss.backward()
        e[2].record()
        torch.cuda.synchronize()
        if it:
            tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
    out = {"fwd_ms": float(np.mean(tf)), "bwd_ms": float(np.mean(tb)), "total_ms": float(np.mean(tf) + np.mean(tb)),
           "peak_extra_gb": (torch.cuda.max_memory_allocated(dev) - base) / 1e9,
           "check": [float(h.grad.float().norm()), float(W.grad.float().norm())]}
    del h, W
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    T = int(os.environ.get("T", 28160)); V, H = 151936, 1024
    only = os.environ.get("ONLY")            # e.g. "kept" or "chunk512f": one variant (for a rocprofv3 --pmc pass)
    variants = [("kept", T, True, False)] + [(f"chunk{c}{'f' if f else ''}", c, False, f) for c in (512, 1024, 2048, 4096) for f in (False, True)]
    res = {"T": T, "V": V, "H": H, "dtype": "bf16", "logits_gb": T * V * 2 / 1e9}
    for name, chunk, keep, fused in variants:
        if only and name != only:
            continue
        try:
            res[name] = run(T, V, H, chunk, keep, fused)
        except Exception as ex:           # e.g. addmm with an fp32 output not supported by this hipBLASLt build
            res[name] = {"error": str(ex)[:200]}
        print(name, res[name], file=sys.stderr, flush=True)
    print(json.dumps(res))
