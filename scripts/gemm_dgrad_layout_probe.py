#!/usr/bin/env python3
"""dgrad GEMMs dx[T, in] = dy[T, out] . W[out, in]: today's operand layout (W row-major: the contraction index strides by `in`) against a
transposed copy of the weight (contraction index contiguous in both operands, the forward's layout class).  usage: python scripts/gemm_dgrad_layout_probe.py [T]"""
import sys, json, torch


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


T = int(sys.argv[1]) if len(sys.argv) > 1 else 28160
dev, dt = "cuda", torch.bfloat16
res = {}
for name, K, N in [("qkv", 1024, 4096), ("o", 2048, 1024), ("gate_up", 1024, 6144), ("down", 3072, 1024), ("head", 1024, 151936)]:
    W = torch.randn(N, K, device=dev, dtype=dt); dy = torch.randn(T, N, device=dev, dtype=dt)
    WT = W.t().contiguous()
    fl = 2.0 * T * K * N / 1e9
    a = timeit(lambda: dy @ W)
    b = timeit(lambda: dy @ WT.t())
    c = timeit(lambda: W.t().contiguous())
    res[name] = {"today_ms": a, "today_TF": fl / a, "transposed_copy_ms": b, "transposed_copy_TF": fl / b, "copy_ms": c}
    print(name, {k: round(v, 3) for k, v in res[name].items()}, flush=True)
print(json.dumps(res))
