#!/usr/bin/env python3
"""Weight-gradient GEMMs of the Qwen3-0.6B projections at a packed row count T: dW[out, in] = dy^T[out, T] . x[T, in] has FEW output tiles and
a LONG K (out x in = 4096x1024 .. 1024x3072 -> 16..96 tiles of 256x256 for 256 CUs; K = T ~ 28k).  Times what ops._Linear issues today
against a manual split-K: the T rows cut into S equal slices, ONE batched GEMM over the slices (S x as many workgroups), partial products
summed.  usage: python scripts/gemm_splitk_probe.py [T]"""
import json, sys
import torch


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


def splitk(dy, x, S, transposed, f32):
    T = x.shape[0]
    per = (T // S) // 256 * 256
    body = per * S
    a, b = (x, dy) if transposed else (dy, x)            # result = a^T @ b
    A = a[:body].view(S, per, a.shape[1]).transpose(1, 2)   # [S, M, per]
    B = b[:body].view(S, per, b.shape[1])                    # [S, per, N]
    part = torch.bmm(A, B, out_dtype=torch.float32) if f32 else torch.bmm(A, B)
    out = part.sum(0, dtype=torch.float32)
    if body < T:
        out = out + (a[body:].t() @ b[body:]).float()
    out = out.to(dy.dtype)
    return out.t() if transposed else out


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 28160
    dev, dt = "cuda", torch.bfloat16
    res = {"T": T}
    for name, K, N in [("qkv", 1024, 4096), ("o", 2048, 1024), ("gate_up", 1024, 6144), ("down", 3072, 1024)]:
        x = torch.randn(T, K, device=dev, dtype=dt); dy = torch.randn(T, N, device=dev, dtype=dt)
        transposed = K >= 2 * N                                   # ops._Linear's layout choice
        base_fn = (lambda: (x.t() @ dy).t()) if transposed else (lambda: dy.t() @ x)
        ref = base_fn().float()
        fl = 2.0 * T * K * N / 1e9
        row = {"today_ms": timeit(base_fn), "tiles_256": (N // 256) * (K // 256)}
        row["today_TF"] = fl / row["today_ms"]
        for S in (2, 3, 4, 6, 8):
            for f32 in (False, True):
                key = f"S{S}{'_f32' if f32 else ''}"
                try:
                    out = splitk(dy, x, S, transposed, f32)
                    err = float((out.float() - ref).norm() / ref.norm())
                    ms = timeit(lambda: splitk(dy, x, S, transposed, f32))
                    row[key] = {"ms": ms, "TF": fl / ms, "rel_diff_vs_today": err}
                except Exception as ex:
                    row[key] = {"error": str(ex)[:120]}
        res[name] = row
        print(name, json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
