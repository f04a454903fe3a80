"""Weight-gradient GEMM of the Qwen3-0.6B projections at a tau2 packed-token count: dy^T @ x (what autograd issues for
F.linear) against x^T @ dy (transposed result) — hipBLASLt picks different kernels for the two layouts."""
import sys, torch
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
T = int(sys.argv[1]) if len(sys.argv) > 1 else 25482
dev, dt = "cuda", torch.bfloat16
SHAPES = {"0.6b": [("qkv", 1024, 4096), ("o", 2048, 1024), ("gate_up", 1024, 6144), ("down", 3072, 1024)],
          "4b": [("qkv", 2560, 6144), ("o", 4096, 2560), ("gate_up", 2560, 19456), ("down", 9728, 2560)]}
for name, K, N in SHAPES[sys.argv[2] if len(sys.argv) > 2 else "0.6b"]:
    for TT in (T, -(-T // 256) * 256):
        x = torch.randn(TT, K, device=dev, dtype=dt); dy = torch.randn(TT, N, device=dev, dtype=dt)
        fl = 2.0 * TT * K * N / 1e9
        a = timeit(lambda: dy.t() @ x)
        b = timeit(lambda: x.t() @ dy)
        xt = x.t().contiguous(); dyt = dy.t().contiguous()
        c = timeit(lambda: dyt @ x)            # A row-major [N,T]
        d = timeit(lambda: dyt @ xt.t())       # both K-contiguous (the forward's layout class)
        print(f"{name:8s} T={TT}: dy^T@x {a:.3f} ms {fl/a:6.0f} TF | x^T@dy {b:.3f} ms {fl/b:6.0f} TF | dyT_c@x {c:.3f} {fl/c:6.0f} TF | dyT_c@xT_c^T {d:.3f} {fl/d:6.0f} TF", flush=True)
