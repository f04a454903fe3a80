"""Time the projection GEMMs of one Qwen3-0.6B layer (+ the LM head) at a packed-token count T as it comes out of a
tau2 trie and at T rounded up to multiples of 128/256: decides whether padding the packed axis pays.
Usage: python scripts/gemm_shapes.py [T] — prints one line per (shape, T, pass) in TFLOP/s."""
import sys
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


def main():
    T0 = int(sys.argv[1]) if len(sys.argv) > 1 else 25482
    if len(sys.argv) > 2:                                   # "cublas" = rocBLAS, "cublaslt" = hipBLASLt (torch's default here)
        torch.backends.cuda.preferred_blas_library(sys.argv[2])
        print("preferred_blas_library:", torch.backends.cuda.preferred_blas_library())
    dev, dt = "cuda", torch.bfloat16
    shapes = [("qkv", 1024, 4096), ("o", 2048, 1024), ("gate_up", 1024, 6144), ("down", 3072, 1024), ("head", 1024, 151936)]
    Ts = sorted({T0, -(-T0 // 128) * 128, -(-T0 // 256) * 256, -(-T0 // 2048) * 2048})
    tot = {T: 0.0 for T in Ts}
    for name, K, N in shapes:
        W = torch.randn(N, K, device=dev, dtype=dt)
        for T in Ts:
            x = torch.randn(T, K, device=dev, dtype=dt); dy = torch.randn(T, N, device=dev, dtype=dt)
            f = timeit(lambda: torch.nn.functional.linear(x, W))
            d = timeit(lambda: dy @ W)
            w = timeit(lambda: dy.t() @ x)
            fl = 2.0 * T * K * N / 1e9
            mult = 1 if name == "head" else 28
            tot[T] += (f + d + w) * mult
            print(f"{name:8s} T={T:6d} fwd {f:7.3f} ms {fl / f:7.1f} TF | dgrad {d:7.3f} ms {fl / d:7.1f} TF | wgrad {w:7.3f} ms {fl / w:7.1f} TF", flush=True)
            del x, dy
        del W
    for T in Ts:
        print(f"T={T}: projection+head GEMM time per step {tot[T]:.1f} ms")


if __name__ == "__main__":
    main()
