#!/usr/bin/env python3
"""Sustained bf16 MFMA rate by shape on random and on zero operands (diagnostic; see scripts/diag/mfma_shape_probe.hip)."""
import ctypes as C, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "build", "libmfma_probe.so"))
lib.mfma_probe.restype = C.c_float
lib.mfma_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
dst = torch.empty(1 << 22, device=dev, dtype=torch.float32)
res = {}
for blocks, tag in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD")):
    for data in ("random", "zeros"):
        src = (torch.randn(1 << 19, device=dev) if data == "random" else torch.zeros(1 << 19, device=dev)).bfloat16()
        for shape, name in ((0, "32x32x16"), (1, "16x16x32")):
            iters = 4000
            ms = lib.mfma_probe(shape, blocks, iters, 20, src.data_ptr(), dst.data_ptr())
            torch.cuda.synchronize()
            flops = blocks * 4 * iters * 8 * 32768
            res[f"{tag} {data} {name}"] = {"ms": round(ms, 3), "TFLOPs": round(flops / (ms * 1e-3) / 1e12, 1)}
            print(tag, data, name, res[f"{tag} {data} {name}"], flush=True)
print(json.dumps(res))
