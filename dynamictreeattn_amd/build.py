"""Builds libdta_mi355x.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("DTA_LIB") or os.path.join(HERE, "libdta_mi355x.so")      # DTA_LIB: diagnostic builds only
SOURCES = ["tree_attn.hip", "tree_attn_f32.hip", "trie_kernels.hip", "logprob_kernels.hip", "elementwise_kernels.hip"]


FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17"]
DEPS = SOURCES + ["dta_common.h"]


def _digest() -> str:
    """sha256 over the sources, the headers and the compile flags: the .so is rebuilt exactly when one of them changed
    (mtimes do not survive a snapshot copy to the GPU box)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in [os.path.join(CSRC, s) for s in DEPS] + [os.path.join(os.path.dirname(HERE), "include", "dta.h")]:
        if os.path.exists(path):
            h.update(path[len(os.path.dirname(HERE)):].encode()); h.update(open(path, "rb").read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".sha256"):
        return True
    return open(LIB + ".sha256").read().strip() != _digest()


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc] + FLAGS + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(LIB + ".sha256", "w") as f:
        f.write(_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
