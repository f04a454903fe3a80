"""Builds libdta_mi355x.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("DTA_LIB") or os.path.join(HERE, "libdta_mi355x.so")      # DTA_LIB: diagnostic builds only
SOURCES = ["tree_attn.hip", "trie_kernels.hip", "logprob_kernels.hip", "elementwise_kernels.hip"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "dta.h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
