#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in a hipcc -save-temps .s file (diagnostic tool).

usage: isa_mix.py file.s kernel_substring [min_mfma]
Finds the basic-block span between a backward-branch target and its branch that holds the most MFMAs and prints
counts per class and per-MFMA ratios (the budget: MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_"):
        return "ds_write"
    if op.startswith("global_load_lds") or ("lds" in op and op.startswith("buffer_load")):
        return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_cvt_pk"):
        return "cvt_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kname = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and kname in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = body[labels[m.group(1)]:i + 1]
            n = sum(1 for s in span if s.strip().startswith("v_mfma"))
            if best is None or n > best[0]:
                best = (n, labels[m.group(1)], i)
    n, a, b = best
    c = Counter()
    for l in body[a:b + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        c[classify(t.split()[0])] += 1
    tot = sum(c.values())
    print(f"{kname}: hottest loop lines {a}-{b} of kernel, {tot} instructions, {n} MFMA, {(tot - n) / max(n, 1):.2f} non-MFMA per MFMA")
    for k, v in c.most_common():
        print(f"  {k:9s} {v:5d}  {v / max(n, 1):.2f}/mfma")


if __name__ == "__main__":
    main()
