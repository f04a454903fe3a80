#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) into per-kernel
averages per launch.  Usage: python scripts/pmc_summary.py gpurun_out/pmc > table.json
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    per_dispatch = defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "tree_attn" not in name:
            continue
        short = name.split("(anonymous namespace)::")[1].split("<")[0].split("(")[0] if "(anonymous namespace)::" in name else name.split("(")[0]
        per_dispatch[(r["Dispatch_Id"], short, r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, short, cname), v in per_dispatch.items():
        acc[short][cname].append(v)
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[k]["launches_averaged"] = len(next(iter(cs.values())))
    if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        out[k]["hbm_bytes_per_launch"] = (2 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out[k] and "GRBM_GUI_ACTIVE" in out[k]:
        out[k]["mfma_busy_frac"] = out[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (out[k]["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_WAVE_CYCLES" in out[k]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if c in out[k]:
                out[k][c + "/SQ_WAVE_CYCLES"] = out[k][c] / out[k]["SQ_WAVE_CYCLES"]
    if out[k].get("SQ_INSTS_MFMA", 0) > 0:
        tot = sum(out[k].get(c, 0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"))
        out[k]["non_mfma_insts_per_mfma(valu+lds+salu; valu counts mfma)"] = (tot - out[k]["SQ_INSTS_MFMA"]) / out[k]["SQ_INSTS_MFMA"]
print(json.dumps(out, indent=1))
