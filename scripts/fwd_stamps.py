#!/usr/bin/env python3
"""Where a forward tile's cycles go, per wave (diagnostic build -DDTA_STAMP=1; DTA_LIB must point at it).
Segments per 64-key tile: 0 DMA issue of the next tile, 1 QK^T (16 ds_read_b128 + 16 MFMA issued), 2 mask + row max + O rescale (waits for the
QK results), 3 exp / pack + PV (32 transposed reads + 16 MFMA issued) + row sums, 4 DMA wait + barrier, 5 loop bookkeeping.
Read the SHARES, not the length (the stamps' fences forbid overlaps the real kernel has).  usage: DTA_LIB=build/libdta_stamp.so python scripts/fwd_stamps.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynamictreeattn_amd import ops, synth
from dynamictreeattn_amd._lib import lib
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import _PackedTrie

dev = torch.device("cuda:0")
seqs = synth.as_tensors(synth.tau2(0))
trie = TokenTrie(seqs); trie.backward_permute()
pk = _PackedTrie(trie, dev)
T = pk.plan.T
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = (torch.randn(T, H, 128, generator=g, device=dev).bfloat16() for H in (16, 8, 8))
nqt = (T + 127) // 128
n_wg = nqt * 8
buf = torch.zeros(n_wg * 8 * 8, dtype=torch.int64, device=dev)
fn = lib().dta_debug_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
for _ in range(3):
    ops.attn_fwd_raw(q, k, v, pk.meta, 128 ** -0.5)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); ops.attn_fwd_raw(q, k, v, pk.meta, 128 ** -0.5); b.record(); torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(n_wg, 8, 8).astype(np.float64)
tiles = d[:, :, 6]
live = tiles[:, 0] > 8                       # workgroups with a real sweep
names = ["dma_issue", "qk", "mask_max_rescale", "exp_pack_pv", "dmawait_barrier", "bookkeeping"]
out = {"launch_ms_with_stamps": a.elapsed_time(b), "workgroups": int(live.sum()), "mean_tiles_per_wg": float(tiles[live, 0].mean())}
for grp, sl in (("waves0-3", slice(0, 4)), ("waves4-7", slice(4, 8))):
    seg = d[live][:, sl, :6].sum(axis=(0, 1)); n = tiles[live][:, sl].sum()
    out[grp] = {"cycles_per_tile": {nm: round(float(x / n), 1) for nm, x in zip(names, seg)}, "total_per_tile": round(float(seg.sum() / n), 1),
                "share": {nm: round(float(x / seg.sum()), 3) for nm, x in zip(names, seg)}}
print(json.dumps(out, indent=1))
