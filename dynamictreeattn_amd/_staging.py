"""Host -> HBM uploads of the small integer tables of a step (plan tables, sequence offsets, token ids).

`torch.from_numpy(x).to(device)` from pageable memory is a BLOCKING copy on the current stream (hipMemcpyWithStream): it waits
for everything already queued there, so each such call inside a step is a host synchronisation.  `upload` packs
any number of arrays into ONE reusable page-locked staging buffer per device and sends them with ONE asynchronous copy; the
returned tensors are views of one device buffer, each 16-byte aligned."""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import numpy as np
import torch

_ARENAS: Dict[tuple, dict] = {}
_RING = 6


def upload(arrays: Sequence[np.ndarray], device, dtype=np.int32) -> List[torch.Tensor]:
    device = torch.device(device)
    dt = np.dtype(dtype)
    per16 = 16 // dt.itemsize
    sizes = [int(a.size) for a in arrays]
    pads = [(-s) % per16 for s in sizes]
    total = sum(sizes) + sum(pads)
    tdt = torch.from_numpy(np.zeros(0, dt)).dtype
    pageable = device.type != "cuda" or os.environ.get("DTA_STAGING") == "pageable"      # env: diagnostic A/B switch (blocking copies, as round 1)
    if pageable:
        flat = np.zeros(total, dt)
        host = torch.from_numpy(flat)
    else:
        ring = _ARENAS.get((device.index, dt.str))
        if ring is None or ring["cap"] < total:
            # (re)allocate ALL slots at once, twice the largest request seen: page-locking is slow and synchronises, so it must
            # happen in the first step of a workload, not whenever a big table lands on a slot that has only held small ones
            if ring is not None:
                for ar_ in ring["slots"]:
                    if ar_["ev"] is not None:
                        ar_["ev"].synchronize()
            cap = max(2 * total, 1 << 16)
            ring = {"cap": cap, "next": 0, "slots": [{"buf": torch.empty(cap, dtype=tdt).pin_memory(), "ev": None} for _ in range(_RING)]}
            _ARENAS[(device.index, dt.str)] = ring
        i = ring["next"]; ring["next"] = (i + 1) % _RING           # a ring: consecutive uploads of one step never wait for each other
        ar = ring["slots"][i]
        if ar["ev"] is not None and not ar["ev"].query():
            ar["ev"].synchronize()                 # the upload issued _RING uploads ago out of this buffer has not left the host yet (never in practice)
        host = ar["buf"][:total]
        flat = host.numpy()
    o = 0
    for a, s, z in zip(arrays, sizes, pads):
        flat[o:o + s] = np.asarray(a).reshape(-1)
        if z:
            flat[o + s:o + s + z] = 0
        o += s + z
    if pageable:
        dev = host if device.type != "cuda" else host.to(device)
    else:
        with torch.cuda.device(device):
            dev = host.to(device, non_blocking=True)
            ar["ev"] = torch.cuda.Event(); ar["ev"].record(torch.cuda.current_stream(device))
    out, o = [], 0
    for s, z in zip(sizes, pads):
        out.append(dev[o:o + s]); o += s + z
    return out
