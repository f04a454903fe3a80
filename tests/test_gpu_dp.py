"""GPU: the public data-parallel balancers on the real HIP TokenTrie (bit-exact bins), and the in-place gradient reducer
on device tensors (single-rank group on the one GPU of the box: layout, views, idempotence)."""
import types

import numpy as np
import pytest
import torch

from dynamictreeattn_amd import data_parallel as lb
from dynamictreeattn_amd import dp, synth
from dynamictreeattn_amd.tree_time_model import TreeTimeModel

pytestmark = pytest.mark.gpu


def test_public_balancers_on_gpu_tokentrie_bit_exact(balancer_golden):
    """LB_by_n_tokens / LB_by_TM / LB_by_DFS_and_TM (data_parallel.py:8-107) build their TokenTrie with the HIP kernels here;
    the bins equal the ones the reference produced (tests/golden/balancer_cases.json, scripts/make_golden.py)."""
    n = 0
    for item in balancer_golden["cases"]:
        seqs = synth.as_tensors(synth.make_case(item["case"]))
        for key, exp in item["bins"].items():
            K, mode, bs, kind = key.split("_")
            K = int(K[1:]); bs = None if bs == "None" else int(bs)
            tm = TreeTimeModel()
            if kind == "tm":
                tm.coeffs = np.array([3.0e-3, 1.0e-5, 4.0e-6, 2.0e-7, 1.5e-9])
            args = types.SimpleNamespace(K=K, mode=mode, block_size=bs)
            assert lb.LB_by_n_tokens(seqs, K) == exp["LB_by_n_tokens"]
            if exp["LB_by_TM"] is not None:
                assert lb.LB_by_TM(seqs, tm, args) == exp["LB_by_TM"], (item["case"], key); n += 1
            if exp["LB_by_DFS_and_TM"] is not None:
                assert lb.LB_by_DFS_and_TM(seqs, tm, args) == exp["LB_by_DFS_and_TM"], (item["case"], key); n += 1
    assert n > 20
    # dp.all_bins is the same thing by name; a bin may be empty when there are fewer leaves than ranks
    bins = dp.all_bins(synth.as_tensors([[1, 2, 3, 4], [1, 2, 3]]), 3, "backward", 2048)
    assert sorted(sum(bins, [])) == [0, 1] and [] in bins


def test_grad_reducer_views_and_layout_on_device():
    m = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.Linear(32, 8)).to("cuda:0", torch.bfloat16)
    red = dp.GradReducer(m.parameters(), bucket_bytes=3000, overlap=False)
    assert sum(f.numel() for f in red.flats) == sum(p.numel() for p in m.parameters())
    x = torch.randn(5, 64, device="cuda:0", dtype=torch.bfloat16)
    m(x).sum().backward()
    for p, v in red._views:
        assert p.grad.data_ptr() == v.data_ptr()             # autograd accumulated INTO the flat buffer
    g0 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)                             # a foreign zero_grad drops the views ...
    m(x).sum().backward()
    red.start()                                               # ... start() adopts the new gradients back into the flat buffers
    for (p, v), g in zip(red._views, g0):
        assert p.grad.data_ptr() == v.data_ptr()
    for p, g in zip(m.parameters(), g0):
        assert torch.equal(p.grad, g)
    red.close()
