"""GPU parity of the fused log-prob / entropy kernels (dta_logprob_entropy_fwd/bwd) and of the packed
LM-head operator built on them, against the oracle's fp32 arithmetic (vocab_parallel.py:13-27) and the
reference-generated values in tests/golden/logprob_cases.pt.  Inputs are rounded to the kernel dtype
first; outputs are fp32: |Δ| <= 2e-5·(1+|ref|) on values; bf16 gradient tensors within 1e-2 relative."""
import os

import numpy as np
import pytest
import torch

import hostmirror
from dynamictreeattn_amd import ops
from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("R,V,temp,dtype", [(37, 64, 1.0, torch.bfloat16), (16, 304, 0.7, torch.bfloat16), (9, 301, 1.3, torch.float16),
                                            (5, 151936, 1.0, torch.bfloat16), (1, 8, 1.0, torch.float16)])
def test_kernels_vs_oracle(R, V, temp, dtype):
    g = torch.Generator().manual_seed(R * 1000 + V)
    logits = (torch.randn(R, V, generator=g) * 3).to(dtype)
    labels = torch.randint(0, V, (R,), generator=g)
    glp, gex, gent = (torch.randn(R, generator=g) for _ in range(3))
    x = logits.float().requires_grad_(True)
    lp_ref, ent_ref = mo.logprobs_entropy_of(x, labels, temp)
    # a second picked label per row stands for fork children (its one-hot term is the caller's job)
    lab2 = (labels + 1) % V
    lp2 = mo.logprobs_of(x, lab2, temp)
    ((lp_ref * glp).sum() + (lp2 * gex).sum() + (ent_ref * gent).sum()).backward()
    ld = logits.to(DEV)
    if V % 8:
        buf = torch.zeros(R, (V + 7) // 8 * 8, dtype=dtype, device=DEV); buf[:, :V] = ld; ld = buf[:, :V]     # 16-B aligned row stride
    lse, ent, lp = ops.logprob_entropy_fwd_raw(ld, labels.to(DEV), True, temp)
    assert (lp.cpu() - lp_ref.detach()).abs().max() <= 2e-5 * (1 + lp_ref.abs().max())
    assert (ent.cpu() - ent_ref.detach()).abs().max() <= 5e-5 * (1 + ent_ref.abs().max())
    ops.logprob_entropy_bwd_raw(ld, labels.to(DEV), lse, ent, glp.to(DEV), gex.to(DEV), gent.to(DEV), temp)
    got = ld.float().cpu()
    got[torch.arange(R), lab2] += (gex / temp).to(dtype).float()
    ref = x.grad
    assert float((got - ref).norm() / ref.norm()) <= (1e-2 if dtype == torch.bfloat16 else 2e-3)


def test_reference_golden_values():
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    for name, c in gold.items():
        g = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["R"], c["V"], generator=g) * 3
        labels = torch.randint(0, c["V"], (c["R"],), generator=g)
        V8 = (c["V"] + 7) // 8 * 8
        buf = torch.zeros(c["R"], V8, dtype=torch.bfloat16, device=DEV); buf[:, :c["V"]] = logits.to(DEV)
        lse, ent, lp = ops.logprob_entropy_fwd_raw(buf[:, :c["V"]], labels.to(DEV), True, c["temp"])
        assert (lp.cpu() - c["logprobs"]).abs().max() < 0.06 and (ent.cpu() - c["entropy"]).abs().max() < 0.06      # bf16-rounded logits vs fp32 reference


def test_lm_head_rows_vs_plain_torch():
    g = torch.Generator().manual_seed(0)
    T, H, V = 700, 64, 1000
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16()
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16()
    nxt = torch.randint(0, V, (T,), generator=g)
    fork_rows = torch.tensor([3, 3, 250, 256, 699]); fork_tok = torch.randint(0, V, (5,), generator=g)
    go = [torch.randn(n, generator=g) for n in (T, 5, T)]
    def run(fn, dev, chunk, keep):
        hh, WW = h.detach().clone().to(dev).requires_grad_(True), W.detach().clone().to(dev).requires_grad_(True)
        bounds = np.searchsorted(fork_rows.numpy(), np.arange(0, T + chunk, chunk)).tolist()
        a, b, c = fn(hh, WW, nxt.to(dev), fork_rows.to(dev), fork_tok.to(dev), bounds, True, chunk, keep)
        ((a * go[0].to(dev)).sum() + (b * go[1].to(dev)).sum() + (c * go[2].to(dev)).sum()).backward()
        return [t.detach().float().cpu() for t in (a, b, c, hh.grad, WW.grad)]
    ref = run(hostmirror._cpu_lm_head_rows, "cpu", 256, None)
    for keep in (1 << 40, 0):                      # logits kept for backward / recomputed
        got = run(ops.lm_head_rows, DEV, 256, keep)
        for i, (x, y) in enumerate(zip(got, ref)):
            tol = 3e-3 if i < 3 else 2e-2
            assert float((x - y).norm() / (y.norm() + 1e-9)) <= tol, (keep, i)


def _tp_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the box: gloo, not RCCL
    from dynamictreeattn_amd import ops as _ops
    g = torch.Generator().manual_seed(0)
    T, H, V = 300, 64, 1024
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16().to(DEV).requires_grad_(True)
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16().to(DEV)
    nxt = torch.randint(0, V, (T,), generator=g).to(DEV)
    fork_rows = torch.tensor([3, 3, 250, 299]).to(DEV); fork_tok = torch.randint(0, V, (4,), generator=g).to(DEV)
    go = [torch.randn(n, generator=g).to(DEV) for n in (T, 4, T)]
    Vp = V // world
    Ws = W[rank * Vp:(rank + 1) * Vp].clone().requires_grad_(True)
    out = {}
    for keep in (1 << 40, 0):
        chunk = 128
        bounds = np.searchsorted(fork_rows.cpu().numpy(), np.arange(0, T + chunk, chunk)).tolist()
        h.grad = None; Ws.grad = None
        a, b, c = _ops.lm_head_rows(h, Ws, nxt, fork_rows, fork_tok, bounds, True, chunk, keep, tp_group=dist.group.WORLD, vocab_offset=rank * Vp)
        ((a * go[0]).sum() + (b * go[1]).sum() + (c * go[2]).sum()).backward()
        out[keep] = [t.detach().float().cpu() for t in (a, b, c, h.grad, Ws.grad)]
    torch.save(out, os.path.join(outdir, f"tp{rank}.pt"))
    dist.destroy_process_group()


def test_vocab_sharded_lm_head_two_ranks_vs_unsharded(tmp_path):
    """Config 4's logit split: two gloo ranks (sharing this box's GPU) each hold half of the vocabulary; per-shard
    HIP statistics + 2 all-reduces must reproduce the unsharded operator (values, dh, and dW on the own slice)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]; [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(0)
    T, H, V = 300, 64, 1024
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16().to(DEV).requires_grad_(True)
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16().to(DEV).requires_grad_(True)
    nxt = torch.randint(0, V, (T,), generator=g).to(DEV)
    fork_rows = torch.tensor([3, 3, 250, 299]).to(DEV); fork_tok = torch.randint(0, V, (4,), generator=g).to(DEV)
    go = [torch.randn(n, generator=g).to(DEV) for n in (T, 4, T)]
    bounds = np.searchsorted(fork_rows.cpu().numpy(), np.arange(0, T + 128, 128)).tolist()
    a, b, c = ops.lm_head_rows(h, W, nxt, fork_rows, fork_tok, bounds, True, 128, 1 << 40)
    ((a * go[0]).sum() + (b * go[1]).sum() + (c * go[2]).sum()).backward()
    ref = [t.detach().float().cpu() for t in (a, b, c, h.grad)]
    for rank in (0, 1):
        res = torch.load(os.path.join(str(tmp_path), f"tp{rank}.pt"), weights_only=True)
        for keep, got in res.items():
            for i in range(3):
                assert (got[i] - ref[i]).abs().max() <= 2e-3 * (1 + ref[i].abs().max()), (rank, keep, i)
            assert float((got[3] - ref[3]).norm() / ref[3].norm()) <= 1.5e-2
            dW_ref = W.grad.float().cpu()[rank * 512:(rank + 1) * 512]
            assert float((got[4] - dW_ref).norm() / dW_ref.norm()) <= 1.5e-2
