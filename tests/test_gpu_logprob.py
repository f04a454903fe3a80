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


def _csr(rows, R):
    return torch.searchsorted(rows, torch.arange(R + 1)).to(torch.int32)


@pytest.mark.parametrize("R,V,temp,dtype", [(37, 64, 1.0, torch.bfloat16), (16, 304, 0.7, torch.bfloat16), (9, 301, 1.3, torch.float16),
                                            (5, 151936, 1.0, torch.bfloat16), (1, 8, 1.0, torch.float16), (11, 1000, 0.9, torch.float32),
                                            (3, 151936, 1.0, torch.float32)])
def test_kernels_vs_oracle(R, V, temp, dtype):
    """fwd + in-place bwd incl. the EXTRA picks (fork children: CSR over the rows) that the kernels pick and back-propagate."""
    g = torch.Generator().manual_seed(R * 1000 + V)
    logits = (torch.randn(R, V, generator=g) * 3).to(dtype)
    labels = torch.randint(0, V, (R,), generator=g)
    ex_rows = torch.sort(torch.randint(0, R, (min(R, 7),), generator=g)).values          # rows may repeat: a node with several fork children
    ex_lab = (labels[ex_rows] + 1 + torch.arange(ex_rows.numel())) % V                    # distinct from the row's own label and from each other
    glp, gent = (torch.randn(R, generator=g) for _ in range(2))
    gex = torch.randn(ex_rows.numel(), generator=g)
    x = logits.float().requires_grad_(True)
    lp_ref, ent_ref = mo.logprobs_entropy_of(x, labels, temp)
    lp_all = torch.log_softmax(x / temp, -1)
    lp2_ref = lp_all[ex_rows, ex_lab]
    ((lp_ref * glp).sum() + (lp2_ref * gex).sum() + (ent_ref * gent).sum()).backward()
    ld = logits.to(DEV)
    if V % 8:
        buf = torch.zeros(R, (V + 7) // 8 * 8, dtype=dtype, device=DEV); buf[:, :V] = ld; ld = buf[:, :V]     # 16-B aligned row stride
    ptr_, exl = _csr(ex_rows, R).to(DEV), ex_lab.to(DEV)
    lp2 = torch.empty(ex_rows.numel(), dtype=torch.float32, device=DEV)
    lse, ent, lp = ops.logprob_entropy_fwd_raw(ld, labels.to(DEV), True, temp, ptr_, exl, lp2)
    assert (lp.cpu() - lp_ref.detach()).abs().max() <= 2e-5 * (1 + lp_ref.abs().max())
    assert (lp2.cpu() - lp2_ref.detach()).abs().max() <= 2e-5 * (1 + lp2_ref.abs().max())
    assert (ent.cpu() - ent_ref.detach()).abs().max() <= 5e-5 * (1 + ent_ref.abs().max())
    ref = x.grad
    tol = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 2e-5}[dtype]
    out = torch.empty_strided(ld.shape, ld.stride(), dtype=ld.dtype, device=DEV)          # out of place first (the caller's logits stay intact) ...
    ops.logprob_entropy_bwd_raw(ld, labels.to(DEV), lse, ent, glp.to(DEV), gent.to(DEV), temp, ptr_, exl, gex.to(DEV), out=out)
    assert torch.equal(ld.float().cpu(), logits.float())
    assert float((out.float().cpu() - ref).norm() / ref.norm()) <= tol
    ops.logprob_entropy_bwd_raw(ld, labels.to(DEV), lse, ent, glp.to(DEV), gent.to(DEV), temp, ptr_, exl, gex.to(DEV))      # ... then in place
    assert torch.equal(ld.float().cpu(), out.float().cpu())


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


def test_public_gather_logprobs_on_hip_vs_reference_golden():
    """The PUBLIC vocab_parallel.gather_logprobs(_entropy) (reference vocab_parallel.py:399-467) on CUDA tensors runs the HIP
    kernels: fp32 logits reproduce the reference's own recorded values and gradients (tests/golden/logprob_cases.pt,
    written by scripts/make_golden.py from the imported reference) to fp32 accuracy."""
    from dynamictreeattn_amd import vocab_parallel as vp
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    for name, c in gold.items():
        g = torch.Generator().manual_seed(c["seed"])
        logits = (torch.randn(c["R"], c["V"], generator=g) * 3).to(DEV).requires_grad_(True)
        labels = torch.randint(0, c["V"], (c["R"],), generator=g).to(DEV)
        go_lp = torch.randn(c["R"], generator=g).to(DEV); go_ent = torch.randn(c["R"], generator=g).to(DEV)    # scripts/make_golden.py:207
        lp, ent = vp.gather_logprobs_entropy(logits, labels, temperature=c["temp"])
        assert lp.dtype == torch.float32 and lp.shape == labels.shape and ent.shape == labels.shape
        assert (lp.detach().cpu() - c["logprobs"]).abs().max() <= 2e-5 * (1 + c["logprobs"].abs().max()), name
        assert (ent.detach().cpu() - c["entropy"]).abs().max() <= 5e-5 * (1 + c["entropy"].abs().max()), name
        lp_only = vp.gather_logprobs(logits, labels, temperature=c["temp"])
        assert (lp_only.detach().cpu() - c["logprobs_only"]).abs().max() <= 2e-5 * (1 + c["logprobs_only"].abs().max()), name
        # the reference recorded d(Σ lp·go_lp)/dlogits and d(Σ ent·go_ent)/dlogits separately (first 48 columns + |row| sums)
        (lp * go_lp).sum().backward(retain_graph=True)
        g1 = logits.grad.clone().cpu(); logits.grad = None
        (ent * go_ent).sum().backward()
        g2 = logits.grad.cpu()
        assert torch.allclose(g1[:, :48], c["grad_lp_head"], atol=5e-6, rtol=1e-4), name          # fp32 kernels (v_exp_f32 / v_log_f32) vs torch fp32
        assert torch.allclose(g2[:, :48], c["grad_ent_head"], atol=5e-6, rtol=1e-4), name
        assert torch.allclose(g1.abs().sum(-1), c["grad_lp_rowsum_abs"], rtol=2e-4), name
        assert torch.allclose(g2.abs().sum(-1), c["grad_ent_rowsum_abs"], rtol=2e-4, atol=1e-5), name


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3), (torch.float32, 2e-5)])
def test_public_functions_shape_quirk_and_grads_vs_oracle(dtype, tol):
    """[1,B,V] logits with [1,B-1] labels (tte:190-193, 256-261: logprobs cover the first B-1 rows, entropy all B),
    temperature != 1, an unaligned vocabulary (V % 8 != 0), and the gradient through the public functions."""
    from dynamictreeattn_amd import vocab_parallel as vp
    g = torch.Generator().manual_seed(7)
    B, V, temp = 29, 301, 0.8
    logits = (torch.randn(1, B, V, generator=g) * 2).to(dtype)
    labels = torch.randint(0, V, (1, B - 1), generator=g)
    g1, g2 = torch.randn(1, B - 1, generator=g), torch.randn(1, B, generator=g)
    x = logits.float().clone().requires_grad_(True)
    lp_all = torch.log_softmax(x / temp, -1)
    lp_ref = lp_all[:, :B - 1].gather(-1, labels[..., None]).squeeze(-1)
    ent_ref = -(lp_all.exp() * lp_all).sum(-1)
    ((lp_ref * g1).sum() + (ent_ref * g2).sum()).backward()
    ld = logits.detach().to(DEV).requires_grad_(True)
    lp, ent = vp.gather_logprobs_entropy(ld, labels.to(DEV), temperature=temp)
    assert lp.shape == (1, B - 1) and ent.shape == (1, B)
    assert (lp.detach().cpu() - lp_ref.detach()).abs().max() <= 2e-5 * (1 + lp_ref.abs().max())
    assert (ent.detach().cpu() - ent_ref.detach()).abs().max() <= 5e-5 * (1 + ent_ref.abs().max())
    ((lp * g1.to(DEV)).sum() + (ent * g2.to(DEV)).sum()).backward()
    assert ld.grad.dtype == dtype and ld.grad.shape == ld.shape
    assert float((ld.grad.float().cpu() - x.grad).norm() / x.grad.norm()) <= tol
    # dense.py's call form: [B, V] logits with [B] labels (dense.py:17-20)
    lp2 = vp.gather_logprobs(ld.detach()[0], torch.randint(0, V, (B,), generator=g).to(DEV), temperature=temp)
    assert lp2.shape == (B,)


def test_cuda_logits_never_fall_back_to_torch():
    """CUDA logits always run the HIP kernels (no silent torch path): with the library handle broken the call raises."""
    from dynamictreeattn_amd import _lib, vocab_parallel as vp
    x = torch.randn(4, 64, device=DEV); lab = torch.randint(0, 64, (4,), device=DEV)
    saved, _lib._lib = _lib._lib, None
    saved_path = _lib.LIB
    try:
        _lib.LIB = "/nonexistent/libdta.so"
        with pytest.raises(RuntimeError):
            vp.gather_logprobs(x, lab)
    finally:
        _lib.LIB, _lib._lib = saved_path, saved


def test_lm_head_rows_vs_plain_torch():
    g = torch.Generator().manual_seed(0)
    T, H, V = 700, 64, 1000
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16()
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16()
    nxt = torch.randint(0, V, (T,), generator=g)
    fork_rows = torch.tensor([3, 3, 250, 256, 699]); fork_tok = torch.tensor([5, 9, 77, 500, 999])
    go = [torch.randn(n, generator=g) for n in (T, 5, T)]
    def run(fn, dev, chunk, keep):
        hh, WW = h.detach().clone().to(dev).requires_grad_(True), W.detach().clone().to(dev).requires_grad_(True)
        bounds = np.searchsorted(fork_rows.numpy(), np.arange(0, T + chunk, chunk)).tolist()
        a, b, c = fn(hh, WW, nxt.to(dev), _csr(fork_rows, T).to(dev), fork_tok.to(dev), fork_rows.to(dev), bounds, True, chunk, keep)
        ((a * go[0].to(dev)).sum() + (b * go[1].to(dev)).sum() + (c * go[2].to(dev)).sum()).backward()
        return [t.detach().float().cpu() for t in (a, b, c, hh.grad, WW.grad)]
    ref = run(hostmirror._cpu_lm_head_rows, "cpu", 256, None)
    for keep in (1 << 40, 0):                      # logits kept for backward / recomputed
        got = run(ops.lm_head_rows, DEV, 256, keep)
        for i, (x, y) in enumerate(zip(got, ref)):
            tol = 3e-3 if i < 3 else 2e-2
            assert float((x - y).norm() / (y.norm() + 1e-9)) <= tol, (keep, i)


def _tp_worker(rank, world, port, outdir, T=300, V=1024, chunk=128):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the box: gloo, not RCCL
    from dynamictreeattn_amd import ops as _ops
    g = torch.Generator().manual_seed(0)
    H = 64
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16().to(DEV).requires_grad_(True)
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16().to(DEV)
    nxt = torch.randint(0, V, (T,), generator=g).to(DEV)
    fork_rows = torch.tensor([3, 3, 250, T - 1]).to(DEV); fork_tok = torch.tensor([1, V // 2 + 3, 17, V - 1]).to(DEV)
    fork_ptr = torch.searchsorted(fork_rows, torch.arange(T + 1, device=DEV)).to(torch.int32)
    go = [torch.randn(n, generator=g).to(DEV) for n in (T, 4, T)]
    Vp = V // world
    Ws = W[rank * Vp:(rank + 1) * Vp].clone().requires_grad_(True)
    out = {}
    for keep in (1 << 40, 0):
        bounds = np.searchsorted(fork_rows.cpu().numpy(), np.arange(0, T + chunk, chunk)).tolist()
        h.grad = None; Ws.grad = None
        a, b, c = _ops.lm_head_rows(h, Ws, nxt, fork_ptr, fork_tok, fork_rows, bounds, True, chunk, keep, tp_group=dist.group.WORLD, vocab_offset=rank * Vp)
        ((a * go[0]).sum() + (b * go[1]).sum() + (c * go[2]).sum()).backward()
        out[keep] = [t.detach().float().cpu() for t in (a, b, c, h.grad, Ws.grad)]
    torch.save(out, os.path.join(outdir, f"tp{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("T,V,chunk", [(300, 1024, 128), (2048, 2 * 18992, 1024)])
def test_vocab_sharded_lm_head_two_ranks_vs_unsharded(tmp_path, T, V, chunk):
    """Config 4's logit split: two gloo ranks (sharing this box's GPU) each hold half of the vocabulary; per-shard
    HIP statistics + 2 all-reduces must reproduce the unsharded operator (values, dh, and dW on the own slice).  The second case
    has the REAL shard shape of Qwen3 over 8 ranks: [rows, 151 936 / 8 = 18 992] bf16 per rank, 1024-row chunks."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, str(tmp_path), T, V, chunk)) for r in range(2)]
    [p.start() for p in procs]; [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(0)
    H = 64
    h = (torch.randn(T, H, generator=g) * 0.5).bfloat16().to(DEV).requires_grad_(True)
    W = (torch.randn(V, H, generator=g) * 0.2).bfloat16().to(DEV).requires_grad_(True)
    nxt = torch.randint(0, V, (T,), generator=g).to(DEV)
    fork_rows = torch.tensor([3, 3, 250, T - 1]).to(DEV); fork_tok = torch.tensor([1, V // 2 + 3, 17, V - 1]).to(DEV)
    fork_ptr = torch.searchsorted(fork_rows, torch.arange(T + 1, device=DEV)).to(torch.int32)
    go = [torch.randn(n, generator=g).to(DEV) for n in (T, 4, T)]
    bounds = np.searchsorted(fork_rows.cpu().numpy(), np.arange(0, T + chunk, chunk)).tolist()
    a, b, c = ops.lm_head_rows(h, W, nxt, fork_ptr, fork_tok, fork_rows, bounds, True, chunk, 1 << 40)
    ((a * go[0]).sum() + (b * go[1]).sum() + (c * go[2]).sum()).backward()
    ref = [t.detach().float().cpu() for t in (a, b, c, h.grad)]
    Vp = V // 2
    for rank in (0, 1):
        res = torch.load(os.path.join(str(tmp_path), f"tp{rank}.pt"), weights_only=True)
        for keep, got in res.items():
            for i in range(3):
                assert (got[i] - ref[i]).abs().max() <= 2e-3 * (1 + ref[i].abs().max()), (rank, keep, i)
            assert float((got[3] - ref[3]).norm() / ref[3].norm()) <= 1.5e-2
            dW_ref = W.grad.float().cpu()[rank * Vp:(rank + 1) * Vp]
            assert float((got[4] - dW_ref).norm() / dW_ref.norm()) <= 1.5e-2
