"""GPU parity of the HIP tree-attention kernels through the C ABI against the fp32 oracle.
Tolerances: inputs are rounded to bf16/f16 first, so the comparison isolates the kernel's own
error: relative Frobenius error <= 6e-3 (bf16) / 1e-3 (f16) on out, dq, dk, dv; lse abs <= 2e-3."""
import math
import os

import numpy as np
import pytest
import torch

import cases

import hostmirror
from dynamictreeattn_amd import ops, packing, synth
from oracle import trie_oracle as to
from oracle.attn_oracle import rect_causal_attention, tree_attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.bfloat16: 6e-3, torch.float16: 1e-3}


def _rel(a, b):
    b = b.detach()
    return float((a.detach().float().cpu() - b).norm() / max(float(b.norm()), 1e-2 * b.numel() ** 0.5))


def _trie_case(seqs, order="backward"):
    t = to.TokenTrieOracle([np.array(s) for s in seqs])
    getattr(t, order + "_permute")()
    plan = packing.plan_segments(t.lens, t.lcp_lens)
    _, depth, parent, se = hostmirror.expand_plan_host(plan)
    return plan, se


def _run(plan, se, Hq, Hkv, dtype, seed=0):
    T = plan.T
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = (torch.randn(T, H, 128, generator=g) for H in (Hq, Hkv, Hkv, Hq))
    qr, kr, vr = (x.to(dtype).float().requires_grad_(True) for x in (q, k, v))
    o_ref, lse_ref = tree_attention(qr, kr, vr, torch.from_numpy(se).long())
    (o_ref * do.to(dtype).float()).sum().backward()
    qd, kd, vd = (x.to(dtype).to(DEV).requires_grad_(True) for x in (q, k, v))
    meta = ops.meta_from_plan(plan, torch.from_numpy(se).to(DEV), DEV)
    out, lse, _, _ = ops.attn_fwd_raw(qd.detach(), kd.detach(), vd.detach(), meta, 128 ** -0.5)
    o = ops.tree_attention(qd, kd, vd, meta)
    o.backward(do.to(dtype).to(DEV))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert torch.equal(out, o.detach())                       # same launch, same bits
    assert _rel(o, o_ref) <= tol and _rel(qd.grad, qr.grad) <= tol and _rel(kd.grad, kr.grad) <= tol and _rel(vd.grad, vr.grad) <= tol
    assert (lse.cpu().t() * math.log(2.0) - lse_ref.detach()).abs().max() <= 2e-3
    return o, qd.grad, kd.grad, vd.grad


@pytest.mark.parametrize("case,hq,hkv,dtype", [
    ({"kind": "literal", "seqs": [[1]]}, 2, 1, torch.bfloat16),
    ({"kind": "literal", "seqs": [[1, 2, 3]]}, 4, 2, torch.bfloat16),
    ({"kind": "random_tree", "seed": 5, "n_seq": 9, "max_len": 40, "alphabet": 2}, 4, 2, torch.bfloat16),
    ({"kind": "random_tree", "seed": 6, "n_seq": 40, "max_len": 300, "alphabet": 2, "dup": 2}, 16, 8, torch.bfloat16),
    ({"kind": "random_tree", "seed": 7, "n_seq": 200, "max_len": 12, "alphabet": 3}, 4, 4, torch.bfloat16),      # wide & shallow
    ({"kind": "chain", "seed": 3, "n_seq": 6, "step": 77}, 4, 2, torch.bfloat16),
    ({"kind": "tau2", "seed": 2, "V": 50, "G": 5, "sys_len": 300, "turns": 4, "lo": 30, "hi": 120, "cap": 2000}, 16, 8, torch.bfloat16),
    ({"kind": "wide", "seed": 1, "V": 1000, "root": 100, "branches": 12, "depth": 400}, 8, 2, torch.float16),
    ({"kind": "config1", "seed": 0, "V": 1000, "prompt": 128, "rollouts": 4, "gen": 128}, 16, 8, torch.float16),
    ({"kind": "random_tree", "seed": 11, "n_seq": 25, "max_len": 260, "alphabet": 2}, 14, 2, torch.bfloat16),    # odd GQA group (7): one head per workgroup
])
def test_packed_tree_attention_vs_oracle(case, hq, hkv, dtype):
    for order in ("forward", "backward"):
        plan, se = _trie_case(synth.make_case(case), order)
        _run(plan, se, hq, hkv, dtype)


def test_bitwise_reproducible_and_accumulate():
    plan, se = _trie_case(synth.make_case({"kind": "random_tree", "seed": 9, "n_seq": 30, "max_len": 200, "alphabet": 2}))
    a = _run(plan, se, 4, 2, torch.bfloat16, seed=3)
    b = _run(plan, se, 4, 2, torch.bfloat16, seed=3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                                # no atomics anywhere: run-to-run identical
    # accumulate=1 adds into the caller's dk/dv (grad-KV stack semantics, tte:447-451)
    T = plan.T
    g = torch.Generator().manual_seed(1)
    q, k, v, do = (torch.randn(T, H, 128, generator=g).bfloat16().to(DEV) for H in (4, 2, 2, 4))
    meta = ops.meta_from_plan(plan, torch.from_numpy(se).to(DEV), DEV)
    out, lse, _, _ = ops.attn_fwd_raw(q, k, v, meta, 128 ** -0.5)
    dq, dk, dv = ops.attn_bwd_raw(q, k, v, out, do, lse, meta, 128 ** -0.5)
    base_k = torch.randn_like(dk); base_v = torch.randn_like(dv)
    dk2, dv2 = base_k.clone(), base_v.clone()
    ops.attn_bwd_raw(q, k, v, out, do, lse, meta, 128 ** -0.5, dk=dk2, dv=dv2, accumulate=True)
    assert _rel(dk2, (base_k.float() + dk.float()).cpu()) < 4e-3 and _rel(dv2, (base_v.float() + dv.float()).cpu()) < 4e-3


@pytest.mark.parametrize("start,B", [(0, 64), (100, 37), (300, 128), (1000, 256), (5, 1), (4095, 130)])
def test_stack_form_vs_rect_causal_oracle(start, B):
    """q_offset = start, no subtree bound: the reference's per-segment attention (tte:171-186).
    K/V are read IN PLACE from a head-major [Hkv, max_len, D] stack through the stride arguments."""
    Hq, Hkv = 4, 2
    q, k, v, do = cases.attn_inputs(start, B, Hq=Hq, Hkv=Hkv, seed=start + B)
    cap = start + B + 19
    kstack = torch.zeros(Hkv, cap, 128, dtype=torch.bfloat16, device=DEV); vstack = torch.zeros_like(kstack)
    kstack[:, :start + B] = k[0].bfloat16().to(DEV); vstack[:, :start + B] = v[0].bfloat16().to(DEV)
    kd = kstack[:, :start + B].transpose(0, 1).requires_grad_(True)         # [Tk, Hkv, D] view of the stack
    vd = vstack[:, :start + B].transpose(0, 1).requires_grad_(True)
    qd = q[0].bfloat16().to(DEV).transpose(0, 1).requires_grad_(True)       # [B, Hq, D] view of [Hq, B, D]
    o = ops.tree_attention(qd, kd, vd, ops.stack_meta(start))
    o.backward(do[0].bfloat16().to(DEV))
    qr, kr, vr = (x[0].bfloat16().float().requires_grad_(True) for x in (q, k, v))
    oref = rect_causal_attention(qr, kr, vr, start)
    (oref * do[0].bfloat16().float()).sum().backward()
    tol = TOL[torch.bfloat16]
    assert _rel(o, oref) <= tol
    assert _rel(qd.grad.transpose(0, 1), qr.grad) <= tol and _rel(kd.grad.transpose(0, 1), kr.grad) <= tol and _rel(vd.grad.transpose(0, 1), vr.grad) <= tol


def test_golden_attention_samples_from_transformers_eager():
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "attn_cases.pt"), weights_only=True)
    for name, c in gold.items():
        q, k, v, do = cases.attn_inputs(c["start"], c["B"], Hq=4, Hkv=2)
        qd, kd, vd = (x[0].transpose(0, 1).contiguous().bfloat16().to(DEV).requires_grad_(True) for x in (q, k, v))
        o = ops.tree_attention(qd, kd, vd, ops.stack_meta(c["start"]))
        o.backward(do[0].bfloat16().to(DEV))
        assert _rel(o, c["out"][0].float()) <= 1e-2                         # bf16 inputs vs fp32 eager
        assert _rel(qd.grad.transpose(0, 1), c["dq"][0].float()) <= 1.5e-2
        assert _rel(kd.grad.transpose(0, 1), c["dk"][0].float()) <= 1.5e-2 and _rel(vd.grad.transpose(0, 1), c["dv"][0].float()) <= 1.5e-2


def test_error_codes():
    from dynamictreeattn_amd._lib import lib
    assert lib().dta_tree_attn_fwd(None, None, None, None, None, None, None, None, 4, 4, 0, 2, 1, 128, 256, 128, 256, 0.1, 0, None) == -1
    x = torch.zeros(4, 2, 64, dtype=torch.bfloat16, device=DEV); l = torch.zeros(4, 2, device=DEV)
    P = lambda t: t.data_ptr()
    assert lib().dta_tree_attn_fwd(P(x), P(x), P(x), P(x), P(l), None, None, None, 4, 4, 0, 2, 2, 64, 128, 128, 128, 0.1, 0, None) == -2
    with pytest.raises(TypeError):
        ops.tree_attention(x.double(), x.double(), x.double(), ops.stack_meta(0))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.attn_fwd_raw(x.cpu(), x.cpu(), x.cpu(), ops.stack_meta(0), 1.0)


def test_full_size_properties_tau2():
    """BASELINE config-2 size (T = 25 482 packed tokens, Hq 16 / Hkv 8): size-independent checks —
    rows of one leaf path equal a dense causal run over that path; linearity in V."""
    plan, se = _trie_case(synth.tau2(0))
    T = plan.T
    assert T == 25482
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(T, H, 128, generator=g).bfloat16().to(DEV) for H in (16, 8, 8))
    meta = ops.meta_from_plan(plan, torch.from_numpy(se).to(DEV), DEV)
    o, lse, _, _ = ops.attn_fwd_raw(q, k, v, meta, 128 ** -0.5)
    o2, _, _, _ = ops.attn_fwd_raw(q, k, (2 * v.float()).bfloat16(), meta, 128 ** -0.5)
    assert _rel(o2, 2 * o.float().cpu()) < 4e-3
    leaf = plan.M - 1
    idx = np.concatenate([np.arange(b, e) for b, e in plan.path_runs[leaf]] + [np.arange(plan.seg_off[leaf], plan.seg_off[leaf + 1])])
    idx_d = torch.from_numpy(idx).to(DEV)
    od, _, _, _ = ops.attn_fwd_raw(q[idx_d].contiguous(), k[idx_d].contiguous(), v[idx_d].contiguous(), ops.stack_meta(0), 128 ** -0.5)
    assert _rel(o[idx_d], od.float().cpu()) < 4e-3


def test_full_size_config5_trie_attention_properties():
    """BASELINE config 5 at FULL size (64 branches x 16 384 deep over a 1 024-token root: 984 064 packed tree tokens, 8.56e9 visible pairs per
    head, fp16) through the packed tree-attention kernels, checked by size-independent properties against the fp32 oracle:
    the rows of ONE leaf's root path must equal plain causal attention over that leaf's 16 384 tokens (forward output, dQ; and dK/dV on
    the leaf's own tail, whose only queries are its own), for the first and the last leaf in DFS order."""
    from dynamictreeattn_amd.token_trie import TokenTrie
    from dynamictreeattn_amd.tree_training_engine import _PackedTrie
    from dynamictreeattn_amd import synth
    from oracle import attn_oracle
    Hq, Hkv, D, dtype = 2, 1, 128, torch.float16
    seqs = synth.as_tensors(synth.wide(seed=1))
    t = TokenTrie(seqs); t.backward_permute()
    st = t.get_stats("backward", 2048)
    assert (st["n_tokens"], st["n_tree_tokens"], st["sum_depth"]) == (1048576, 984064, 8556412416)
    pk = _PackedTrie(t, torch.device(DEV), Hkv)
    T = pk.plan.T
    g = torch.Generator(device=DEV).manual_seed(5)
    q, do = (torch.randn(T, Hq, D, generator=g, device=DEV).to(dtype) for _ in range(2))
    k, v = (torch.randn(T, Hkv, D, generator=g, device=DEV).to(dtype) for _ in range(2))
    out, lse, _, _ = ops.attn_fwd_raw(q, k, v, pk.meta, D ** -0.5)
    dq, dk, dv = ops.attn_bwd_raw(q, k, v, out, do, lse, pk.meta, D ** -0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all()
    for leaf in (0, len(t.lens) - 1):
        path = pk.paths[leaf]
        assert path.numel() == 16384
        M = len(t.lens)
        excl = max(t.lcp_lens[leaf - 1] if leaf > 0 else 0, t.lcp_lens[leaf] if leaf < M - 1 else 0)     # depths >= excl belong to this leaf alone
        assert excl == 1024
        qs, ks, vs, dos = (x[path].float().cpu().transpose(0, 1).contiguous().requires_grad_(r) for x, r in ((q, True), (k, True), (v, True), (do, False)))
        ref = attn_oracle.rect_causal_attention(qs, ks, vs, 0)      # [B, Hq, D] fp32
        ref.backward(dos.transpose(0, 1))
        rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
        assert rel(out[path], ref.detach()) <= 2e-3, leaf
        assert rel(dq[path], qs.grad.transpose(0, 1)) <= 4e-3, leaf
        # the part of a leaf below every fork is seen by that leaf's queries only: its dK/dV equal the dense ones
        assert rel(dk[path[excl:]], ks.grad.transpose(0, 1)[excl:]) <= 4e-3, leaf
        assert rel(dv[path[excl:]], vs.grad.transpose(0, 1)[excl:]) <= 4e-3, leaf


@pytest.mark.parametrize("form", ["3", "4"])
def test_alternative_forward_forms_pass_the_same_cases(form):
    """`DTA_FWD_FORM=4`: the 8-wave forward whose second head group runs half a tile behind the first (three-slot ring).  `DTA_FWD_FORM=3` selects the forward in which a workgroup is 4 waves and every wave carries BOTH query heads of a kv group (one wave
    per SIMD, 512 registers).  The switch is read once per process, so the packed / stack / golden cases of this file run
    again in a child process under it."""
    import subprocess, sys
    env = dict(os.environ, DTA_FWD_FORM=form)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "packed_tree_attention or stack_form or golden_attention or bitwise"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
