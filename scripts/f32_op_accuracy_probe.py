#!/usr/bin/env python3
"""Accuracy of the fp32 operators at REAL sizes against float64 torch on the same device (relative Frobenius error of outputs and gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamictreeattn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rel = lambda a, b: float((a.double() - b).norm() / b.norm())

# log-prob / entropy at the full vocabulary
R, V = 256, 151936
x = (torch.randn(R, V, generator=g, device=dev) * 0.7).requires_grad_(True)
lab = torch.randint(0, V, (R,), generator=g, device=dev)
glp, gent = torch.randn(R, generator=g, device=dev), torch.randn(R, generator=g, device=dev)
lp, ent = ops.logprob_entropy(x, lab, 1.0, True)
((lp * glp).sum() + (ent * gent).sum()).backward()
xd = x.detach().double().requires_grad_(True)
ls = torch.log_softmax(xd, -1)
lpd = ls.gather(-1, lab[:, None]).squeeze(-1); entd = -(ls.exp() * ls).sum(-1)
((lpd * glp.double()).sum() + (entd * gent.double()).sum()).backward()
print("logprob: lp", rel(lp, lpd.detach()), "ent", rel(ent, entd.detach()), "grad", rel(x.grad, xd.grad))
ge_only = torch.autograd.grad((ops.logprob_entropy(x, lab, 1.0, True)[1] * gent).sum(), x)[0]
ged = torch.autograd.grad((-(torch.log_softmax(xd, -1).exp() * torch.log_softmax(xd, -1)).sum(-1) * gent.double()).sum(), xd)[0]
print("logprob: entropy-only grad", rel(ge_only, ged))

# swiglu
T, C = 4096, 3072
gu = torch.randn(T, 2 * C, generator=g, device=dev).requires_grad_(True); gy = torch.randn(T, C, generator=g, device=dev)
y = ops.swiglu_fused(gu); y.backward(gy)
gud = gu.detach().double().requires_grad_(True)
yd = torch.nn.functional.silu(gud[:, :C]) * gud[:, C:]; yd.backward(gy.double())
print("swiglu: y", rel(y, yd.detach()), "dgate", rel(gu.grad[:, :C], gud.grad[:, :C]), "dup", rel(gu.grad[:, C:], gud.grad[:, C:]))

# rmsnorm
H = 1024
xx = torch.randn(T, H, generator=g, device=dev).requires_grad_(True); w = (1 + 0.1 * torch.randn(H, generator=g, device=dev)).requires_grad_(True); gy = torch.randn(T, H, generator=g, device=dev)
y = ops.rms_norm(xx, w, 1e-6); y.backward(gy)
xd_, wd = xx.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
yd = wd * (xd_ * torch.rsqrt(xd_.pow(2).mean(-1, keepdim=True) + 1e-6)); yd.backward(gy.double())
print("rmsnorm: y", rel(y, yd.detach()), "dx", rel(xx.grad, xd_.grad), "dw", rel(w.grad, wd.grad))

# head-norm + rope
Tn, NH = 2048, 16
q = torch.randn(Tn, NH, 128, generator=g, device=dev).requires_grad_(True); wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).requires_grad_(True)
depth = torch.randint(0, 16000, (Tn,), generator=g, device=dev).to(torch.int32)
cs = ops.rope_cos_sin(depth, 128, 1e6); gyq = torch.randn(Tn, NH, 128, generator=g, device=dev)
yq = ops.qk_norm_rope(q, wq, cs, 1e-6); yq.backward(gyq)
qd, wqd = q.detach().double().requires_grad_(True), wq.detach().double().requires_grad_(True)
inv = 1.0 / (1e6 ** (torch.arange(0, 128, 2, dtype=torch.float64, device=dev) / 128))
ang = depth.double()[:, None] * inv[None, :]
cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :]; sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
xn = wqd * (qd * torch.rsqrt(qd.pow(2).mean(-1, keepdim=True) + 1e-6))
ydq = xn * cos + torch.cat([-xn[..., 64:], xn[..., :64]], -1) * sin; ydq.backward(gyq.double())
print("qk_norm_rope: y", rel(yq, ydq.detach()), "dx", rel(q.grad, qd.grad), "dw", rel(wq.grad, wqd.grad), "| cos/sin table vs float64", rel(cs[:, :64], ang.cos()), rel(cs[:, 64:], ang.sin()))
