"""Developer smoke (not collected by pytest): quick check of the attention kernels vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostmirror
from dynamictreeattn_amd import packing, synth, ops
from oracle import trie_oracle as to
from oracle.attn_oracle import tree_attention as ref_attn, rect_causal_attention

torch.manual_seed(0)
dev = torch.device("cuda:0")

def run_case(seqs, Hq=4, Hkv=2, dtype=torch.bfloat16, tag=""):
    t = to.TokenTrieOracle([np.array(s) for s in seqs]); t.backward_permute()
    plan = packing.plan_segments(t.lens, t.lcp_lens)
    _, depth, parent, se = hostmirror.expand_plan_host(plan)
    T = plan.T
    q = torch.randn(T, Hq, 128); k = torch.randn(T, Hkv, 128); v = torch.randn(T, Hkv, 128); do = torch.randn(T, Hq, 128)
    qd, kd, vd, dod = [x.to(dtype).to(dev) for x in (q, k, v, do)]
    qr, kr, vr = [x.to(dtype).float().requires_grad_(True) for x in (q, k, v)]
    o_ref, lse_ref = ref_attn(qr, kr, vr, torch.from_numpy(se).long())
    (o_ref * do.to(dtype).float()).sum().backward()
    se_d = torch.from_numpy(se).to(dev)
    meta = ops.meta_from_plan(plan, se_d, dev)
    qd.requires_grad_(True); kd.requires_grad_(True); vd.requires_grad_(True)
    o = ops.tree_attention(qd, kd, vd, meta)
    o.backward(dod)
    torch.cuda.synchronize()
    def rel(a, b): return float((a.float().cpu() - b).norm() / (b.norm() + 1e-9))
    print(f"{tag:28s} T={T:6d} M={plan.M:3d} out {rel(o, o_ref.detach()):.3e} dq {rel(qd.grad, qr.grad):.3e} dk {rel(kd.grad, kr.grad):.3e} dv {rel(vd.grad, vr.grad):.3e}", flush=True)

run_case([[1,2,3]], tag="single3")
run_case(synth.make_case({"kind": "random_tree", "seed": 5, "n_seq": 9, "max_len": 40, "alphabet": 2}), tag="tiny tree")
run_case(synth.make_case({"kind": "random_tree", "seed": 6, "n_seq": 40, "max_len": 300, "alphabet": 2, "dup": 2}), tag="tree 40x300")
run_case(synth.make_case({"kind": "tau2", "seed": 2, "V": 50, "G": 5, "sys_len": 300, "turns": 4, "lo": 30, "hi": 120, "cap": 2000}), tag="mini tau2")
run_case(synth.make_case({"kind": "wide", "seed": 1, "V": 1000, "root": 100, "branches": 12, "depth": 400}), tag="wide", dtype=torch.float16)
# stack form vs rect causal
for (start, B) in [(0, 64), (100, 37), (1000, 256)]:
    q, k, v, do = cases.attn_inputs(start, B)
    qd = q[0].transpose(0, 1).contiguous().bfloat16().to(dev).requires_grad_(True)     # [B,Hq,D]
    kd = k[0].transpose(0, 1).contiguous().bfloat16().to(dev).requires_grad_(True)
    vd = v[0].transpose(0, 1).contiguous().bfloat16().to(dev).requires_grad_(True)
    o = ops.tree_attention(qd, kd, vd, ops.stack_meta(start))
    o.backward(do[0].bfloat16().to(dev))
    qr = q[0].bfloat16().float().requires_grad_(True); kr = k[0].bfloat16().float().requires_grad_(True); vr = v[0].bfloat16().float().requires_grad_(True)
    oref = rect_causal_attention(qr, kr, vr, start)
    (oref * do[0].bfloat16().float()).sum().backward()
    def rel(a, b): return float((a.float().cpu() - b).norm() / (b.norm() + 1e-9))
    print(f"stack start={start} B={B}: out {rel(o, oref.detach()):.3e} dq {rel(qd.grad.transpose(0,1), qr.grad):.3e} dk {rel(kd.grad.transpose(0,1), kr.grad):.3e} dv {rel(vd.grad.transpose(0,1), vr.grad):.3e}", flush=True)
print("SMOKE DONE")
