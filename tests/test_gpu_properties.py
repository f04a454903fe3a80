"""GPU property tests (hypothesis): random rollouts through the HIP trie kernels and the HIP tree attention, against the oracle.
Few examples per test — every example is a handful of launches — but drawn fresh over sequence counts, lengths, duplicates,
prefix-of-each-other pairs and head geometries that the fixed cases of test_gpu_trie.py / test_gpu_attention.py do not enumerate."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import test_gpu_attention as ga
from dynamictreeattn_amd.token_trie import TokenTrie
from oracle import trie_oracle as to

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])

rollouts = st.lists(st.lists(st.integers(0, 3), min_size=1, max_size=70), min_size=1, max_size=14)


@settings(max_examples=40, **COMMON)
@given(rollouts)
def test_hip_tokentrie_equals_the_oracle(seqs):
    t = TokenTrie([torch.tensor(s, dtype=torch.int64) for s in seqs], [{"i": i} for i in range(len(seqs))], device=DEV)
    o = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in seqs], [{"i": i} for i in range(len(seqs))])
    for step in ("sorted", "forward", "backward"):
        if step != "sorted":
            getattr(t, step + "_permute")(); getattr(o, step + "_permute")()
        assert [x.tolist() for x in t.inputs] == [x.tolist() for x in o.inputs], step
        assert list(t.lens) == list(o.lens) and list(t.lcp_lens) == list(o.lcp_lens), step
        assert [[(a["_sequence_batch_id"], n) for a, n in al] for al in t.attach_lists] == \
               [[(a["_sequence_batch_id"], n) for a, n in al] for al in o.attach_lists], step


@settings(max_examples=12, **COMMON)
@given(rollouts, st.sampled_from([(2, 1), (4, 2), (6, 2), (8, 8)]), st.sampled_from(["forward", "backward"]), st.integers(0, 3))
def test_hip_tree_attention_equals_the_oracle_on_random_tries(seqs, heads, order, seed):
    plan, se = ga._trie_case(seqs, order)
    ga._run(plan, se, heads[0], heads[1], torch.bfloat16, seed=seed)


@settings(max_examples=12, **COMMON)
@given(rollouts, st.sampled_from([(2, 1), (4, 2), (6, 2)]), st.sampled_from(["forward", "backward"]), st.integers(0, 3))
def test_fp32_tree_attention_equals_the_oracle_on_random_tries(seqs, heads, order, seed):
    """The plain-FMA fp32 kernels (tree_attn_f32.hip) on random tries: out, lse and all three gradients at 2e-5 against the fp32 oracle."""
    import math
    import hostmirror
    from dynamictreeattn_amd import ops, packing
    from oracle.attn_oracle import tree_attention
    t = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in seqs]); getattr(t, order + "_permute")()
    plan = packing.plan_segments(t.lens, t.lcp_lens)
    _, _, _, se = hostmirror.expand_plan_host(plan)
    g = torch.Generator().manual_seed(seed)
    q, k, v, do = (torch.randn(plan.T, H, 128, generator=g) for H in (heads[0], heads[1], heads[1], heads[0]))
    qr, kr, vr = (x.clone().requires_grad_(True) for x in (q, k, v))
    o_ref, lse_ref = tree_attention(qr, kr, vr, torch.from_numpy(se).long())
    (o_ref * do).sum().backward()
    qd, kd, vd = (x.to(DEV).requires_grad_(True) for x in (q, k, v))
    meta = ops.meta_from_plan(plan, torch.from_numpy(se).to(DEV), DEV)
    _, lse, _, _ = ops.attn_fwd_raw(qd.detach(), kd.detach(), vd.detach(), meta, 128 ** -0.5)
    o = ops.tree_attention(qd, kd, vd, meta)
    o.backward(do.to(DEV))
    rel = lambda a, b: float((a.detach().cpu() - b.detach()).norm() / max(float(b.detach().norm()), 1e-3 * b.numel() ** 0.5))
    assert rel(o, o_ref) <= 2e-5 and rel(qd.grad, qr.grad) <= 2e-5 and rel(kd.grad, kr.grad) <= 2e-5 and rel(vd.grad, vr.grad) <= 2e-5
    assert (lse.cpu().t() * math.log(2.0) - lse_ref.detach()).abs().max() <= 2e-5


@settings(max_examples=25, **COMMON)
@given(st.integers(1, 40), st.integers(1, 40), st.sampled_from([torch.bfloat16, torch.float32]), st.integers(0, 5))
def test_transpose_kernel_on_random_shapes(rb, cb, dtype, seed):
    """dta_transpose on ragged tile counts: rows / cols any multiple of the 16-byte vector (8 or 4 elements)."""
    from dynamictreeattn_amd import ops
    v = 8 if dtype == torch.bfloat16 else 4
    R, C = rb * v, cb * v
    x = torch.randn(R, C, generator=torch.Generator().manual_seed(seed)).to(dtype).to(DEV)
    assert torch.equal(ops.transpose_2d(x), x.t().contiguous())


@settings(max_examples=10, **COMMON)
@given(st.integers(33, 80), st.sampled_from([(1024, 512), (256, 768), (512, 1024)]), st.booleans(), st.integers(0, 3))
def test_split_k_weight_gradient_equals_the_single_gemm(t256, shape, transposed, seed):
    """ops._wgrad: the 4-slice batched GEMM with fp32 partials (+ the remainder rows) against the plain product in fp32."""
    from dynamictreeattn_amd import ops
    T, (out_f, in_f) = 256 * t256, shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, in_f, generator=g).bfloat16().to(DEV); dy = torch.randn(T, out_f, generator=g).bfloat16().to(DEV)
    got = ops._wgrad(x, dy, transposed).float()
    ref = dy.float().t() @ x.float()
    assert got.shape == ref.shape
    assert float((got - ref).norm() / ref.norm()) <= 3e-3                  # one bf16 rounding of the result


rollouts2 = st.lists(st.lists(st.integers(0, 3), min_size=2, max_size=70), min_size=1, max_size=14)      # >= 2 tokens: run.py's loss is a mean over len-1 logprobs


@settings(max_examples=30, **COMMON)
@given(rollouts2, st.sampled_from(["packed", "stack"]), st.sampled_from([7, 16, 2048]), st.sampled_from(["backward", "forward", "sorted"]), st.integers(0, 2))
def test_fp32_engine_equals_the_reference_schedule_on_random_tries(seqs, mode, block, order, wseed):
    """The whole hot path in fp32 on random rollouts (duplicates, prefixes of each other, single sequences, ...): HIP TokenTrie -> permute ->
    engine.backward (one packed pass or the block-wise walk with small blocks: fork picks, pending picks, cut tails) against the ORACLE's
    restatement of the reference's push / pop schedule on the same leaf order - loss and every parameter gradient at 1e-4 (fp32 leaves four
    orders of magnitude below what a bf16 comparison can resolve, so indexing or bookkeeping slips cannot hide in rounding noise)."""
    import cases
    from dynamictreeattn_amd.model import Qwen3TreeLM
    from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine
    from oracle import model_oracle as mo
    cfg = cases.TINY_CFGS["d128"]
    w = mo.init_weights(cfg, seed=10 + wseed)
    att = lambda: [{"w_logprobs": -1.0 - 0.01 * i, "w_entropy": 0.1 + 0.003 * i} for i in range(len(seqs))]
    t = TokenTrie([torch.tensor(s, dtype=torch.int64) for s in seqs], att(), device=DEV)
    o = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in seqs], att())
    if order != "sorted":
        getattr(t, order + "_permute")(); getattr(o, order + "_permute")()
    wo = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    loss_o = mo.StackEngineOracle(cfg, wo, max(map(len, seqs))).backward(o, mo.default_loss, block)
    m = Qwen3TreeLM.from_named(cfg, w, DEV, torch.float32)
    e = TreeTrainingEngine(m.config, DEV, torch.float32, max(map(len, seqs))); e.mode = mode
    if mode == "stack":
        e._stack_block_rows = lambda *a: block
    loss = e.backward(m, t, mo.default_loss, block)
    assert abs(loss - float(loss_o)) <= 1e-5 * max(abs(float(loss_o)), 1.0), (loss, float(loss_o))
    gmax = max(float(v.grad.norm()) for v in wo.values() if v.grad is not None)
    for n, p in m.named_parameters():
        go = wo[n].grad if wo[n].grad is not None else torch.zeros_like(wo[n])
        gp = p.grad.cpu() if p.grad is not None else torch.zeros_like(go)
        # relative to the parameter's own gradient, with a floor relative to the largest gradient of the model (a 2-token sequence leaves
        # the q projection a gradient of 1e-9: its rounding noise is not an error)
        assert float((gp - go).norm()) <= 1e-4 * float(go.norm()) + 1e-6 * gmax, (n, mode, block, order, float((gp - go).norm()), float(go.norm()), gmax)
