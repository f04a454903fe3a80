"""Host logic of the product (no GPU): TokenTrie grouping/permutation, CompressedTrie, stats, fork
positions, balancers, packing plan and the engine's loss assembly — against the reference-generated
golden fixtures.  The three device steps are replaced by tests/hostmirror.py stand-ins."""
import os
import types

import numpy as np
import pytest
import torch

import cases

import hostmirror
from dynamictreeattn_amd import data_parallel as dp
from dynamictreeattn_amd import packing, synth
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_time_model import TreeTimeModel
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine, _get_forkpos
from dynamictreeattn_amd.trie import CompressedTrie, _get_stats, _get_subtrie
from oracle import model_oracle as mo

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CPU = torch.device("cpu")


@pytest.fixture(autouse=True)
def _hooks(monkeypatch):
    hostmirror.install(monkeypatch)


def _attach_table(t):
    return [[[a["_sequence_batch_id"], n] for a, n in al] for al in t.attach_lists]


def test_trie_surface_bit_exact(trie_golden):
    for item in trie_golden["cases"]:
        seqs = synth.as_tensors(synth.make_case(item["case"])); ref = item["ref"]
        t = TokenTrie(seqs, device=CPU)
        assert t.lens == ref["lens"] and t.lcp_lens == ref["lcp_lens"] and _attach_table(t) == ref["attach"], item["case"]
        assert (t.n_sequences, t.n_tokens) == (ref["n_sequences"], ref["n_tokens"])
        assert all(t.inputs[i] is seqs[a[-1][0]] for i, a in enumerate(ref["attach"]))      # holds references, not copies
        ct = CompressedTrie(t.lens, t.lcp_lens)
        assert [[n.depth, n.seq_id, n.child_ids] for n in ct.nodes] == ref["nodes"]
        assert [list(x) for x in CompressedTrie(t.lens, t.lcp_lens).get_order_forward()] == ref["order_forward"]
        ob = CompressedTrie(t.lens, t.lcp_lens).get_order_backward()
        assert [list(x) for x in ob] == ref["order_backward"]
        assert CompressedTrie(t.lens, t.lcp_lens).get_order_random(seed=7) == ref["order_random_seed7"]
        for sub, lens, lcps in ref["sub_lens"]:
            got = CompressedTrie(t.lens, t.lcp_lens).get_lens(set(sub))
            assert [list(got[0]), list(got[1])] == [lens, lcps]
            got = CompressedTrie(t.lens, t.lcp_lens).get_lens(list(sub))                    # a list works too (data_parallel.py:52)
            assert [list(got[0]), list(got[1])] == [lens, lcps]
            st = _get_subtrie(CompressedTrie(t.lens, t.lcp_lens), set(sub))
            assert len(st.nodes) >= len(lens)
        assert t.get_stats("forward") == {**ref["stats"]["forward"], "n_sequences": t.n_sequences, "n_tokens": t.n_tokens}
        for bs in (None, 7, 2048):
            for tag, (ln, lc) in (("sorted", (t.lens, t.lcp_lens)), ("bwd", (ob[1], ob[2]))):
                exp = ref["stats"][f"backward_{tag}_{bs}"]
                if exp is not None:
                    assert _get_stats(ln, lc, "backward", bs) == exp
                else:
                    assert _get_stats(ln, lc, "backward", bs)["n_f1_tokens"] >= 0           # reference raises IndexError here
            assert _get_forkpos(ob[1], ob[2], bs) == ref["forkpos"][str(bs)]
        for name in ("forward_permute", "backward_permute"):
            t2 = TokenTrie(synth.as_tensors(synth.make_case(item["case"])), device=CPU); getattr(t2, name)()
            assert t2.lens == ref[name]["lens"] and t2.lcp_lens == ref[name]["lcp_lens"] and _attach_table(t2) == ref[name]["attach"]


def test_error_conventions_match_reference():
    with pytest.raises(ValueError, match="Input_ids not sorted in lexicographic order."):
        TokenTrie(synth.as_tensors([[5, 1], [3, 2]]), sorted=True, device=CPU)
    with pytest.raises(AssertionError, match="Length of inputs and attachs must match."):
        TokenTrie(synth.as_tensors([[1], [2]]), attachs=[{}], device=CPU)
    with pytest.raises(ValueError):
        CompressedTrie([3, 2], [1, 1])
    with pytest.raises(ValueError, match="Unsupported mode"):
        _get_stats([3], [], "sideways")
    att = [{"x": 1}, {"x": 2}]
    TokenTrie(synth.as_tensors([[2], [1]]), att, device=CPU)
    assert [a["_sequence_batch_id"] for a in att] == [0, 1]            # caller's dicts are tagged in place


def test_balancers_bit_exact(balancer_golden):
    for item in balancer_golden["cases"]:
        seqs = synth.as_tensors(synth.make_case(item["case"]))
        tt = TokenTrie(seqs, device=CPU)
        for key, exp in item["bins"].items():
            K, mode, bs, kind = key.split("_")
            K = int(K[1:]); bs = None if bs == "None" else int(bs)
            tm = TreeTimeModel()
            if kind == "tm":
                tm.coeffs = np.array([3.0e-3, 1.0e-5, 4.0e-6, 2.0e-7, 1.5e-9])
            assert dp.LB_by_n_tokens(seqs, K) == exp["LB_by_n_tokens"]
            if exp["LB_by_TM"] is not None:
                lb = dp._leaf_bins_by_TM(tt.lens, tt.lcp_lens, tm, K, mode, bs)
                assert dp.get_original_bins(tt, lb) == exp["LB_by_TM"], (item["case"], key)
            if exp["LB_by_DFS_and_TM"] is not None:
                lb = dp._leaf_bins_by_DFS_and_TM(tt.lens, tt.lcp_lens, tm, K, mode, bs)
                assert dp.get_original_bins(tt, lb) == exp["LB_by_DFS_and_TM"], (item["case"], key)


def test_random_orders_under_a_seeded_global_rng():
    """`--permute random` (run.py:93-94 -> token_trie.py:106-109) uses `seed=None`: the global RNG is consumed in the reference's node
    order, so a seeded run reproduces the reference's leaf order bit for bit (tests/golden/random_order_cases.json)."""
    import json, random
    gold = json.load(open(os.path.join(GOLD, "random_order_cases.json")))
    for item in gold["cases"]:
        seqs = synth.as_tensors(synth.make_case(item["case"]))
        t = TokenTrie(seqs, device=CPU)
        for k, exp in item["orders"].items():
            random.seed(int(k))
            assert CompressedTrie(t.lens, t.lcp_lens).get_order_random() == exp, (item["case"], k)
            t2 = TokenTrie(seqs, device=CPU); random.seed(int(k)); t2.random_permute()
            assert t2.lens == item["permuted"][k]["lens"] and t2.lcp_lens == item["permuted"][k]["lcp_lens"]
    p = gold["probe"]
    for k, exp in p["orders"].items():
        random.seed(int(k))
        assert CompressedTrie(p["lens"], p["lcp"]).get_order_random() == exp, k


def test_public_balancer_entries_bit_exact(balancer_golden, monkeypatch):
    """The PUBLIC entries LB_by_n_tokens / LB_by_TM / LB_by_DFS_and_TM(seqs, tm, args) (data_parallel.py:8-16, 39-56, 81-107) against
    the reference's recorded bins.  They build their own TokenTrie; here its device steps are the CPU stand-ins (the same entries
    run on the real HIP TokenTrie in tests/test_gpu_dp.py)."""
    from dynamictreeattn_amd import token_trie
    orig = token_trie.TokenTrie.__init__
    monkeypatch.setattr(token_trie.TokenTrie, "__init__", lambda self, *a, **kw: orig(self, *a, **{**kw, "device": CPU}))
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
            assert dp.LB_by_n_tokens(seqs, K) == exp["LB_by_n_tokens"]
            if exp["LB_by_TM"] is not None:
                assert dp.LB_by_TM(seqs, tm, args) == exp["LB_by_TM"], (item["case"], key); n += 1
            if exp["LB_by_DFS_and_TM"] is not None:
                assert dp.LB_by_DFS_and_TM(seqs, tm, args) == exp["LB_by_DFS_and_TM"], (item["case"], key); n += 1
    assert n > 20


def test_packing_plan_matches_bruteforce():
    from oracle import trie_oracle as to
    for case in cases.trie_cases():
        seqs = synth.make_case(case)
        if sum(map(len, seqs)) > 3000:
            continue
        t = TokenTrie(synth.as_tensors(seqs), device=CPU); t.backward_permute()
        plan = packing.plan_segments(t.lens, t.lcp_lens)
        _, depth, parent, se = hostmirror.expand_plan_host(plan)
        # brute force: simulate the stack walk token by token
        stack, par, dep = [], [], []
        prev = None
        for leaf in t.inputs:
            leaf = leaf.tolist()
            c = 0 if prev is None else to.lcp_pair(np.array(prev), np.array(leaf))
            stack = stack[:c]
            for d in range(c, len(leaf)):
                par.append(stack[-1] if stack else -1); dep.append(d); stack.append(len(par) - 1)
            prev = leaf
        T = len(par)
        anc = []
        for x in range(T):
            s, p = set(), x
            while p != -1:
                s.add(p); p = par[p]
            anc.append(s)
        se_b = [max(y for y in range(T) if s in anc[y]) + 1 for s in range(T)]
        assert depth.tolist() == dep and parent.tolist() == par and se.tolist() == se_b, case
        rp, runs = packing.plan_qtile_runs(plan, 16)
        for qt in range(len(rp) - 1):
            rows = range(qt * 16, min(T, qt * 16 + 16))
            need = set().union(*(anc[x] for x in rows))
            cov = set()
            for b, e, f, _ in runs[rp[qt]:rp[qt + 1]]:
                cov |= set(range(b, e))
                if f == 0:
                    assert all(set(range(b, e)) <= anc[x] for x in rows)
            assert need <= cov


def test_padded_plan_keeps_real_tokens_and_isolates_filler():
    for case in cases.trie_cases()[:12]:
        seqs = synth.make_case(case)
        if sum(map(len, seqs)) > 3000:
            continue
        t = TokenTrie(synth.as_tensors(seqs), device=CPU); t.backward_permute()
        plan = packing.plan_segments(t.lens, t.lcp_lens)
        for mult in (16, 64):
            pp = packing.pad_plan(plan, mult)
            assert pp.T % mult == 0 and pp.T - plan.T < mult and pp.n_real == plan.M
            _, d0, p0, s0 = hostmirror.expand_plan_host(plan)
            _, d1, p1, s1 = hostmirror.expand_plan_host(pp)
            T = plan.T
            assert d1[:T].tolist() == d0.tolist() and p1[:T].tolist() == p0.tolist() and s1[:T].tolist() == s0.tolist()
            assert int(s0.max(initial=0)) <= T                                   # no real subtree reaches the filler
            n = pp.T - T
            assert d1[T:].tolist() == list(range(n)) and s1[T:].tolist() == [pp.T] * n
            assert p1[T:].tolist() == ([-1] + list(range(T, pp.T - 1)) if n else [])
            rp, runs = packing.plan_qtile_runs(pp, 16)
            for qt in range(len(rp) - 1):                                       # filler rows only ever meet filler keys
                for b, e, f, _ in runs[rp[qt]:rp[qt + 1]]:
                    if qt * 16 >= T:
                        assert b >= T
                    if e > T and b < T:
                        assert f == 1


def test_engine_padding_changes_nothing(monkeypatch):
    from dynamictreeattn_amd import tree_training_engine as tte
    m, seqs = _model("d128_minitau")
    outs = []
    for pad_from in (1 << 30, 1):
        monkeypatch.setattr(tte._PackedTrie, "PAD_FROM", pad_from)
        monkeypatch.setattr(tte._PackedTrie, "PAD_TO", 64)
        m.zero_grad(set_to_none=True)
        t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
        eng = tte.TreeTrainingEngine(m.config, CPU, torch.float32, 4096)
        loss = eng.backward(m, t, mo.default_loss, 2048)
        assert (eng.last_packed.plan.T % 64 == 0) == (pad_from == 1) or eng.last_packed.n_real_tokens % 64 == 0
        outs.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0])
    for k in outs[0][1]:
        assert mo.grad_ratio(outs[1][1][k], outs[0][1][k]) < 1e-5, k


def test_layer_recomputation_gives_the_same_gradients():
    m, seqs = _model("d128_tree")
    outs = []
    for ck in (False, True):
        m.zero_grad(set_to_none=True)
        t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
        eng = TreeTrainingEngine(m.config, CPU, torch.float32, 4096)
        eng.checkpoint_layers = ck
        loss = eng.backward(m, t, mo.default_loss, 2048)
        outs.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * abs(outs[0][0])
    for k in outs[0][1]:
        assert mo.grad_ratio(outs[1][1][k], outs[0][1][k]) < 1e-6, k


@pytest.fixture(scope="module")
def eng_gold():
    return torch.load(os.path.join(GOLD, "engine_tiny.pt"), weights_only=True)


def _att(n):
    return [{"w_logprobs": -1.0 - 0.01 * i, "w_entropy": 0.1 + 0.003 * i} for i in range(n)]


def _model(name):
    from dynamictreeattn_amd.model import Qwen3TreeLM
    case = cases.engine_cases()[name]
    cfg = cases.TINY_CFGS[case["cfg"]]
    m = Qwen3TreeLM(cfg).load_named(mo.init_weights(cfg, seed=case["wseed"]))
    return m, synth.as_tensors(synth.make_case(case["data"]))


@pytest.mark.parametrize("name", list(cases.engine_cases()))
def test_engine_forward_fp32_vs_reference(name, eng_gold):
    m, seqs = _model(name); g = eng_gold[name]
    for perm in ("idx", "forward"):
        t = TokenTrie(seqs, device=CPU)
        if perm == "forward":
            t.forward_permute()
        out = TreeTrainingEngine(m.config, CPU, torch.float32, max(map(len, seqs)), forward_only=True).forward(m, t)
        for a, b in zip(out, g[f"fwd_{perm}"]):
            assert a.dtype == torch.float32 and torch.allclose(a, b, atol=3e-5, rtol=0)


@pytest.mark.parametrize("name", list(cases.engine_cases()))
@pytest.mark.parametrize("perm", ["ours", "idx", "random"])
def test_engine_backward_fp32_vs_reference(name, perm, eng_gold):
    m, seqs = _model(name); g = eng_gold[name]
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU)
    if perm == "ours":
        t.backward_permute()
    elif perm == "random":
        t.random_permute()
    eng = TreeTrainingEngine(m.config, CPU, torch.float32, max(map(len, seqs)))
    loss = eng.backward(m, t, mo.default_loss, block_size=2048)
    assert abs(loss - g["bwd_bs2048_loss"]) < 2e-4 * max(1.0, abs(loss))
    for n, p in m.named_parameters():
        assert mo.grad_ratio(g["bwd_bs2048_grads"][n], p.grad) <= 2e-5, (name, n)
    with pytest.raises(AssertionError, match="Exceeds max_seq_len"):
        TreeTrainingEngine(m.config, CPU, torch.float32, 3).backward(m, t, mo.default_loss, 2048)


@pytest.mark.parametrize("name", list(cases.engine_cases()))
@pytest.mark.parametrize("bs,perm", [(7, "ours"), (64, "ours"), (2048, "ours"), (7, "idx"), (33, "random")])
def test_blockwise_stack_engine_fp32_vs_reference(name, bs, perm, eng_gold):
    """engine.mode = "stack": the memory-bounded push/pop walk over the KV stack (dynamictreeattn_amd/stack_engine.py — the
    reference's own schedule, tte:232-616, block_size = rows per popped block) reproduces the reference's recorded loss and
    every parameter gradient for any block size and leaf order: the side channels (grad-KV stack, g_lp, g_ent, pending picks
    that replace the fork-position logits) are exact."""
    m, seqs = _model(name); g = eng_gold[name]
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU)
    if perm == "ours":
        t.backward_permute()
    elif perm == "random":
        t.random_permute()
    eng = TreeTrainingEngine(m.config, CPU, torch.float32, max(map(len, seqs)))
    eng.mode = "stack"
    loss = eng.backward(m, t, mo.default_loss, block_size=bs)
    assert eng.last_mode.startswith("stack[")
    assert abs(loss - g["bwd_bs2048_loss"]) < 2e-4 * max(1.0, abs(loss))
    for n, p in m.named_parameters():
        assert mo.grad_ratio(g["bwd_bs2048_grads"][n], p.grad) <= 3e-5, (name, n)
    with pytest.raises(AssertionError, match="Exceeds max_seq_len"):
        e2 = TreeTrainingEngine(m.config, CPU, torch.float32, 3); e2.mode = "stack"
        e2.backward(m, t, mo.default_loss, 2048)


def test_engine_mode_auto_switches_by_footprint():
    """"auto": one packed pass while the predicted footprint fits the budget, the block-wise walk otherwise (the budget is the
    free HBM on the GPU; here it is set by hand)."""
    m, seqs = _model("d16_tree")
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
    eng = TreeTrainingEngine(m.config, CPU, torch.float32, 4096)
    eng.memory_budget_bytes = 1 << 40
    l0 = eng.backward(m, t, mo.default_loss, 2048); assert eng.last_mode.startswith("packed")
    g0 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    eng.memory_budget_bytes = 1 << 10
    l1 = eng.backward(m, t, mo.default_loss, 16); assert eng.last_mode.startswith("stack[")
    assert abs(l0 - l1) < 1e-4 * abs(l0)
    for n, p in m.named_parameters():
        assert mo.grad_ratio(g0[n], p.grad) <= 3e-5, n


def test_engine_accepts_a_huggingface_module_by_duck_typing(eng_gold):
    """the engine drives an HF Qwen3ForCausalLM through its own nn.Parameters."""
    transformers = pytest.importorskip("transformers")
    name = "d128_minitau"
    case = cases.engine_cases()[name]; cfg = cases.TINY_CFGS[case["cfg"]]
    c = transformers.Qwen3Config(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                                 num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                                 num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"], tie_word_embeddings=True,
                                 rms_norm_eps=cfg["rms_norm_eps"], rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"]})
    hf = transformers.Qwen3ForCausalLM(c)
    w = mo.init_weights(cfg, seed=case["wseed"])
    hf.load_state_dict({**w, "lm_head.weight": w["model.embed_tokens.weight"]}, strict=False)
    hf = hf.float().train()
    seqs = synth.as_tensors(synth.make_case(case["data"]))
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
    loss = TreeTrainingEngine(hf.config, CPU, torch.float32, 64).backward(hf, t, mo.default_loss, 2048)
    g = eng_gold[name]
    assert abs(loss - g["bwd_bs2048_loss"]) < 2e-4 * abs(loss)
    for n, p in hf.named_parameters():
        if n in g["bwd_bs2048_grads"]:
            assert mo.grad_ratio(g["bwd_bs2048_grads"][n], p.grad) <= 2e-5, n


def test_engine_accepts_a_huggingface_qwen2_module_with_biases():
    """Qwen2ForCausalLM (q/k/v biases, no per-head q/k norm) through the engine vs the REFERENCE engine on the same class
    (tests/golden/engine_qwen2_tiny.pt): loss and all 26 gradients, fp32 on the CPU stand-ins."""
    pytest.importorskip("transformers")
    hf = cases.hf_qwen2()
    g = torch.load(os.path.join(GOLD, "engine_qwen2_tiny.pt"), weights_only=True)
    seqs = synth.as_tensors(synth.make_case(cases.QWEN2_DATA))
    t = TokenTrie(seqs, device=CPU); t.forward_permute()
    out = TreeTrainingEngine(hf.config, CPU, torch.float32, 64, forward_only=True).forward(hf, t)
    for a, b in zip(out, g["fwd_forward"]):
        assert torch.allclose(a, b, atol=2e-5)
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
    loss = TreeTrainingEngine(hf.config, CPU, torch.float32, 64).backward(hf, t, mo.default_loss, 2048)
    assert abs(loss - g["bwd_bs2048_loss"]) < 2e-5 * abs(loss)
    named = dict(hf.named_parameters())
    for n, (q, s_) in g["bwd_bs2048_grads_fp16_scaled"].items():
        assert mo.grad_ratio(q.float() * s_, named[n].grad) <= 1e-3, n                        # fp16-packed golden: 5e-4 per element
        assert abs(float(named[n].grad.norm()) - g["grad_norms"][n]) <= 2e-5 * g["grad_norms"][n] + 1e-9, n


def test_dkv_work_units_partition_every_key_tile():
    """packing.plan_dkv_units: the units of a key tile tile its query range exactly, in 64-row multiples,
    split tiles get consecutive slabs listed in `splits`, heaviest units first."""
    rng = np.random.default_rng(0)
    for trial in range(20):
        nkt = int(rng.integers(1, 60)); Tk = nkt * packing.KTILE - int(rng.integers(0, packing.KTILE))
        Tq = Tk
        kq = np.minimum(np.arange(nkt) * packing.KTILE + rng.integers(1, 40000, nkt), Tq).astype(np.int32)
        kq = np.maximum(kq, np.minimum(np.arange(nkt) * packing.KTILE + 1, Tq)).astype(np.int32)
        units, splits, n_slabs = packing.plan_dkv_units(kq, Tk, Tq, 0, Hkv=int(rng.integers(1, 9)))
        assert units.dtype == np.int32 and units.shape[1] == 4
        sizes = units[:, 2] - units[:, 1]
        assert (np.diff(sizes) <= 0).all()                                  # heaviest first
        slabs_seen = []
        for kt in range(nkt):
            u = units[units[:, 0] == kt]
            u = u[np.argsort(u[:, 1])]
            assert u[0, 1] == kt * packing.KTILE and u[-1, 2] == min(int(kq[kt]), Tq)
            assert (u[1:, 1] == u[:-1, 2]).all() and ((u[:-1, 2] - u[:-1, 1]) % 64 == 0).all()
            if len(u) > 1:
                assert (np.diff(u[:, 3]) == 1).all()
                row = splits[splits[:, 0] == kt][0]
                assert row[1] == u[0, 3] and row[2] == len(u)
                slabs_seen += u[:, 3].tolist()
            else:
                assert u[0, 3] == -1
        assert sorted(slabs_seen) == list(range(n_slabs))


def test_grad_reducer_start_after_a_foreign_zero_grad_does_not_keep_stale_sums():
    """`optimizer.zero_grad()` (set_to_none) -> `red.start()` -> backward: the flat buffers still hold the previous step's reduced
    sums when start() re-attaches the views; they must be zeroed, or step 2 accumulates onto step 1 (2.0 where 1.0 is expected)."""
    from dynamictreeattn_amd import dp as dpr
    m = torch.nn.Linear(4, 3)
    red = dpr.GradReducer(m.parameters(), overlap=False)
    x = torch.ones(2, 4)
    red.zero_grad(); red.start(); m(x).sum().backward()
    g1 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    red.start(); m(x).sum().backward()
    views = {id(p): v for p, v in red._views}
    for p, g in zip(m.parameters(), g1):
        assert p.grad.data_ptr() == views[id(p)].data_ptr() and torch.equal(p.grad, g)
    dpr._ACTIVE.discard(red)


def test_loss_fn_may_return_a_plain_number_for_some_sequences(monkeypatch):
    """The reference sums `loss += loss_fn(...)` from 0.0 (tte:396-398), so a callback returning a Python number for some sequence
    works there; it must here too (packed and block-wise)."""
    case = cases.engine_cases()["d16_tree"]; cfg = cases.TINY_CFGS[case["cfg"]]
    from dynamictreeattn_amd.model import Qwen3TreeLM
    seqs = synth.as_tensors(synth.make_case(case["data"]))
    fn = lambda lp, ent, att: (0.25 if att["_sequence_batch_id"] % 2 else mo.default_loss(lp, ent, att))
    ref = None
    for mode in ("packed", "stack"):
        model = Qwen3TreeLM(cfg).load_named(mo.init_weights(cfg, seed=case["wseed"]))
        t = TokenTrie(seqs, [{"w_logprobs": -1.0, "w_entropy": 0.1} for _ in seqs], device=CPU); t.backward_permute()
        e = TreeTrainingEngine(model.config, "cpu", torch.float32, 4096); e.mode = mode
        loss = e.backward(model, t, fn, 16)
        n_odd = sum(1 for i in range(len(seqs)) if i % 2)
        full = TreeTrainingEngine(model.config, "cpu", torch.float32, 4096)
        if ref is None:
            ref = loss
        assert abs(loss - ref) < 1e-4 * abs(ref) and n_odd > 0


def test_blockwise_fp32_gradient_sink_on_16bit_parameters():
    """stack_engine._GradSink (bf16 parameters: flat model-dtype buffer absorbed into a flat fp32 buffer per block, one rounding at
    the end): same gradients as the per-block model-dtype accumulation up to bf16 rounding, closer to the fp32 run, added to what
    the caller already held in `param.grad`, and `param.grad` keeps the model dtype."""
    from dynamictreeattn_amd.model import Qwen3TreeLM
    case = cases.engine_cases()["d16_tree"]; cfg = cases.TINY_CFGS[case["cfg"]]
    seqs = synth.as_tensors(synth.make_case(case["data"]))
    w = mo.init_weights(cfg, seed=case["wseed"])

    def run(dtype, sink, twice=False):
        m = Qwen3TreeLM(cfg).load_named(w).to(dtype)
        for _ in range(2 if twice else 1):
            t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
            e = TreeTrainingEngine(m.config, "cpu", dtype, 4096); e.mode, e.stack_fp32_grads = "stack", sink
            e.backward(m, t, mo.default_loss, 8)
            assert e.last_mode.startswith("stack[8]")
        assert all(p.grad.dtype == dtype for p in m.parameters())
        return {n: p.grad.float() for n, p in m.named_parameters()}
    ref = run(torch.float32, False)
    on, off, on2 = run(torch.bfloat16, True), run(torch.bfloat16, False), run(torch.bfloat16, True, twice=True)
    r_on = np.mean([mo.grad_ratio(ref[n], on[n]) for n in ref]); r_off = np.mean([mo.grad_ratio(ref[n], off[n]) for n in ref])
    assert r_on <= r_off * 1.02 and r_on < 0.05, (r_on, r_off)
    assert max(mo.grad_ratio(2 * on[n], on2[n]) for n in ref) < 2e-2          # a second call ADDS to the caller's gradient


def test_blockwise_walk_hands_the_gradients_back_when_the_loss_callback_raises():
    """stack_engine._GradSink.abort: for the walk every param.grad is a view of the sink's buffer; if the user's loss_fn raises half-way,
    the caller's own gradient tensors come back untouched (not dangling views, no partial sums)."""
    from dynamictreeattn_amd.model import Qwen3TreeLM
    case = cases.engine_cases()["d16_tree"]; cfg = cases.TINY_CFGS[case["cfg"]]
    seqs = synth.as_tensors(synth.make_case(case["data"]))
    m = Qwen3TreeLM(cfg).load_named(mo.init_weights(cfg, seed=case["wseed"])).to(torch.bfloat16)
    for p in m.parameters():
        p.grad = torch.full_like(p, 0.5)
    before = {n: p.grad for n, p in m.named_parameters()}
    calls = []

    def bad_loss(lp, ent, att):
        calls.append(1)
        if len(calls) == 3:
            raise RuntimeError("user callback failed")
        return mo.default_loss(lp, ent, att)
    t = TokenTrie(seqs, _att(len(seqs)), device=CPU); t.backward_permute()
    e = TreeTrainingEngine(m.config, "cpu", torch.bfloat16, 4096); e.mode = "stack"
    with pytest.raises(RuntimeError, match="user callback failed"):
        e.backward(m, t, bad_loss, 8)
    for n, p in m.named_parameters():
        assert p.grad is before[n] and bool((p.grad == 0.5).all()), n
