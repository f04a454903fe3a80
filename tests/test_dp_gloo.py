"""N>1 path on CPU: two gloo ranks on loopback.  (1) trie-sharded data parallelism: every rank
partitions the same global batch with LB_by_DFS_and_TM, runs its bin, gradients are summed with
dp.allreduce_grads / dp.GradReducer and must equal the single-process full-batch gradients recorded from the
reference (tests/golden/engine_tiny.pt).  (2) the vocab-sharded logprob/entropy branch against the
unsharded one and the reference's recorded values."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _dp_worker(rank, world, port, name, q, overlapped=False, one_leaf=False, stack_block=0):
    import sys
    sys.path.insert(0, os.path.dirname(__file__)); sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import cases
    import hostmirror
    from dynamictreeattn_amd import dp, ops, synth, token_trie, tree_training_engine
    from dynamictreeattn_amd.model import Qwen3TreeLM
    from oracle import model_oracle as mo
    _init(rank, world, port)
    # CPU stand-ins for the device steps (no pytest monkeypatch in a spawned process)
    token_trie._device_trie_arrays = hostmirror._cpu_trie_arrays
    tree_training_engine._PackedTrie._expand = hostmirror._cpu_expand
    for k, fn in {"tree_attention": "_cpu_attention", "lm_head_rows": "_cpu_lm_head_rows", "rms_norm": "_cpu_rms_norm", "add_rms_norm": "_cpu_add_rms_norm",
                  "qk_norm_rope": "_cpu_qk_norm_rope", "qkv_prep": "_cpu_qkv_prep", "swiglu": "_cpu_swiglu", "swiglu_fused": "_cpu_swiglu_fused"}.items():
        setattr(ops, k, getattr(hostmirror, fn))
    _orig = token_trie.TokenTrie.__init__
    token_trie.TokenTrie.__init__ = lambda self, *a, **kw: _orig(self, *a, **{**kw, "device": torch.device("cpu")})
    case = cases.engine_cases()[name]; cfg = cases.TINY_CFGS[case["cfg"]]
    model = Qwen3TreeLM(cfg).load_named(mo.init_weights(cfg, seed=case["wseed"]))
    seqs = synth.as_tensors(synth.make_case(case["data"]))
    att = [{"w_logprobs": -1.0 - 0.01 * i, "w_entropy": 0.1 + 0.003 * i} for i in range(len(seqs))]
    if one_leaf:                      # fewer leaves than ranks: the balancer leaves rank 1's bin EMPTY (it must still reduce)
        seqs, att = seqs[:1], att[:1]
    ids = dp.my_bin(seqs, rank, world, "backward", 2048)
    t = token_trie.TokenTrie([seqs[i] for i in ids], [att[i] for i in ids])
    if ids:
        t.backward_permute()
    eng = tree_training_engine.TreeTrainingEngine(model.config, "cpu", torch.float32, 4096)
    ops.stack_attention = hostmirror._cpu_stack_attention
    block = 2048
    if stack_block:                   # block-wise engine: every parameter accumulates once per block, the reducer must wait for the last one
        eng.mode, block = "stack", stack_block
    if overlapped:
        # tiny buckets: several collectives are in flight while the backward is still running; a second step re-uses the hooks;
        # every param.grad is a view of a persistent flat buffer that is reduced in place
        red = dp.GradReducer(model.parameters(), bucket_bytes=20000)
        assert len(red.buckets) > 2
        for _ in range(2):
            red.zero_grad()
            red.start(); loss = eng.backward(model, t, mo.default_loss, block); red.finish()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in red._views)
        red.close()
    else:
        loss = eng.backward(model, t, mo.default_loss, block)
        dp.allreduce_grads(model.parameters())
    lt = torch.tensor([loss], dtype=torch.float64); dist.all_reduce(lt)
    torch.save({"loss": float(lt), "ids": sorted(ids), "grads": {n: p.grad.clone() for n, p in model.named_parameters()} if rank == 0 else None},
               os.path.join(q, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,overlapped,world", [("d128_minitau", False, 2), ("d128_tree", False, 2), ("d128_minitau", True, 2), ("d128_tree", True, 2),
                                                   ("d128_tree", True, 3), ("d128_tree", False, 4)])
def test_trie_sharded_dp_grads_equal_full_batch(name, overlapped, world):
    """world 3 and 4: more ranks than the 2-rank ring; bins may be empty (d128_tree has 4 leaves) and every rank still reduces."""
    from oracle import model_oracle as mo
    gold = torch.load(os.path.join(GOLD, "engine_tiny.pt"), weights_only=True)[name]
    import tempfile
    ctx = mp.get_context("spawn"); q = tempfile.mkdtemp(); port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, name, q, overlapped)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res = [torch.load(os.path.join(q, f"rank{r}.pt"), weights_only=True) for r in range(world)]
    loss, grads = res[0]["loss"], res[0]["grads"]
    ids = sorted(sum((r["ids"] for r in res), []))
    assert ids == list(range(len(ids)))                                         # a partition of the batch
    if world == 2:
        assert res[0]["ids"] and res[1]["ids"]                                  # both bins non-empty
    assert abs(loss - gold["bwd_bs2048_loss"]) < 2e-4 * abs(loss)
    for n, g in gold["bwd_bs2048_grads"].items():
        assert mo.grad_ratio(g, grads[n]) <= 3e-5, n


def test_overlapped_reduce_waits_for_the_last_block_of_the_blockwise_engine():
    """engine.mode = "stack" accumulates into every parameter once per popped block; the hook-driven reducer is deferred by the engine
    and reduces in finish() — otherwise a bucket would go out after its parameters' FIRST accumulation."""
    from oracle import model_oracle as mo
    import tempfile
    name = "d128_tree"
    gold = torch.load(os.path.join(GOLD, "engine_tiny.pt"), weights_only=True)[name]
    ctx = mp.get_context("spawn"); q = tempfile.mkdtemp(); port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, name, q, True, False, 16)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res = torch.load(os.path.join(q, "rank0.pt"), weights_only=True)
    assert abs(res["loss"] - gold["bwd_bs2048_loss"]) < 2e-4 * abs(res["loss"])
    for n, g in gold["bwd_bs2048_grads"].items():
        assert mo.grad_ratio(g, res["grads"][n]) <= 4e-5, n


@pytest.mark.parametrize("overlapped", [False, True])
def test_dp_with_an_empty_bin_still_reduces(overlapped):
    """K = 2 ranks, ONE sequence: rank 1's bin is empty.  It runs no backward but issues the same collectives; the reduced
    gradients on rank 0 equal the single-process gradients of that sequence."""
    import sys, tempfile
    sys.path.insert(0, os.path.dirname(__file__))
    import cases
    import hostmirror
    from oracle import model_oracle as mo
    ctx = mp.get_context("spawn"); q = tempfile.mkdtemp(); port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, "d128_minitau", q, overlapped, True)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res = [torch.load(os.path.join(q, f"rank{r}.pt"), weights_only=True) for r in range(2)]
    assert res[0]["ids"] == [0] and res[1]["ids"] == []
    # reference value: the oracle's dense pass over that one sequence (fp32)
    from dynamictreeattn_amd import synth
    case = cases.engine_cases()["d128_minitau"]; cfg = cases.TINY_CFGS[case["cfg"]]
    w = {k: v.clone().requires_grad_(True) for k, v in mo.init_weights(cfg, seed=case["wseed"]).items()}
    seq = synth.make_case(case["data"])[0]
    loss = mo.dense_backward(cfg, w, [seq], [{"w_logprobs": -1.0, "w_entropy": 0.1}], mo.default_loss)
    assert abs(res[0]["loss"] - float(loss)) < 2e-4 * abs(float(loss))
    for n, g in res[0]["grads"].items():
        assert mo.grad_ratio(w[n].grad, g) <= 3e-5, n


def _vp_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    sys.path.insert(0, os.path.dirname(__file__))
    from dynamictreeattn_amd import vocab_parallel as vp
    import hostmirror
    hostmirror.install_cpu_logprob()              # the product has no CPU path: host logits raise there
    _init(rank, world, port)
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    out = {}
    for name in ("small", "temp"):
        c = gold[name]
        if c["V"] % world:
            continue
        g = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["R"], c["V"], generator=g) * 3
        labels = torch.randint(0, c["V"], (c["R"],), generator=g)
        go_lp = torch.randn(c["R"], generator=g); go_ent = torch.randn(c["R"], generator=g)
        Vp = c["V"] // world
        shard = logits[:, rank * Vp:(rank + 1) * Vp].clone().requires_grad_(True)
        lp, ent = vp.gather_logprobs_entropy(shard, labels, temperature=c["temp"], tp_group=dist.group.WORLD, chunk_size=16)
        lp_only = vp.gather_logprobs(shard.detach(), labels, temperature=c["temp"], tp_group=dist.group.WORLD, chunk_size=7)
        ((lp * go_lp).sum() + (ent * go_ent).sum()).backward()
        full = logits.clone().requires_grad_(True)
        lp_f, ent_f = vp.gather_logprobs_entropy(full, labels, temperature=c["temp"])
        ((lp_f * go_lp).sum() + (ent_f * go_ent).sum()).backward()
        out[name] = dict(lp=lp.detach(), ent=ent.detach(), lp_only=lp_only, gshard=shard.grad, gfull=full.grad[:, rank * Vp:(rank + 1) * Vp],
                         lp_f=lp_f.detach(), ent_f=ent_f.detach())
    torch.save(out, os.path.join(q, f"vp{rank}.pt"))
    dist.destroy_process_group()


def test_vocab_parallel_matches_unsharded_and_reference():
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    import tempfile
    ctx = mp.get_context("spawn"); q = tempfile.mkdtemp(); port = _free_port()
    procs = [ctx.Process(target=_vp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res = {r: torch.load(os.path.join(q, f"vp{r}.pt"), weights_only=True) for r in range(2)}
    for rank in (0, 1):
        for name, o in res[rank].items():
            c = gold[name]
            assert torch.allclose(o["lp"], c["logprobs"], atol=2e-5) and torch.allclose(o["ent"], c["entropy"], atol=2e-5)
            assert torch.allclose(o["lp_only"], c["logprobs_only"], atol=2e-5)
            assert torch.allclose(o["lp_f"], c["logprobs"], atol=2e-5) and torch.allclose(o["ent_f"], c["entropy"], atol=2e-5)
            assert torch.allclose(o["gshard"], o["gfull"], atol=2e-6, rtol=2e-4), (o["gshard"] - o["gfull"]).abs().max()
    assert set(res[0]) == {"small", "temp"}


def test_single_process_logprob_surface_and_shape_quirk(monkeypatch):
    """[1,B,V] logits with [1,B-1] labels (tte:190-193): logprobs B-1 rows, entropy B rows (the public surface's shape handling;
    the arithmetic is the host stand-in here and the HIP kernels in tests/test_gpu_logprob.py)."""
    from dynamictreeattn_amd import vocab_parallel as vp
    from oracle import model_oracle as mo
    import hostmirror
    with pytest.raises(RuntimeError, match="no CPU path"):            # the product itself refuses host logits
        vp.gather_logprobs(torch.randn(2, 8), torch.zeros(2, dtype=torch.long))
    with pytest.raises(AssertionError):                               # vocab_parallel.py:46
        vp.gather_logprobs(torch.zeros(0, 8), torch.zeros(0, dtype=torch.long))
    hostmirror.install_cpu_logprob(monkeypatch)
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(1, 9, 50, generator=g); labels = torch.randint(0, 50, (1, 8), generator=g)
    lp, ent = vp.gather_logprobs_entropy(logits, labels, chunk_size=4)
    lp_o, ent_o = mo.logprobs_entropy_of(logits[0], labels[0])
    assert lp.shape == (1, 8) and ent.shape == (1, 9)
    assert torch.allclose(lp[0], lp_o, atol=1e-6) and torch.allclose(ent[0], ent_o, atol=1e-6)
    assert torch.allclose(vp.gather_logprobs(logits, labels, temperature=0.5)[0], mo.logprobs_of(logits[0], labels[0], 0.5), atol=1e-6)
