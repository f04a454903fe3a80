"""GPU: the multi-rank paths on the device before a multi-GPU node exists (BASELINE configs 3 and 4).

* `engine.backward` on the HIP path under a LIVE `dp.GradReducer` (hook-launched asynchronous in-place all-reduce on the device
  flat buffers): a 1-rank "nccl" group — that IS RCCL on ROCm, so the backend the 8-GPU run uses is initialised and its
  collectives execute — and 2 gloo ranks that share this box's one card, the full batch split by `dp.my_bin`, an empty bin included.
  The summed gradients must equal the single-process full-batch gradients within the reference's recorded bf16 bound.
* the PUBLIC `vocab_parallel.gather_logprobs(_entropy)(cuda_logits, tp_group=…)` (reference vocab_parallel.py:399-467, sharded
  Functions :82-396) with 2 ranks against the reference's recorded values and gradients (tests/golden/logprob_cases.pt).
* `python bench.py --gpus 2` with WORLD_SIZE unset starts its own ranks (gloo rehearsal on the one card) and prints ONE JSON line.

Every worker is a spawned child (its own HIP context and process group); ranks that touch the card: at most 2.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases

from dynamictreeattn_amd import synth
from oracle import model_oracle as mo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
DEV = torch.device("cuda:0")
REF_BF16_BOUND = json.load(open(os.path.join(GOLD, "recorded_bf16_table.json")))["max"]

# a tau2-shaped bin at a size the d128 model runs in a second: 4 rollouts x 4 turns over a 300-token shared prompt, ~7k tokens
TAU2_BIN = {"kind": "tau2", "seed": 9, "V": 512, "G": 4, "sys_len": 300, "turns": 4, "lo": 40, "hi": 160, "cap": 1024}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _att(n):
    return [{"w_logprobs": -1.0 - 0.01 * i, "w_entropy": 0.1 + 0.003 * i} for i in range(n)]


def _model(dtype=torch.bfloat16):
    from dynamictreeattn_amd.model import Qwen3TreeLM
    cfg = cases.TINY_CFGS["d128"]
    return Qwen3TreeLM.from_named(cfg, mo.init_weights(cfg, seed=1), DEV, dtype)


def _dp_worker(rank, world, port, backend, outdir, one_leaf, overlap, stack=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch.distributed as dist
    from dynamictreeattn_amd import dp
    from dynamictreeattn_amd.token_trie import TokenTrie
    from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": DEV} if backend == "nccl" else {}))
    model = _model()
    seqs = synth.as_tensors(synth.make_case(TAU2_BIN))
    att = _att(len(seqs))
    if one_leaf:                       # fewer leaves than ranks: rank 1's bin is EMPTY and it must still take part in every reduce
        seqs, att = seqs[:1], att[:1]
    red = dp.GradReducer(model.parameters(), bucket_bytes=60000, overlap=overlap)
    assert len(red.buckets) > 3 and all(f.is_cuda for f in red.flats)
    eng = TreeTrainingEngine(model.config, DEV, torch.bfloat16, 1024)
    if stack:                          # block-wise walk: parameters accumulate once per block into the walk's fp32 sink, the reducer's
        eng.mode = "stack"             # hooks are deferred and everything goes out in finish()
        eng._stack_block_rows = lambda *a: 256
    loss = None
    for it in range(2):                # the second step re-uses hooks and buffers (and must not see the first step's sums)
        if it == 0:
            red.zero_grad()
        else:
            model.zero_grad(set_to_none=True)          # the default PyTorch idiom instead of red.zero_grad(): start() re-attaches AND zeroes
        ids = dp.my_bin(seqs, rank, world, "backward", 2048)
        t = TokenTrie([seqs[i] for i in ids], [att[i] for i in ids], device=DEV)
        if ids:
            t.backward_permute()
        red.start()
        loss = eng.backward(model, t, mo.default_loss, 2048)
        red.finish()
        torch.cuda.synchronize()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in red._views)
    lt = torch.tensor([loss], dtype=torch.float64, device=DEV); dist.all_reduce(lt)
    torch.save({"loss": float(lt), "ids": sorted(ids), "mode": eng.last_mode,
                "grads": {n: p.grad.float().cpu() for n, p in model.named_parameters()}}, os.path.join(outdir, f"rank{rank}.pt"))
    red.close()
    dist.destroy_process_group()


def _run(target, world, args, timeout=600):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world) + args) for r in range(world)]
    [p.start() for p in procs]
    [p.join(timeout) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _full_batch_reference(one_leaf):
    from dynamictreeattn_amd.token_trie import TokenTrie
    from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine
    model = _model()
    seqs = synth.as_tensors(synth.make_case(TAU2_BIN)); att = _att(len(seqs))
    if one_leaf:
        seqs, att = seqs[:1], att[:1]
    t = TokenTrie(seqs, att, device=DEV); t.backward_permute()
    loss = TreeTrainingEngine(model.config, DEV, torch.bfloat16, 1024).backward(model, t, mo.default_loss, 2048)
    return loss, {n: p.grad.float().cpu() for n, p in model.named_parameters()}, len(seqs)


def test_engine_backward_under_live_reducer_one_rank_rccl(tmp_path):
    """BASELINE config 3's mechanism on the device with the real backend: a 1-rank "nccl" (= RCCL) group; the hooks launch the
    in-place asynchronous all-reduces on the device flat buffers while the HIP backward is still running.  With one rank the sum
    is the identity: loss and every gradient equal the plain single-process run bit for bit."""
    _run(_dp_worker, 1, (_free_port(), "nccl", str(tmp_path), False, True))
    res = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    loss, grads, n = _full_batch_reference(False)
    assert res["ids"] == list(range(n)) and res["mode"] == "packed"
    assert res["loss"] == loss
    for name, g in grads.items():
        assert torch.equal(res["grads"][name], g), name


@pytest.mark.parametrize("one_leaf,overlap,stack", [(False, True, False), (True, True, False), (False, False, False), (False, True, True)])
def test_engine_backward_two_ranks_sharing_the_card_sum_to_full_batch(tmp_path, one_leaf, overlap, stack):
    """Two gloo ranks on the one card, the batch split by `dp.my_bin` (LB_by_DFS_and_TM): after `finish()` BOTH ranks hold the sum
    of the per-bin gradients = the full-batch gradient, within the reference's recorded bf16 bound (sharing across bins is lost, so
    the arithmetic differs); `one_leaf`: rank 1's bin is empty."""
    _run(_dp_worker, 2, (_free_port(), "gloo", str(tmp_path), one_leaf, overlap, stack))
    res = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(2)]
    assert all(r["mode"].startswith("stack[256]" if stack else "packed") for r in res if r["ids"])
    loss, grads, n = _full_batch_reference(one_leaf)
    assert sorted(res[0]["ids"] + res[1]["ids"]) == list(range(n))
    assert (res[1]["ids"] == []) == one_leaf
    assert abs(res[0]["loss"] - loss) <= 3e-3 * abs(loss)
    ratios = {}
    for name, g in grads.items():
        assert torch.equal(res[0]["grads"][name], res[1]["grads"][name]), name          # both ranks hold the same reduced buffer
        ratios[name] = mo.grad_ratio(g, res[0]["grads"][name])
    assert max(ratios.values()) <= (1e-6 if one_leaf else REF_BF16_BOUND), max(ratios.items(), key=lambda kv: kv[1])
    assert float(np.median(list(ratios.values()))) <= 2.55e-2


def _tp1_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch.distributed as dist
    from dynamictreeattn_amd import ops
    from dynamictreeattn_amd.token_trie import TokenTrie
    from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    # (1) the sharded autograd Function of the public entry on RCCL: with one rank the shard is the whole vocabulary
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(40, 512, generator=g) * 3).to(DEV).requires_grad_(True)
    labels = torch.randint(0, 512, (40,), generator=g).to(DEV)
    go = [torch.randn(40, generator=g).to(DEV) for _ in range(2)]
    lp, ent = ops.logprob_entropy(logits, labels, 0.7, True, dist.group.WORLD)          # per-shard statistics + ONE MAX + ONE SUM all-reduce
    ((lp * go[0]).sum() + (ent * go[1]).sum()).backward()
    res = {"lp": lp.detach().cpu(), "ent": ent.detach().cpu(), "g": logits.grad.cpu()}
    logits.grad = None
    lp0, ent0 = ops.logprob_entropy(logits, labels, 0.7, True, None)
    ((lp0 * go[0]).sum() + (ent0 * go[1]).sum()).backward()
    res.update(lp0=lp0.detach().cpu(), ent0=ent0.detach().cpu(), g0=logits.grad.cpu())
    # (2) the engine with the LM-head vocabulary "split" over the 1-rank group (BASELINE config 4's code path: shard statistics,
    #     combine, asynchronous all-reduce of dh under the weight-gradient GEMM) against the plain engine
    out = {}
    for tp in (False, True):
        model = _model()
        seqs = synth.as_tensors(synth.make_case(TAU2_BIN)); att = _att(len(seqs))
        t = TokenTrie(seqs, att, device=DEV); t.backward_permute()
        e = TreeTrainingEngine(model.config, DEV, torch.bfloat16, 1024)
        if tp:
            e.tp_group = dist.group.WORLD
        loss = e.backward(model, t, mo.default_loss, 2048)
        out[tp] = (loss, {n: p.grad.float().cpu() for n, p in model.named_parameters()})
    res["engine"] = out
    torch.save(res, os.path.join(outdir, "tp1.pt"))
    dist.destroy_process_group()


def test_vocab_parallel_code_path_on_rccl_with_one_rank(tmp_path):
    """The collectives of BASELINE config 4 (MAX + packed SUM of the shard statistics, the dh all-reduce) executed by RCCL: a 1-rank
    "nccl" group, where every reduce is the identity - the sharded Function must equal the unsharded kernels, and the engine with
    `tp_group` set must equal the plain engine within bf16 rounding of the head stage."""
    _run(_tp1_worker, 1, (_free_port(), str(tmp_path)))
    r = torch.load(os.path.join(str(tmp_path), "tp1.pt"), weights_only=True)
    assert torch.allclose(r["lp"], r["lp0"], atol=2e-6) and torch.allclose(r["ent"], r["ent0"], atol=5e-6)
    assert torch.allclose(r["g"], r["g0"], atol=2e-6, rtol=1e-5)
    (l0, g0), (l1, g1) = r["engine"][False], r["engine"][True]
    assert abs(l0 - l1) <= 1e-3 * abs(l0)
    assert max(mo.grad_ratio(g0[n], g1[n]) for n in g0) <= 2e-2


def _vp_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import torch.distributed as dist
    from dynamictreeattn_amd import vocab_parallel as vp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks share the card: gloo carries the 2 small all-reduces
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    out = {}
    for name, c in gold.items():
        g = torch.Generator().manual_seed(c["seed"])
        logits = torch.randn(c["R"], c["V"], generator=g) * 3
        labels = torch.randint(0, c["V"], (c["R"],), generator=g).to(DEV)
        go_lp = torch.randn(c["R"], generator=g).to(DEV); go_ent = torch.randn(c["R"], generator=g).to(DEV)
        Vp = c["V"] // world
        shard = logits[:, rank * Vp:(rank + 1) * Vp].contiguous().to(DEV).requires_grad_(True)       # fp32 shard [R, V/tp]
        lp, ent = vp.gather_logprobs_entropy(shard, labels, temperature=c["temp"], tp_group=dist.group.WORLD)
        lp_only = vp.gather_logprobs(shard.detach(), labels, temperature=c["temp"], tp_group=dist.group.WORLD, chunk_size=7)
        (lp * go_lp).sum().backward(retain_graph=True)
        g1 = shard.grad.clone(); shard.grad = None
        (ent * go_ent).sum().backward()
        # the engine's shape quirk through the sharded path too: [1,B,V/tp] logits with [1,B-1] labels
        lpq, entq = vp.gather_logprobs_entropy(shard.detach()[None], labels[None, :-1], temperature=c["temp"], tp_group=dist.group.WORLD)
        out[name] = {"lp": lp.detach().cpu(), "ent": ent.detach().cpu(), "lp_only": lp_only.cpu(), "g_lp": g1.cpu(), "g_ent": shard.grad.cpu(),
                     "lpq": lpq.cpu(), "entq": entq.cpu()}
    torch.save(out, os.path.join(outdir, f"vp{rank}.pt"))
    dist.destroy_process_group()


def test_public_sharded_gather_logprobs_two_ranks_vs_reference_golden(tmp_path):
    """BASELINE config 4's public entry: each of 2 ranks holds half the vocabulary as fp32 CUDA logits; values equal the reference's
    recorded ones, and the gradient on the own shard equals the reference's recorded gradient columns (the golden keeps the first 48
    columns, which lie in rank 0's shard for every case, and the per-row |.| sums, which add up over the shards)."""
    _run(_vp_worker, 2, (_free_port(), str(tmp_path)))
    gold = torch.load(os.path.join(GOLD, "logprob_cases.pt"), weights_only=True)
    res = [torch.load(os.path.join(str(tmp_path), f"vp{r}.pt"), weights_only=True) for r in range(2)]
    assert set(res[0]) == set(gold) == {"small", "temp", "vocab"}
    for name, c in gold.items():
        for r in (0, 1):
            o = res[r][name]
            assert o["lp"].dtype == torch.float32 and o["lp"].shape == (c["R"],) and o["ent"].shape == (c["R"],)
            assert (o["lp"] - c["logprobs"]).abs().max() <= 2e-5 * (1 + c["logprobs"].abs().max()), (name, r)
            assert (o["ent"] - c["entropy"]).abs().max() <= 5e-5 * (1 + c["entropy"].abs().max()), (name, r)
            assert (o["lp_only"] - c["logprobs_only"]).abs().max() <= 2e-5 * (1 + c["logprobs_only"].abs().max()), (name, r)
            assert o["lpq"].shape == (1, c["R"] - 1) and o["entq"].shape == (1, c["R"])
            assert torch.allclose(o["lpq"][0], o["lp"][:-1], atol=1e-6) and torch.allclose(o["entq"][0], o["ent"], atol=1e-6)
        assert c["V"] // 2 >= 48 or name == "small"
        ncol = min(48, c["V"] // 2)
        assert torch.allclose(res[0][name]["g_lp"][:, :ncol], c["grad_lp_head"][:, :ncol], atol=5e-6, rtol=1e-4), name
        assert torch.allclose(res[0][name]["g_ent"][:, :ncol], c["grad_ent_head"][:, :ncol], atol=5e-6, rtol=1e-4), name
        if ncol < 48:                                                       # "small": V/2 = 32, the rest of the 48 columns is rank 1's
            assert torch.allclose(res[1][name]["g_lp"][:, :48 - ncol], c["grad_lp_head"][:, ncol:], atol=5e-6, rtol=1e-4), name
            assert torch.allclose(res[1][name]["g_ent"][:, :48 - ncol], c["grad_ent_head"][:, ncol:], atol=5e-6, rtol=1e-4), name
        row_lp = res[0][name]["g_lp"].abs().sum(-1) + res[1][name]["g_lp"].abs().sum(-1)
        row_ent = res[0][name]["g_ent"].abs().sum(-1) + res[1][name]["g_ent"].abs().sum(-1)
        assert torch.allclose(row_lp, c["grad_lp_rowsum_abs"], rtol=2e-4), name
        assert torch.allclose(row_ent, c["grad_ent_rowsum_abs"], rtol=2e-4, atol=1e-5), name


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and WORLD_SIZE unset (exp/exp_dp.py:28-49 is the protocol it reports): the parent
    starts the ranks before touching the GPU, relays rank 0's single JSON line and its exit code.  Rehearsal backend gloo, the two
    ranks share this box's card, one timed step - the launch path is what is under test."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["strong_scaling"]["value"] > 0 and 0 < out["strong_scaling"]["balance"] <= 1.0
    assert "gloo" in out["config"]["grad_allreduce"]
    # a failing rank must fail the parent
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=dict(env, DTA_BENCH_BACKEND="nccl"))
    assert bad.returncode != 0 and "RCCL needs one" in bad.stderr
