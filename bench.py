#!/usr/bin/env python3
"""bench.py — tree-attn fwd+bwd tokens/s, Qwen3-0.6B bf16, synthetic tau2-16k-shaped tries.

One *step* = one pass of the hot path over one batch: every rank takes one bin of the global batch
(N_ranks tau2-shaped calls merged and partitioned by ``LB_by_DFS_and_TM`` before the clock starts, as
data_parallel.py does offline), builds its TokenTrie (HIP LCP/leafization), permutes it for backward, runs
``TreeTrainingEngine.backward`` (HIP tree attention fwd+bwd inside the full model pass) and the
ranks sum their parameter gradients with one RCCL all-reduce.  Timed sync-to-sync like run.py:90-108;
metric = Σ original-sequence tokens / wall (run_all.py:156-159), MAX over ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant
attention kernel, HIP-event timed on the launch stream) and `cpu_baseline` (the oracle's
restatement of the reference schedule timed on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from dynamictreeattn_amd import dp, ops, synth
from dynamictreeattn_amd.model import Qwen3TreeLM, make_config
from dynamictreeattn_amd.token_trie import TokenTrie
from dynamictreeattn_amd.tree_training_engine import TreeTrainingEngine

MODEL_NAME = {"qwen3-0.6b": "Qwen3-0.6B", "qwen3-4b": "Qwen3-4B"}
PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16 peak, MI355X_MICROARCH.md "Chip-level parameters"
ATTACH = {"w_logprobs": -1.0, "w_entropy": 0.1}      # run_all.py:11-14
DKV_KERNEL = "tree_attn_bwd_dkv2_kernel"             # the dominant kernel as rocprofv3 names it (8-wave dK/dV)


def loss_fn(logprob, entropy, attachment):            # run.py:149-152
    return attachment["w_logprobs"] * logprob.mean() + attachment["w_entropy"] * entropy.mean()


def build_model(cfg: dict, device, dtype, seed: int = 0):
    m = Qwen3TreeLM(cfg).to(device=device, dtype=dtype)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm.weight") or "layernorm" in name:
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(dtype))
    return m.train()


def host_threads() -> int:
    """Cores this process may actually use (the GPU box gives one GPU's share, 16), not os.cpu_count()."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline_worker():
    """The oracle's restatement of the reference push/pop schedule (tree_training_engine.py:555-616) on
    the host cores, Qwen3-0.6B dims, on a bounded tau2-shaped sample.  Runs in a child process."""
    import numpy as np
    from oracle import model_oracle as mo
    from oracle import trie_oracle as to
    threads = host_threads()
    torch.set_num_threads(threads)
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    dtype = torch.bfloat16 if ("avx512_bf16" in flags or "amx_bf16" in flags) else torch.float32    # reference dtypes: run.py:122-126
    case = {"kind": "tau2", "seed": 0, "G": 3, "sys_len": 750, "turns": 3, "lo": 75, "hi": 340, "cap": 6144}
    seqs = [np.asarray(s, dtype=np.int64) for s in synth.make_case(case)]
    w = {k: v.requires_grad_(True) for k, v in mo.init_weights(mo.QWEN3_0P6B, seed=0, dtype=dtype).items()}
    t0 = time.time()
    trie = to.TokenTrieOracle(seqs, [dict(ATTACH) for _ in seqs]); trie.backward_permute()
    eng = mo.StackEngineOracle(mo.QWEN3_0P6B, w, max(len(s) for s in seqs), dtype=dtype)
    eng.backward(trie, mo.default_loss, 2048)
    dt = time.time() - t0
    st = trie.get_stats("backward", 2048)
    name = "bf16" if dtype == torch.bfloat16 else "fp32 (host has no avx512_bf16/amx)"
    print("CPU_BASELINE " + json.dumps({
        "value": st["n_tokens"] / dt, "unit": "tokens/s", "cores": threads, "kind": "port",
        "sample": f"tau2-shaped call at 3/8 length scale (G=3, sys 750, 3 turns of 75-340): {st['n_sequences']} seqs, "
                  f"{st['n_tokens']} tokens, {st['n_tree_tokens']} tree tokens, Qwen3-0.6B {name}, reference push/pop schedule, "
                  f"block_size 2048, {dt:.1f} s",
        "tree_tokens_per_s": st["n_tree_tokens"] / dt}), flush=True)


def cpu_baseline(limit_s: int = 240):
    """Bounded: the child is killed after `limit_s` and the leg reports null instead of stalling the bench."""
    import subprocess
    print(f"[bench] timing the CPU baseline (oracle, host cores) in a child process, limit {limit_s}s ...", file=sys.stderr, flush=True)
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=limit_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "tokens/s", "cores": host_threads(), "kind": "port", "sample": "worker failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "tokens/s", "cores": host_threads(), "kind": "port", "sample": f"worker exceeded {limit_s}s and was stopped"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--block-size", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--model", default="qwen3-0.6b", choices=["qwen3-0.6b", "qwen3-4b"])
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    local = local % max(n_dev, 1)               # rehearsal on a 1-GPU box: ranks share the card (gloo only, below)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # RCCL ("nccl") needs one GPU per rank; DTA_BENCH_BACKEND=gloo rehearses the N>1 code path on one GPU
        backend = os.environ.get("DTA_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    cfg = synth.QWEN3_0P6B if args.model == "qwen3-0.6b" else synth.QWEN3_4B
    model = build_model(cfg, dev, torch.bfloat16)
    engine = TreeTrainingEngine(make_config(cfg), dev, torch.bfloat16, max_seq_len=16384)     # run_all.py:86
    V = cfg["vocab_size"]

    # synthetic global batches, one per step: `world` tau2-shaped calls merged (CPU LongTensors, as the
    # reference's .pt batches are)
    total_steps = args.warmup + args.steps
    batches = []
    for s in range(total_steps):
        seqs = []
        for r in range(world):
            seqs += synth.as_tensors(synth.tau2(seed=s * world + r, V=V))
        batches.append(seqs)

    stats_acc = {"n_tokens": 0, "n_tree_tokens": 0, "pairs": 0}

    # The partition is OFFLINE in the reference (data_parallel.py writes {name}_bin{k}.pt ahead of the runs,
    # exp/exp_dp.py:43-49) and its timed region is trie build + permute + engine call per bin (run.py:90-108):
    # same here — bins are planned before the clock starts, everything from TokenTrie(...) on is timed.
    my_seqs = [[seqs[i] for i in dp.my_bin(seqs, rank, world, "backward", args.block_size)] for seqs in batches]

    # N > 1: plain post-backward gradient all-reduce (SUM, no averaging) by default; DTA_BENCH_OVERLAP=1 runs it bucket by
    # bucket under the backward pass (dp.OverlappedGradAllReduce, gloo-tested; opt-in until it has run on RCCL: this
    # round's GPU boxes have one GPU)
    overlap = world > 1 and os.environ.get("DTA_BENCH_OVERLAP", "0") == "1"
    reducer = dp.OverlappedGradAllReduce(model.parameters()) if overlap else None

    def step(mine, timed: bool):
        model.zero_grad(set_to_none=True)
        trie = TokenTrie(mine, [dict(ATTACH) for _ in mine])
        trie.backward_permute()
        if reducer is not None:
            reducer.start()
        loss = engine.backward(model, trie, loss_fn, args.block_size)
        if reducer is not None:
            reducer.finish()
        elif world > 1:
            dp.allreduce_grads(model.parameters())
        if timed:
            st = trie.get_stats("backward", args.block_size)
            stats_acc["n_tokens"] += st["n_tokens"]; stats_acc["n_tree_tokens"] += st["n_tree_tokens"]
            stats_acc["pairs"] += st["sum_depth"] + st["n_tree_tokens"]
        return loss

    for s in range(args.warmup):
        step(my_seqs[s], False)
        model.zero_grad(set_to_none=True)

    timer = ops.KernelTimer()
    ops.KernelTimer.active = timer
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(args.warmup, total_steps):
        step(my_seqs[s], True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.time() - t0
    ops.KernelTimer.active = None

    tot = torch.tensor([wall, float(stats_acc["n_tokens"]), float(stats_acc["n_tree_tokens"])], device=dev, dtype=torch.float64)
    if world > 1:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall, n_tokens, n_tree = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        n_tokens, n_tree = float(tot[1]), float(tot[2])

    # roofline of the dominant kernel (dK/dV): algorithmic FLOPs = 4 products x 2·Hq·D per visible pair
    ms = timer.totals_ms()
    L, Hq, D = cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["head_dim"]
    pairs = stats_acc["pairs"]
    def tf(coef, key):
        t, n = ms[key]
        return (coef * Hq * D * pairs * L) / (t * 1e-3) / 1e12 if t > 0 else 0.0, (t / n if n else 0.0)
    dkv_tf, dkv_ms = tf(8, "bwd_dkv")
    dq_tf, dq_ms = tf(2, "bwd_dq")
    fwd_tf, fwd_ms = tf(4, "fwd")
    attn_ms_total = ms["fwd"][0] + ms["bwd_dq"][0] + ms["bwd_dkv"][0] + ms["bwd_dkv_finalize"][0]
    all_tf = (14 * Hq * D * pairs * L) / (attn_ms_total * 1e-3) / 1e12 if attn_ms_total > 0 else 0.0

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; take the latest
    # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE result for the same kernel and trie shape (profiles/)
    traffic, traffic_src = None, None
    try:
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_attn_traffic.json")))
        if cand:
            tj = json.load(open(cand[-1]))
            traffic = tj["kernels"][DKV_KERNEL]["hbm_bytes_per_launch"]
            traffic_src = os.path.relpath(cand[-1], ROOT) + " (tau2 seed-0 trie, one layer; 2*FETCH_SIZE+WRITE_SIZE)"
    except Exception:
        pass

    out = {
        "metric": f"tree-attn fwd+bwd tokens/sec, {MODEL_NAME[args.model]} tau2-16k tries", "value": n_tokens / wall, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "tau2-16k-shaped trie batch (8 rollouts x 6 turns, 2000-token shared prompt, 48 seqs ~180k tokens "
                               f"per call), one call per GPU per step, {MODEL_NAME[args.model]} random-init bf16, tree fwd+bwd, block_size {args.block_size}, permute=ours",
                   "calls_per_step": world, "balancer": "LB_by_DFS_and_TM" if world > 1 else "none",
                   "grad_allreduce": ("RCCL sum, 256 MB buckets overlapped with backward" if overlap else "RCCL sum after backward") if world > 1 else "none"},
        "tree_tokens_per_s": n_tree / wall,
        "roofline": {"bound": "mfma", "kernel": DKV_KERNEL, "achieved": dkv_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": dkv_tf / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": dkv_ms,
                     "flops_per_pair_per_layer": 8 * Hq * D, "pairs_per_step": pairs / max(args.steps, 1),
                     "other_kernels": {"tree_attn_fwd_kernel": {"achieved": fwd_tf, "avg_launch_ms": fwd_ms, "flops_per_pair_per_layer": 4 * Hq * D},
                                       "tree_attn_bwd_dq_kernel": {"achieved": dq_tf, "avg_launch_ms": dq_ms, "flops_per_pair_per_layer": 2 * Hq * D},
                                       "tree_attn_bwd_dkv_finalize_kernel": {"avg_launch_ms": (ms["bwd_dkv_finalize"][0] / ms["bwd_dkv_finalize"][1]) if ms["bwd_dkv_finalize"][1] else 0.0},
                                       "attention_fwd+bwd_14HqD": {"achieved": all_tf, "ms_per_step": attn_ms_total / max(args.steps, 1)}}},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
