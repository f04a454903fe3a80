#!/usr/bin/env python3
"""bench.py — tree-attn fwd+bwd tokens/s on synthetic tries (BASELINE.json metric; SURVEY §8d protocol).

One *step* = one pass of the hot path over one batch: build the TokenTrie (HIP LCP / leafization), permute it for backward,
run ``TreeTrainingEngine.backward`` (HIP tree attention fwd+bwd inside the full model pass), and — for N > 1 — sum the
parameter gradients over the ranks (RCCL, in place, overlapped with the backward).  Timed sync-to-sync like run.py:90-108;
metric = Σ original-sequence tokens / wall (run_all.py:156-159), MAX over ranks.

Legs of one invocation (all in the ONE JSON line rank 0 prints):
* ``value``            WEAK scaling: step s gives every rank one tau2-shaped call (N calls merged and split by
                       ``LB_by_DFS_and_TM`` before the clock starts, as data_parallel.py does offline).  No kernel timers run here.
* ``strong_scaling``   (N > 1) the reference's own DP protocol (exp/exp_dp.py:28-49, exp/calc_time.py:24-42): a FIXED batch of calls,
                       each split N ways; throughput = Σ n_tokens / Σ_calls max_rank(time), plus the balance Σ t_k / (N · max t_k).
* ``roofline``         the same steps once more with HIP events around every C-ABI launch (their cost is reported as
                       ``timer_overhead_frac``): dominant MFMA kernel vs the 2.5 PFLOP/s peak, HBM-bound kernels vs 8 TB/s.
* ``cpu_baseline``     (N = 1) the oracle's restatement of the reference schedule on the host cores: config 1 and a tau2-shaped sample.

Other workloads: ``--workload wide`` (BASELINE config 5: 64 x 16 384 over a 1 024-token root, fp16), ``--model qwen3-4b | qwen3-8b | qwen3-14b``,
``--vocab-parallel`` (BASELINE config 4: the LM-head vocabulary split over the ranks, vocab_parallel.py:82-396).
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

MODEL_NAME = {"qwen3-0.6b": "Qwen3-0.6B", "qwen3-4b": "Qwen3-4B", "qwen3-8b": "Qwen3-8B", "qwen3-14b": "Qwen3-14B"}
MODEL_CFG = {"qwen3-0.6b": synth.QWEN3_0P6B, "qwen3-4b": synth.QWEN3_4B, "qwen3-8b": synth.QWEN3_8B, "qwen3-14b": synth.QWEN3_14B}
PEAK_TFLOPS = 2500.0               # dense MFMA bf16 / f16 peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0              # HBM3E spec, same table (6 290 GB/s measured copy)
T_START = time.time()
ATTACH = {"w_logprobs": -1.0, "w_entropy": 0.1}      # run_all.py:11-14
# the bounded tau2-shaped sample the CPU baseline is timed on (and the GPU too, for a like-for-like ratio): 3/8 length scale
CPU_SAMPLE_CASE = {"kind": "tau2", "seed": 0, "G": 3, "sys_len": 750, "turns": 3, "lo": 75, "hi": 340, "cap": 6144}
DKV_KERNEL = "tree_attn_bwd_dkv2_kernel"             # the dominant kernel as rocprofv3 names it (8-wave dK/dV)
HBM_KERNELS = {"dta_logprob_entropy_fwd": "logprob_entropy_fwd_kernel", "dta_logprob_entropy_bwd": "logprob_entropy_bwd_kernel",
               "dta_rmsnorm_fwd": "rmsnorm_fwd_kernel", "dta_rmsnorm_bwd": "rmsnorm_bwd_kernel",
               "dta_qk_norm_rope_fwd": "qk_norm_rope_fwd_kernel", "dta_qk_norm_rope_bwd": "qk_norm_rope_bwd_kernel",
               "dta_swiglu_fwd": "swiglu_fwd_kernel", "dta_swiglu_bwd": "swiglu_bwd_kernel"}


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


def note(rank: int, msg: str) -> None:
    """Progress on stderr (rank 0): the JSON line on stdout stays the only stdout output."""
    if rank == 0:
        print(f"[bench +{time.time() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
    """Cores this process may actually use (the GPU box gives one GPU's share, 16), not os.cpu_count()."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


# --------------------------------------------------------------------------------------------------------------
# CPU baseline (child process; the only place besides tests/ and smoke() that touches oracle/)
# --------------------------------------------------------------------------------------------------------------
def cpu_baseline_worker():
    """The oracle's restatement of the reference push/pop schedule (tree_training_engine.py:555-616) on the host cores, Qwen3-0.6B
    dims: BASELINE config 1 and a bounded tau2-shaped sample."""
    import numpy as np
    from oracle import model_oracle as mo
    from oracle import trie_oracle as to
    threads = host_threads()
    torch.set_num_threads(threads)
    flags = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    dtype = torch.bfloat16 if ("avx512_bf16" in flags or "amx_bf16" in flags) else torch.float32    # reference dtypes: run.py:122-126
    name = "bf16" if dtype == torch.bfloat16 else "fp32 (host has no avx512_bf16/amx)"
    w = {k: v.requires_grad_(True) for k, v in mo.init_weights(mo.QWEN3_0P6B, seed=0, dtype=dtype).items()}

    def run(seqs):
        for v in w.values():
            v.grad = None
        t0 = time.time()
        trie = to.TokenTrieOracle(seqs, [dict(ATTACH) for _ in seqs]); trie.backward_permute()
        eng = mo.StackEngineOracle(mo.QWEN3_0P6B, w, max(len(s) for s in seqs), dtype=dtype)
        eng.backward(trie, mo.default_loss, 2048)
        return time.time() - t0, trie.get_stats("backward", 2048)

    seqs1 = [np.asarray(s, dtype=np.int64) for s in synth.config1(0)]
    run(seqs1)                                                   # warm-up (thread pool, allocator), as run_all.py:88-92
    dt1, st1 = run(seqs1)
    seqs2 = [np.asarray(s, dtype=np.int64) for s in synth.make_case(CPU_SAMPLE_CASE)]
    dt2, st2 = run(seqs2)
    # the FULL tau2 call (48 seqs, ~180k tokens): estimated from the sample by its (query,key) pair count and tree tokens - attention
    # scales with the pairs, everything else with the tree tokens; it is not run (the leg is bounded to ~30 s of CPU work)
    full = to.TokenTrieOracle([np.asarray(s, dtype=np.int64) for s in synth.tau2(0)]); full.backward_permute()
    stf = full.get_stats("backward", 2048)
    pairs = lambda st: st["sum_depth"] + st["n_tree_tokens"]
    est_lo = dt2 * stf["n_tree_tokens"] / st2["n_tree_tokens"]
    est_hi = dt2 * pairs(stf) / pairs(st2)
    print("CPU_BASELINE " + json.dumps({
        "value": st2["n_tokens"] / dt2, "unit": "tokens/s", "cores": threads, "kind": "port",
        "sample": f"tau2-shaped call at 3/8 length scale (G=3, sys 750, 3 turns of 75-340): {st2['n_sequences']} seqs, {st2['n_tokens']} tokens, "
                  f"{st2['n_tree_tokens']} tree tokens, Qwen3-0.6B {name}, reference push/pop schedule, block_size 2048, {dt2:.1f} s on {threads} threads; "
                  "the GPU `value` is measured on the FULL tau2 call (48 seqs, ~180k tokens, ~25.5k tree tokens) — shallower tries favour the CPU",
        "tree_tokens_per_s": st2["n_tree_tokens"] / dt2,
        "full_call": {"timed": False, "estimate_s": [round(est_lo, 1), round(est_hi, 1)],
                      "why": f"one full tau2 call would take between {est_lo:.0f} s (scaling the sample by tree tokens, x{stf['n_tree_tokens'] / st2['n_tree_tokens']:.1f}) and "
                             f"{est_hi:.0f} s (by (query,key) pairs, x{pairs(stf) / pairs(st2):.1f}) on these {threads} threads: outside the bounded CPU leg, so it is not run"},
        "config1": {"value": st1["n_tokens"] / dt1, "unit": "tokens/s", "seconds": dt1,
                    "sample": f"BASELINE config 1: 4 rollouts x 256 tokens sharing a 256-token prompt ({st1['n_tokens']} tokens, {st1['n_tree_tokens']} tree tokens), "
                              f"Qwen3-0.6B {name}, tree backward after one warm-up"}}), flush=True)


def cpu_baseline(limit_s: int = 300):
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


# --------------------------------------------------------------------------------------------------------------
def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes of
    `python -m torch.distributed.run` (the same command line the driver uses), BEFORE this process has made any GPU call -
    a process that has touched the GPU is never re-executed.  The children inherit stdout / stderr, so rank 0's single JSON
    line is this process's output; the exit code is the launcher's (non-zero if any rank failed)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] WORLD_SIZE is not set: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--block-size", type=int, default=2048)
    ap.add_argument("--model", default="qwen3-0.6b", choices=list(MODEL_NAME))
    ap.add_argument("--workload", default="tau2", choices=["tau2", "wide"])
    ap.add_argument("--vocab-parallel", action="store_true", help="BASELINE config 4: split the LM-head vocabulary over the ranks (every rank runs the same batch)")
    ap.add_argument("--engine-mode", default="auto", choices=["auto", "packed", "stack"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling leg at N > 1")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # decided before anything below touches the GPU
        sys.exit(self_launch(args.gpus))
    wide = args.workload == "wide"
    steps = args.steps if args.steps is not None else (2 if wide else 6)
    warmup = args.warmup if args.warmup is not None else (1 if wide else 2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    n_dev = torch.cuda.device_count()
    backend = None
    if world > 1:
        # RCCL ("nccl") needs one GPU per rank; DTA_BENCH_BACKEND=gloo rehearses the N>1 code path with ranks sharing a card
        backend = os.environ.get("DTA_BENCH_BACKEND", "nccl")
        if backend == "nccl" and n_dev < world:
            raise SystemExit(f"bench.py --gpus {world}: this node shows {n_dev} GPU(s) and RCCL needs one per rank "
                             "(DTA_BENCH_BACKEND=gloo rehearses the multi-rank path with the ranks sharing the card)")
    local = local % max(n_dev, 1)               # rehearsal on a 1-GPU box: ranks share the card (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if 0 < n_dev < world:                       # shared card: cap every rank's caching allocator at its share of the HBM; the engine's
        share = (n_dev * 0.92) / world          # footprint decisions (ops.free_hbm) honour the cap, so no rank plans with the whole card
        torch.cuda.set_per_process_memory_fraction(share, dev)
    elif os.environ.get("DTA_BENCH_HBM_FRACTION"):          # diagnostic: independent processes sharing one card, each under its own cap
        torch.cuda.set_per_process_memory_fraction(float(os.environ["DTA_BENCH_HBM_FRACTION"]), dev)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    note(rank, f"world {world}, backend {backend}, device {dev}")
    if os.environ.get("DTA_BENCH_WATCHDOG"):        # diagnostic: every rank dumps its Python stacks to stderr if it is still running after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["DTA_BENCH_WATCHDOG"]), repeat=False, file=sys.stderr)
    dtype = torch.float16 if wide else torch.bfloat16                     # BASELINE config 5 is fp16
    cfg = MODEL_CFG[args.model]
    model = build_model(cfg, dev, dtype)
    engine = TreeTrainingEngine(make_config(cfg), dev, dtype, max_seq_len=16384)     # run_all.py:86
    engine.mode = args.engine_mode
    engine.stack_fp32_grads = os.environ.get("DTA_STACK_FP32_GRADS", "1") == "1"          # diagnostic A/B switch of the block-wise walk's fp32 gradient sums
    tp = args.vocab_parallel and world > 1
    if tp:
        engine.tp_group = dist.group.WORLD
    V = cfg["vocab_size"]
    loss_scale = 1024.0 if wide else 1.0                                   # fp16 at depth 16 384 needs it (tests/test_gpu_engine.py)
    attach = {k: v * loss_scale for k, v in ATTACH.items()}

    def call(seed):
        if wide:
            return synth.as_tensors(synth.wide(seed=seed + 1, V=V))
        return synth.as_tensors(synth.tau2(seed=seed, V=V))

    # ---- gradient reduction (N > 1, data parallel): persistent flat buffers, every param.grad a view, reduced in place and
    # overlapped with the backward.  DTA_BENCH_OVERLAP=0 reduces after the backward instead.
    reducer = None
    if world > 1 and not tp:
        # one-card rehearsals (gloo, ranks sharing the GPU) reduce after the backward by default: with 3+ ranks on one card the
        # hook-launched asynchronous gloo all-reduces (host-staged copies of the CUDA buckets, issued while every rank's backward
        # kernels are still queued on the same GPU) stall the step for minutes, while the same reduce issued after the backward
        # runs at the 4-way time-sliced rate (gpurun records r3/bench_n4_gloo*.err).  RCCL always overlaps.
        shared_card = 0 < n_dev < world
        reducer = dp.GradReducer(model.parameters(), overlap=os.environ.get("DTA_BENCH_OVERLAP", "0" if (shared_card and backend == "gloo") else "1") == "1")

    def zero():
        if reducer is not None:
            reducer.zero_grad()
        else:
            model.zero_grad(set_to_none=True)

    def step(mine, acc=None):
        """One pass of the hot path over this rank's sequences (possibly none: an empty bin still takes part in the reduce)."""
        zero()
        t_begin = time.time()
        trie = TokenTrie(mine, [dict(attach) for _ in mine], device=dev)
        if mine:
            trie.backward_permute()
        if reducer is not None:
            reducer.start()
        loss = engine.backward(model, trie, loss_fn, args.block_size)
        t_compute = None
        if acc is not None and acc.get("split_times"):
            torch.cuda.synchronize(); t_compute = time.time() - t_begin
        if reducer is not None:
            reducer.finish()
        if acc is not None:
            st = trie.get_stats("backward", args.block_size) if mine else {"n_tokens": 0, "n_tree_tokens": 0, "sum_depth": 0}
            acc["n_tokens"] += st["n_tokens"]; acc["n_tree_tokens"] += st["n_tree_tokens"]
            acc["pairs"] += st["sum_depth"] + st["n_tree_tokens"]
        return loss, t_compute

    def timed(batches, acc):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        marks = []
        for b in batches:
            step(b, acc)                 # ends with the engine's loss.item(): the step's own work is done when it returns
            marks.append(time.time())
            if world > 1 and os.environ.get("DTA_BENCH_WATCHDOG"):
                print(f"[bench +{time.time() - T_START:6.1f}s] rank {rank}: step done in {marks[-1] - (marks[-2] if len(marks) > 1 else t0):.2f}s, mode {engine.last_mode}, "
                      f"reserved {torch.cuda.memory_reserved(dev) / 1e9:.1f} GB", file=sys.stderr, flush=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        acc["step_ms"] = [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + marks[:-1], marks)]
        return time.time() - t0

    def reduce_stats(wall, acc):
        tot = torch.tensor([wall, float(acc["n_tokens"]), float(acc["n_tree_tokens"]), float(acc["pairs"])], device=dev, dtype=torch.float64)
        if world > 1:
            mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            if tp:                                  # every rank ran the same batch: count it once
                return float(mx[0]), float(mx[1]), float(mx[2]), float(mx[3])
            return float(mx[0]), float(sm[1]), float(sm[2]), float(sm[3])
        return float(tot[0]), float(tot[1]), float(tot[2]), float(tot[3])

    new_acc = lambda **kw: dict({"n_tokens": 0, "n_tree_tokens": 0, "pairs": 0}, **kw)

    # ---- WEAK leg (= `value`): one call per rank per step.  The partition is OFFLINE in the reference (data_parallel.py writes
    # {name}_bin{k}.pt ahead of the runs, exp/exp_dp.py:43-49) and its timed region is trie build + permute + engine call per bin
    # (run.py:90-108): same here — bins are planned before the clock starts, everything from TokenTrie(...) on is timed.
    total_steps = warmup + steps
    note(rank, "model built; planning the per-rank batches")
    weak = []
    for s in range(total_steps):
        if tp:
            seqs = call(s)
            weak.append(seqs)
        else:
            seqs = []
            for r in range(world):
                seqs += call(s * world + r)
            weak.append([seqs[i] for i in dp.my_bin(seqs, rank, world, "backward", args.block_size)])
    # library warm-up (not steps): hipBLASLt resolves a solution and loads its code object per exact GEMM shape on first use, and the
    # packed row count differs from call to call; the row counts come from trie statistics computed before the clock starts
    note(rank, "batches planned; warming the GEMM shapes")
    if engine.mode != "stack":
        rows = set()
        for b in weak:
            if b:
                st_ = TokenTrie(b, device=dev).get_stats("forward")
                rows.add(engine.packed_rows(st_["n_tree_tokens"]))
        if not wide:
            for i_, r_ in enumerate(sorted(rows)):
                engine.warm_gemm_shapes(model, [r_])
                note(rank, f"GEMM shapes warmed for {r_} packed rows ({i_ + 1}/{len(rows)}); free HBM for this rank {ops.free_hbm(dev) / 1e9:.1f} GB")
    for s in range(warmup):
        step(weak[s])
        print(f"[bench +{time.time() - T_START:6.1f}s] rank {rank}: warm-up step {s + 1}/{warmup} done, engine mode {engine.last_mode}, "
              f"free HBM {ops.free_hbm(dev) / 1e9:.1f} GB, peak allocated {torch.cuda.max_memory_allocated(dev) / 1e9:.1f} GB", file=sys.stderr, flush=True)
    zero()
    acc = new_acc()
    wall = timed(weak[warmup:], acc)
    wall, n_tokens, n_tree, pairs = reduce_stats(wall, acc)
    note(rank, f"timed leg done: {steps} steps in {wall:.2f}s")
    engine_mode = engine.last_mode
    peak_hbm = torch.cuda.max_memory_allocated(dev)

    # ---- ROOFLINE leg: the same steps with HIP events on the launch stream around every C-ABI launch
    timer = ops.KernelTimer()
    ops.KernelTimer.active = timer
    acc_r = new_acc()
    wall_r = timed(weak[warmup:], acc_r)
    ops.KernelTimer.active = None
    ms = timer.totals_ms()
    nbytes = timer.totals_bytes()
    wall_r, _, _, _ = reduce_stats(wall_r, acc_r)
    note(rank, "roofline leg done")
    pairs_r = acc_r["pairs"]                                  # this rank's pairs: its own kernels' work
    L, Hq, D = cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["head_dim"]

    def tf(coef, key):
        t, n = ms.get(key, (0.0, 0))
        return (coef * Hq * D * pairs_r * L) / (t * 1e-3) / 1e12 if t > 0 else 0.0, (t / n if n else 0.0)
    dkv_tf, dkv_ms = tf(8, "bwd_dkv")
    dq_tf, dq_ms = tf(2, "bwd_dq")
    fwd_tf, fwd_ms = tf(4, "fwd")
    fin = ms.get("bwd_dkv_finalize", (0.0, 0))
    attn_ms_total = ms["fwd"][0] + ms["bwd_dq"][0] + ms["bwd_dkv"][0] + fin[0]
    all_tf = (14 * Hq * D * pairs_r * L) / (attn_ms_total * 1e-3) / 1e12 if attn_ms_total > 0 else 0.0
    hbm_rows = {}
    for api, kern in HBM_KERNELS.items():
        t, n = ms.get(api, (0.0, 0))
        if n:
            gbs = nbytes[api] / (t * 1e-3) / 1e9
            hbm_rows[kern] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                              "avg_launch_ms": t / n, "launches_per_step": n / max(steps, 1), "algorithmic_bytes_per_launch": nbytes[api] / n}

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

    # ---- TREE_FORWARD leg (run.py:28-61 -> engine.forward): TokenTrie + forward_permute + per-sequence logprobs, same calls
    fwd_leg = None
    if not wide and not tp:
        def fwd_step(mine):
            trie = TokenTrie(mine, device=dev)
            if mine:
                trie.forward_permute()
            engine.forward(model, trie)
            return trie.n_tokens
        zero()
        fwd_step(weak[0])                                      # warm-up (run_all.py:53-55)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); t0 = time.time()
        tok_f = sum(fwd_step(b) for b in weak[warmup:])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(); wall_f = time.time() - t0
        tf_ = torch.tensor([wall_f, float(tok_f)], device=dev, dtype=torch.float64)
        if world > 1:
            mxf = tf_.clone(); dist.all_reduce(mxf, op=dist.ReduceOp.MAX)
            smf = tf_.clone(); dist.all_reduce(smf, op=dist.ReduceOp.SUM)
            wall_f, tok_f = float(mxf[0]), float(smf[1])
        fwd_leg = {"value": tok_f / wall_f, "unit": "tokens/s", "ms_per_step": wall_f / steps * 1e3,
                   "definition": "sum n_tokens / wall of TokenTrie build + forward_permute + engine.forward (no-grad logprobs of every sequence, run.py:28-61), same calls as `value`"}
        note(rank, "tree_forward leg done")

    # ---- the GPU on the CPU baseline's own two workloads (N = 1): like-for-like ratios next to the full-call `value`
    same_sample = None
    if world == 1 and not wide and not args.no_cpu_baseline and args.model == "qwen3-0.6b":
        def gpu_rate(seqs_, reps=3):
            seqs_ = synth.as_tensors(seqs_)
            def once():
                zero()
                tr = TokenTrie(seqs_, [dict(ATTACH) for _ in seqs_], device=dev); tr.backward_permute()
                engine.backward(model, tr, loss_fn, 2048)
                return tr
            once()                                             # warm-up, as the CPU leg has
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(reps):
                tr = once()
            torch.cuda.synchronize(); dt = (time.time() - t0) / reps
            return {"value": tr.n_tokens / dt, "unit": "tokens/s", "ms": dt * 1e3, "n_tokens": tr.n_tokens}
        same_sample = {"gpu_config1": gpu_rate(synth.config1(0)), "gpu_same_sample": gpu_rate(synth.make_case(CPU_SAMPLE_CASE))}
        zero()
        note(rank, "like-for-like GPU legs done")

    # ---- STRONG leg (N > 1, data parallel): the reference's DP protocol on a fixed batch of calls
    strong = None
    if world > 1 and not tp and not args.no_strong and not wide:
        n_calls = max(steps, 1)
        per_call = []
        for s in range(n_calls):
            seqs = call(s % 16)                                                  # the 16-call batch of SURVEY §8d, in order
            bins = dp.all_bins(seqs, world, "backward", args.block_size)
            per_call.append(([seqs[i] for i in bins[rank]], sum(int(x.numel()) for x in seqs)))
        if engine.mode != "stack":
            engine.warm_gemm_shapes(model, {engine.packed_rows(TokenTrie(m_, device=dev).get_stats("forward")["n_tree_tokens"]) for m_, _ in per_call if m_})
        step(per_call[0][0])                                                     # warm-up of the smaller per-rank shapes
        t_tot, t_cmp, tok = [], [], 0
        for mine, n_tok in per_call:
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.time()
            _, tc = step(mine, {"n_tokens": 0, "n_tree_tokens": 0, "pairs": 0, "split_times": True})
            torch.cuda.synchronize()
            t_tot.append(time.time() - t0); t_cmp.append(tc); tok += n_tok
        note(rank, "strong-scaling leg done")
        tt = torch.tensor([t_tot, t_cmp], device=dev, dtype=torch.float64)        # [2, calls]
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        strong = {"value": tok / float(mx[0].sum()), "unit": "tokens/s", "calls": n_calls,
                  "definition": "sum n_tokens / sum_calls max_rank(time), every call split N ways by LB_by_DFS_and_TM (exp/exp_dp.py:28-49, exp/calc_time.py:24-42); "
                                "time = trie build + permute + backward + gradient all-reduce",
                  "balance": float((sm[1] / (world * mx[1])).mean()), "balance_definition": "mean over calls of sum_k t_k / (N * max_k t_k), t_k = rank k's time up to the reduce",
                  "ms_per_call": float(mx[0].mean()) * 1e3}

    workload = ("BASELINE config 5: wide/deep trie, 64 branches x 16 384 deep over a 1 024-token root (1 048 576 tokens, 984 064 tree tokens per call), "
                f"{MODEL_NAME[args.model]} random-init fp16, loss scale {loss_scale:g}") if wide else \
               ("tau2-16k-shaped trie batch (8 rollouts x 6 turns, 2000-token shared prompt, 48 seqs ~180k tokens per call), "
                f"{MODEL_NAME[args.model]} random-init bf16")
    out = {
        "metric": f"tree-attn fwd+bwd tokens/sec, {MODEL_NAME[args.model]} {'wide/deep' if wide else 'tau2-16k'} tries", "value": n_tokens / wall, "unit": "tokens/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None, "dtype": "f16" if wide else "bf16", "data": "synthetic",
        "config": {"workload": workload + f", tree fwd+bwd, block_size {args.block_size}, permute=ours; "
                               + ("every rank runs the SAME call with the LM-head vocabulary split over the ranks (config 4)" if tp else "one call per GPU per step"),
                   "calls_per_step": 1 if tp else world, "balancer": "LB_by_DFS_and_TM" if (world > 1 and not tp) else "none",
                   "engine_mode": engine_mode,
                   "grad_allreduce": ("none (vocabulary-parallel run: body replicated, head rows owned per rank)" if tp else
                                      (("RCCL" if backend == "nccl" else backend + " (rehearsal)") + (" sum in place on persistent flat buckets, overlapped with backward" if reducer.overlap else " sum in place after backward"))) if world > 1 else "none",
                   "value_is": "the WEAK-scaling leg (per-GPU work fixed); the reference's fixed-batch protocol is `strong_scaling`"},
        "tree_tokens_per_s": n_tree / wall, "peak_hbm_gb": peak_hbm / 1e9, "step_ms": acc.get("step_ms"), "step_ms_roofline_leg": acc_r.get("step_ms"),
        "strong_scaling": strong, "tree_forward": fwd_leg,
        "roofline": {"bound": "mfma", "kernel": DKV_KERNEL, "achieved": dkv_tf, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": dkv_tf / PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": dkv_ms,
                     "flops_per_pair_per_layer": 8 * Hq * D, "pairs_per_step": pairs_r / max(steps, 1),
                     "sustained_mfma_ceiling": {"value": 1729.0, "unit": "TFLOP/s", "frac_of_it": dkv_tf / 1729.0,
                                                "source": "profiles/r2_mfma_dvfs_probe.json: a loop of nothing but v_mfma_f32_32x32x16_bf16 on random operands, two waves per SIMD "
                                                          "(the chip lowers its clock under MFMA load; 2 455 on zeros) - `peak` stays the nominal 2.5 PFLOP/s"},
                     "measured_in": "a repeat of the timed steps with HIP events around every launch (not inside `value`'s timed region)",
                     "ms_per_step_with_timers": wall_r / steps * 1e3, "timer_overhead_frac": wall_r / wall - 1.0,
                     "other_kernels": {"tree_attn_fwd_kernel": {"bound": "mfma", "achieved": fwd_tf, "frac": fwd_tf / PEAK_TFLOPS, "avg_launch_ms": fwd_ms, "flops_per_pair_per_layer": 4 * Hq * D},
                                       "tree_attn_bwd_dq_kernel": {"bound": "mfma", "achieved": dq_tf, "frac": dq_tf / PEAK_TFLOPS, "avg_launch_ms": dq_ms, "flops_per_pair_per_layer": 2 * Hq * D,
                                                                   "note": "executes 6*Hq*D per pair (S and dP recomputed); see DESIGN.md §4 for why the backward stays two kernels"},
                                       "tree_attn_bwd_dkv_finalize_kernel": {"avg_launch_ms": (fin[0] / fin[1]) if fin[1] else 0.0},
                                       "attention_fwd+bwd_14HqD": {"bound": "mfma", "achieved": all_tf, "frac": all_tf / PEAK_TFLOPS, "ms_per_step": attn_ms_total / max(steps, 1)},
                                       **hbm_rows}},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            if same_sample is not None and cb.get("value"):
                # like for like: the GPU timed on exactly the two workloads the CPU ran (after one warm-up each, as the CPU leg)
                cb["gpu_same_sample"] = dict(same_sample["gpu_same_sample"], ratio_gpu_over_cpu=same_sample["gpu_same_sample"]["value"] / cb["value"])
                cb["gpu_config1"] = dict(same_sample["gpu_config1"], ratio_gpu_over_cpu=same_sample["gpu_config1"]["value"] / cb["config1"]["value"])
                cb["note"] = ("`value` of this object and `gpu_same_sample` are the SAME bounded sample; `config1` and `gpu_config1` the same BASELINE config 1; "
                              "the headline `value` of the line is the full tau2 call, which the CPU leg does not run (see full_call)")
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
