#!/usr/bin/env python3
"""Summarise one scripts/profile_round.sh output directory: per-kernel stats of the bench under rocprofv3 --kernel-trace, the
fraction of the step the GPU is busy, and the host synchronisation calls per step from the --hip-trace pass.
usage: python scripts/prof_summary.py gpurun_out/prof_r2 > summary.json"""
import csv, glob, json, os, sys

root = sys.argv[1]
out = {}


def first(pattern):
    c = sorted(glob.glob(os.path.join(root, pattern), recursive=True))
    return c[0] if c else None


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return None


ks = first("trace/**/*kernel_stats.csv")
b = bench_line(os.path.join(root, "bench_under_rocprof.json"))
if ks and b:
    rows = list(csv.DictReader(open(ks)))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    n_steps = b["warmup"] + 2 * b["steps"]                      # warm-up + timed leg + roofline leg
    out["bench_under_kernel_trace"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "roofline_frac": b["roofline"]["frac"],
                                       "dkv_avg_launch_ms_by_events": b["roofline"]["avg_launch_ms"]}
    out["kernel_time_ms_per_step_incl_setup"] = total_ns / 1e6 / n_steps
    short = lambda n: (n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n.split("(")[0])[:90]
    out["top_kernels"] = [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "ms_per_step": float(r["TotalDurationNs"]) / 1e6 / n_steps, "pct": float(r["Percentage"])} for r in rows[:14]]
kt = first("trace/**/*kernel_trace.csv")
if kt:
    # GPU timeline per step: a step starts at its packed trie's preorder_meta_kernel (on the engine's stream; the TokenTrie's own kernels
    # run on a side stream UNDER the previous step's backward); busy = union of the kernel intervals up to the next one
    recs = []
    for r in csv.DictReader(open(kt)):
        recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    recs.sort()
    marks = [i for i, x in enumerate(recs) if "preorder_meta_kernel" in x[2]]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = recs[a:b]
        if sum(1 for x in seg if "tree_attn_bwd_dkv2_kernel" in x[2]) < 2:
            continue                                     # a statistics-only TokenTrie (bench set-up), not a step
        span = recs[b][0] - recs[a][0]
        busy, cur_end = 0, seg[0][0]
        for s0, e0, _ in seg:
            if e0 > cur_end:
                busy += e0 - max(s0, cur_end); cur_end = e0
        steps.append({"gpu_span_ms": round(span / 1e6, 2), "gpu_busy_ms": round(busy / 1e6, 2), "busy_frac": round(busy / span, 4), "kernels": len(seg)})
    steady = [x for x in steps if x["gpu_span_ms"] < 2 * min(y["gpu_span_ms"] for y in steps)] if steps else []
    out["gpu_timeline_per_step"] = steady
    if steady:
        out["gpu_busy_frac_of_step"] = round(sum(x["gpu_busy_ms"] for x in steady) / sum(x["gpu_span_ms"] for x in steady), 4)
        out["gpu_busy_note"] = ("from the kernel trace's own timestamps: step = one preorder_meta_kernel (packing, first kernel of an engine call on its stream) to the "
                                "next; under rocprofv3 tracing, which slows the host side")
hs = first("hip/**/*hip_api_stats.csv") or first("hip/**/*hip_stats.csv")
hb = bench_line(os.path.join(root, "bench_under_hiptrace.json"))
if hs and hb:
    n_steps = hb["warmup"] + 2 * hb["steps"]
    rows = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(hs))}
    sync = {k: v for k, v in rows.items() if any(s in k for s in ("Synchronize", "hipMemcpy", "hipMemcpyAsync", "hipEventQuery", "hipStreamWaitEvent"))}
    out["hip_api_calls_total"] = sync
    out["hip_api_calls_per_step"] = {k: v / n_steps for k, v in sync.items()}
    out["launches_per_step"] = sum(v for k, v in rows.items() if "Launch" in k) / n_steps
    out["note"] = (f"{n_steps} steps in the run (warm-up {hb['warmup']} + timed {hb['steps']} + roofline leg {hb['steps']}); the bench itself "
                   "synchronises twice per leg around the timed region; hipMemcpyAsync counts every H2D table upload and the D2H reads")
ht = first("hip/**/*hip_api_trace.csv")
if ht:
    # steady-state host synchronisation per step: a step ends where the host waits for the loss value (hipEventSynchronize on the
    # asynchronous copy issued before the backward was launched; a blocking hipMemcpyWithStream before round 3's second half)
    BLOCK = ("hipMemcpyWithStream", "hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy")
    ev = []
    for r in csv.DictReader(open(ht)):
        fn = r["Function"]
        if fn in BLOCK or "Launch" in fn:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fn))
    ends = [i for i, (a, b, fn) in enumerate(ev) if fn in ("hipMemcpyWithStream", "hipEventSynchronize") and b - a > 20e6]
    steps = []
    for a, b in zip(ends[:-1], ends[1:]):
        seg = ev[a + 1:b + 1]
        n_launch = sum(1 for x in seg if "Launch" in x[2])
        if n_launch < 1000 or n_launch > 2200:
            continue                                     # leg boundary / first step of the process
        blk = [x for x in seg if x[2] in BLOCK]
        steps.append({"launches": n_launch, "blocking_calls": {k: sum(1 for x in blk if x[2] == k) for k in sorted({x[2] for x in blk})},
                      "host_blocked_ms": round(sum(x[1] - x[0] for x in blk) / 1e6, 2), "step_ms": round((ev[b][1] - ev[a][1]) / 1e6, 2)})
    out["steady_state_steps_from_hip_trace"] = steps
    out["steady_state_note"] = ("per step: the host blocks in the TokenTrie's LCP/leaf-table read (side stream: two small kernels) and in the wait for the loss value "
                                "(lands when the FORWARD is done; the backward is queued by then); everything else is asynchronous launches and copies")
print(json.dumps(out, indent=1))
