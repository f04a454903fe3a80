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
    out["kernel_time_ms_per_step"] = total_ns / 1e6 / n_steps
    out["gpu_busy_frac_of_step"] = out["kernel_time_ms_per_step"] / b["ms_per_step"]
    short = lambda n: (n.split("(anonymous namespace)::")[1].split("(")[0] if "(anonymous namespace)::" in n else n.split("(")[0])[:90]
    out["top_kernels"] = [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                           "ms_per_step": float(r["TotalDurationNs"]) / 1e6 / n_steps, "pct": float(r["Percentage"])} for r in rows[:14]]
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
print(json.dumps(out, indent=1))
