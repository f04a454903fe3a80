#!/bin/bash
# N independent single-rank bench processes on ONE card (no process group, no collectives), each capped at 0.92/N of the HBM:
# separates "several processes share the GPU" from "gloo" in the stalls of the N>2 one-card rehearsals.
N=${1:-4}; LIMIT=${2:-120}
mkdir -p gpurun_out/r3
FR=$(python3 -c "print(0.92/$N)")
for i in $(seq 0 $((N-1))); do
  DTA_BENCH_HBM_FRACTION=$FR DTA_BENCH_WATCHDOG=$((LIMIT-20)) timeout -k 10 $LIMIT python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline \
      > gpurun_out/r3/probe_n${N}_$i.json 2> gpurun_out/r3/probe_n${N}_$i.err &
done
wait
for i in $(seq 0 $((N-1))); do echo "== proc $i"; grep "bench +" gpurun_out/r3/probe_n${N}_$i.err | tail -4; python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r3/probe_n${N}_$i.json')); print('ms_per_step', d['ms_per_step'], d['step_ms'])
except Exception as e: print('no result', e)"; done
