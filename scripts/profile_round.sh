#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 evidence for one round, written under gpurun_out/prof_$1/.
#   1. kernel trace + stats of the default bench (per-kernel average durations -> profiles/<round>_bench_kernel_stats.csv)
#   2. HIP-API trace stats of a short bench (host synchronisations per step)
#   3. PMC counters of the three attention kernels on the tau2 seed-0 trie, one counter group per pass (the guide's HBM
#      section: FETCH_SIZE and WRITE_SIZE cannot share a pass); --pmc is combined with --kernel-trace only.
#   4. FETCH_SIZE / WRITE_SIZE of the LM-head stage (scripts/head_stage_probe.py), kept logits vs row chunks.
# usage: bash scripts/profile_round.sh r2
set -o pipefail
R=${1:-r2}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
echo "[profile] kernel trace"; date
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || echo "kernel trace run failed"
echo "[profile] hip api trace"; date
timeout -k 10 400 rocprofv3 --hip-trace --stats --output-format csv -d $OUT/hip -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_hiptrace.json 2> $OUT/hip.err || echo "hip trace run failed"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  echo "[profile] pmc $grp"; date
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc/$tag -o attn -- python3 scripts/attn_bench.py 3 > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
done
# 4. the LM-head stage, kept logits vs cache-sized row chunks (f1 experiment): fabric-side bytes of every kernel of the stage
if [ "${HEAD_PMC:-1}" = "1" ]; then
  for var in kept chunk512 chunk4096; do for cnt in FETCH_SIZE WRITE_SIZE; do
    echo "[profile] head stage $var $cnt"; date
    ONLY=$var timeout -k 10 200 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $OUT/head/${var}__$cnt -o head -- python3 scripts/head_stage_probe.py > $OUT/head_${var}_$cnt.log 2>&1 || echo "head pass $var $cnt failed"
  done; done
  python3 scripts/pmc_total.py $OUT/head > $OUT/head_stage_traffic.json 2> $OUT/head_total.err || true
fi
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_attn.json 2> $OUT/pmc_summary.err || true
python3 scripts/prof_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err || true
ls -la $OUT; date
