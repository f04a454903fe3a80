#!/bin/bash
# PMC counters of the attention kernels with a diagnostic library (DTA_LIB), one counter group per pass.
set -o pipefail
OUT=gpurun_out/pmc_dkv3
mkdir -p $OUT
export TMPDIR=/tmp
export DTA_LIB=$PWD/build/${1:-libdta_dkv3.so}     # built from scripts/diag/tree_attn_dkv3_experiment.hip (see its header)
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '+' | cut -c1-40)
  echo "[pmc] $grp"; date
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc/$tag -o attn -- python3 scripts/attn_bench.py 3 > $OUT/pmc_$tag.log 2>&1 || echo "pmc pass $tag failed"
done
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_attn.json 2> $OUT/pmc_summary.err || true
python3 -c "
import json; d=json.load(open('$OUT/pmc_attn.json'))
for k,v in d.items():
    if 'dkv' in k and 'final' not in k: print(k, json.dumps(v, indent=0))
"
