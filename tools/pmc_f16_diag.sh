#!/bin/bash
# PMC diagnosis of the fp16 MFMA conv on dc_conv1 (where do the waves spend their cycles?)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_diag"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1
export PWC_BENCH_F16_ONLY=1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/p_$tag" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv_f16.py" ${1:-dc_conv1} > "$OUT/run_$tag.log" 2>&1
  for c in $set; do python3 "$ROOT/tools/pmc_avg.py" "$OUT/p_$tag" conv3x3_f16 $c; done
  rm -rf "$OUT/p_$tag"
done
