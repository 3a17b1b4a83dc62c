#!/bin/bash
# PMC diagnosis of the Winograd conv on dc_conv1: wave-cycle split, matrix-pipe busy cycles, GPU cycles (clock).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_wino"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d "$OUT/p_$tag" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino.py" pmc > "$OUT/run_$tag.log" 2>&1
  for c in $set; do python3 "$ROOT/tools/pmc_avg.py" "$OUT/p_$tag" conv3x3_wino $c; python3 "$ROOT/tools/pmc_avg.py" "$OUT/p_$tag" conv3x3_mfma $c; done
  rm -rf "$OUT/p_$tag"
done
