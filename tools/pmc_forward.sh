#!/bin/bash
# LDS bank conflicts kernel by kernel over a whole bench.py run: one rocprofv3 --pmc pass per counter (from /tmp, program after `--`).
# usage: tools/pmc_forward.sh [fp32|fp16|fp16-strict]  -> gpurun_out/pmc_fwd_<prec>.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PREC=${1:-fp32}
OUT="$ROOT/gpurun_out/pmc_fwd"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in ${PWC_PMC_COUNTERS:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS}; do
  rm -rf "$OUT/$c"
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o p --output-format csv -- python3 "$ROOT/bench.py" --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 "$ROOT/tools/pmc_by_kernel.py" $(for c in ${PWC_PMC_COUNTERS:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS}; do echo "$OUT/$c"; done) > "$ROOT/gpurun_out/pmc_fwd_$PREC.txt"
rm -rf "$OUT"
cat "$ROOT/gpurun_out/pmc_fwd_$PREC.txt"
