#!/bin/bash
# per-launch timelines of one forward (fp32, fp16 and fp16-strict plans, batch 16) into gpurun_out/tl/
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/tl"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for prec in ${PWC_TL_PRECS:-fp16 fp16-strict fp32}; do
  rocprofv3 --kernel-trace --stats -d "$OUT/prof_$prec" -o p --output-format csv -- python3 "$ROOT/bench.py" --precision $prec --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_$prec.log" 2>&1
  f=$(find "$OUT/prof_$prec" -name "*kernel_trace.csv" | head -1)
  if [ "$prec" = "fp16-strict" ]; then python3 "$ROOT/tools/timeline.py" "$f" --full --fp32 --strict > "$OUT/timeline_$prec.txt"
  elif [ "$prec" = "fp32" ]; then python3 "$ROOT/tools/timeline.py" "$f" --full --fp32 > "$OUT/timeline_$prec.txt"; else python3 "$ROOT/tools/timeline.py" "$f" --full > "$OUT/timeline_$prec.txt"; fi
  cp "$(find "$OUT/prof_$prec" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_$prec.csv"
  rm -rf "$OUT/prof_$prec"
done
