#!/bin/bash
# fp32 forward timelines only (batch 16 and batch 1): tools/collect_timelines.sh [outdir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/tl}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/prof_b16" -o b16 --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_b16.log" 2>&1
python3 "$ROOT/tools/timeline.py" "$(find "$OUT/prof_b16" -name "*kernel_trace.csv" | head -1)" --full --fp32 > "$OUT/forward_timeline_b16.txt"
cp "$(find "$OUT/prof_b16" -name "*kernel_trace.csv" | head -1)" "$OUT/trace_b16.csv"
rm -rf "$OUT/prof_b16"
rocprofv3 --kernel-trace -d "$OUT/prof_b1" -o b1 --output-format csv -- python3 "$ROOT/bench.py" --batch 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_b1.log" 2>&1
python3 "$ROOT/tools/timeline.py" "$(find "$OUT/prof_b1" -name "*kernel_trace.csv" | head -1)" --full --fp32 --min-grid=50000 > "$OUT/forward_timeline_b1.txt"
cp "$(find "$OUT/prof_b1" -name "*kernel_trace.csv" | head -1)" "$OUT/trace_b1.csv"
rm -rf "$OUT/prof_b1"
grep -A8 "one forward" "$OUT/forward_timeline_b16.txt" "$OUT/forward_timeline_b1.txt"
