#!/bin/bash
# Collect the end-of-round evidence on the GPU box into gpurun_out/final/ (copied into profiles/ afterwards).
#   tools/collect_profiles.sh            -> bench lines, rocprofv3 kernel stats, timelines, PMC counters
# rocprofv3 runs from /tmp (TMPDIR=/tmp) with the program itself after `--`; PMC passes are separate runs.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/final"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
say() { echo "[collect $(date +%H:%M:%S)] $*"; }

say "bench default (batch 16, with cpu baseline)"
python3 "$ROOT/bench.py" > "$OUT/bench_b16.json" 2> "$OUT/bench_b16.stderr.log"
say "bench batch sweep"
for b in 1 4 32; do
  python3 "$ROOT/bench.py" --batch $b --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_b$b.json" 2> /dev/null
done
python3 "$ROOT/bench.py" --conv-backend torch --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_b16_miopen_convs.json" 2> /dev/null
python3 "$ROOT/tools/bench_video.py" 448 1024 1 16 > "$OUT/bench_video.txt" 2>&1

say "kernel trace + stats, batch 16"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_b16" -o b16 --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_b16.log" 2>&1
f=$(find "$OUT/prof_b16" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$f" --full --fp32 > "$OUT/forward_timeline_b16.txt"
cp "$(find "$OUT/prof_b16" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_bench_b16.csv"
rm -rf "$OUT/prof_b16"

say "kernel trace, batch 1"
rocprofv3 --kernel-trace -d "$OUT/prof_b1" -o b1 --output-format csv -- python3 "$ROOT/bench.py" --batch 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_b1.log" 2>&1
f=$(find "$OUT/prof_b1" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$f" --full --fp32 --min-grid=50000 > "$OUT/forward_timeline_b1.txt"
rm -rf "$OUT/prof_b1"

say "kernel stats of the dominant kernel alone (dc_conv1) and of the level-2 correlation"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_dc1" -o dc1 --output-format csv -- python3 "$ROOT/tools/bench_conv.py" dc_conv1 > "$OUT/microbench_dc_conv1.txt" 2>&1
cp "$(find "$OUT/prof_dc1" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_dc_conv1.csv"
rm -rf "$OUT/prof_dc1"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_corr" -o corr --output-format csv -- python3 "$ROOT/tools/bench_corr.py" > "$OUT/microbench_corr.txt" 2>&1
cp "$(find "$OUT/prof_corr" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_corr.csv"
rm -rf "$OUT/prof_corr"

say "fp16 path: bench lines, kernel stats, timeline, PMC of the dominant kernel"
python3 "$ROOT/bench.py" --precision fp16 > "$OUT/f16_bench_b16.json" 2> "$OUT/f16_bench_b16.stderr.log"
for b in 1 32; do
  python3 "$ROOT/bench.py" --precision fp16 --batch $b --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/f16_bench_b$b.json" 2> /dev/null
done
rocprofv3 --kernel-trace --stats -d "$OUT/prof_f16" -o f16 --output-format csv -- python3 "$ROOT/bench.py" --precision fp16 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_f16.log" 2>&1
f=$(find "$OUT/prof_f16" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$f" --full > "$OUT/f16_forward_timeline_b16.txt"
cp "$(find "$OUT/prof_f16" -name "*kernel_stats.csv" | head -1)" "$OUT/f16_kernel_stats_bench_b16.csv"
rm -rf "$OUT/prof_f16"
python3 "$ROOT/tools/bench_conv_f16.py" > "$OUT/f16_microbench_conv.txt" 2>&1
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_f16_$c" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv_f16.py" dc_conv1 > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/pmc_f16_$c" conv3x3_f16 $c >> "$OUT/f16_pmc_summary.txt"
  rm -rf "$OUT/pmc_f16_$c"
done

say "PMC passes (one counter per run)"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_dc1_$c" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv.py" dc_conv1 > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/pmc_dc1_$c" conv3x3_mfma_kernel $c >> "$OUT/pmc_summary.txt"
  rm -rf "$OUT/pmc_dc1_$c"
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_corr_$c" -o p --output-format csv -- python3 "$ROOT/tools/bench_corr.py" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/pmc_corr_$c" corr81_dma_kernel $c >> "$OUT/pmc_summary.txt"
  rm -rf "$OUT/pmc_corr_$c"
done
say "done"
ls -la "$OUT"
