#!/bin/bash
# Round-4 evidence, collected on the GPU box into gpurun_out/final_r04/ (tools/install_profiles_r04.py copies it into profiles/).
# rocprofv3 runs from /tmp (TMPDIR=/tmp) with the program itself after `--`; PMC passes are separate runs (--kernel-trace only).
# PWC_COLLECT=part1|part2|all: two gpurun calls keep each under the call's time limit.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/final_r04"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
say() { echo "[collect $(date +%H:%M:%S)] $*"; }
PART="${PWC_COLLECT:-all}"

if [ "$PART" = part1 ] || [ "$PART" = all ]; then
say "bench lines: fp32 default, fp16-strict, fp16, kitti streams (strict = the default), batch sweep"
python3 "$ROOT/bench.py" > "$OUT/bench_b16.json" 2> "$OUT/bench_b16.stderr.log"
python3 "$ROOT/bench.py" --precision fp16-strict --no-cpu-baseline > "$OUT/f16s_bench_b16.json" 2> "$OUT/f16s_bench_b16.stderr.log"
python3 "$ROOT/bench.py" --precision fp16 --no-cpu-baseline > "$OUT/f16_bench_b16.json" 2> "$OUT/f16_bench_b16.stderr.log"
python3 "$ROOT/bench.py" --workload kitti > "$OUT/kitti_bench_strict.json" 2> /dev/null
python3 "$ROOT/bench.py" --workload kitti --precision fp16 > "$OUT/kitti_bench.json" 2> /dev/null
python3 "$ROOT/bench.py" --workload kitti --precision fp32 > "$OUT/kitti_bench_fp32.json" 2> /dev/null
for b in 1 2 4 8 32; do
  python3 "$ROOT/bench.py" --batch $b --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench_b$b.json" 2> /dev/null
done
say "A/B in one process: whole-launch split of small F(4x4) launches off / on"
python3 "$ROOT/tools/bench_ab_option.py" w4_smallsplit 0 1 1,2,4,8,16 > "$OUT/ab_smallsplit.txt" 2>&1
say "N > 1 rehearsal on the one GPU (self-launching bench.py, gloo, every rank on cuda:0: plumbing, not a measurement)"
PWC_BENCH_REHEARSE=1 python3 "$ROOT/bench.py" --gpus 2 --steps 5 --warmup 2 --batch 4 > "$OUT/rehearse_n2_fp32.json" 2> "$OUT/rehearse_n2_fp32.stderr.log"
PWC_BENCH_REHEARSE=1 python3 "$ROOT/bench.py" --gpus 2 --steps 5 --warmup 2 --workload kitti --batch 4 > "$OUT/rehearse_n2_kitti.json" 2> "$OUT/rehearse_n2_kitti.stderr.log"
say "per-launch timelines: ONE forward each, cut by periodicity out of a trace that holds nothing but forwards of that plan"
mkdir -p "$ROOT/gpurun_out/tl"
for pb in "fp32 16" "fp32 1" "fp16 16" "fp16-strict 16"; do
  set -- $pb
  "$ROOT/tools/collect_timeline_one.sh" "$1" "$2" "$OUT/forward_timeline_${1}_b$2.txt"
done
say "correlation kernels alone: round-2 and round-4 kernels on the forward's own operands, HIP events and rocprofv3 averages"
PWC_BENCH_LEVELS=2,3 python3 "$ROOT/tools/bench_corr_pipe.py" time plan > "$OUT/microbench_corr_pipe.txt" 2>&1
PWC_BENCH_LEVELS=2 rocprofv3 --kernel-trace --stats -d "$OUT/p2" -o p --output-format csv -- python3 "$ROOT/tools/bench_corr_pipe.py" time plan > /dev/null 2>&1
cp "$(find "$OUT/p2" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_warpcorr.csv"; rm -rf "$OUT/p2"
python3 "$ROOT/tools/bench_warpcorr.py" > "$OUT/microbench_warpcorr.txt" 2>&1
make -s -C "$ROOT/tools/experiments/ubench" > /dev/null 2>&1
"$ROOT/tools/experiments/ubench/rw_mix" > "$OUT/ubench_rw_mix.txt" 2>&1
"$ROOT/tools/experiments/ubench/valu_rate" > "$OUT/ubench_valu_rate.txt" 2>&1
fi

if [ "$PART" = part2 ] || [ "$PART" = all ]; then
say "dominant conv kernel alone (rocprofv3 averages) + layer table"
rocprofv3 --kernel-trace --stats -d "$OUT/p4" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
cp "$(find "$OUT/p4" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_wino4_dc_conv1.csv"; rm -rf "$OUT/p4"
python3 "$ROOT/tools/bench_wino4.py" all > "$OUT/microbench_wino4.txt" 2>&1
say "PMC passes (one counter per run): calibration, dc_conv1 F(4x4), correlation kernels"
: > "$OUT/pmc_summary.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/calib_fetch.py" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "calib_dma_read_kernel<4>" $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "calib_dma_read_kernel<16>" $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino4 $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino4 $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
done
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT; do
  PWC_BENCH_LEVELS=2 rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_corr_pipe.py" time plan > /dev/null 2>&1
  for k in "warp_corr81_pipe_kernel<8>" "corr81_dma_kernel<true>" "corr81_roll_kernel" "corr81_dma_kernel<false>"; do
    python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "$k" $c >> "$OUT/pmc_summary.txt"
  done
  rm -rf "$OUT/q"
done
fi
say "done"
[ -f "$OUT/pmc_summary.txt" ] && cat "$OUT/pmc_summary.txt"
ls "$OUT"
