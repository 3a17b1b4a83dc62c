#!/bin/bash
# Round-3 evidence, collected on the GPU box into gpurun_out/final_r03/ (tools/install_profiles_r03.py copies it into profiles/).
# rocprofv3 runs from /tmp (TMPDIR=/tmp) with the program itself after `--`; PMC passes are separate runs (--kernel-trace only).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/final_r03"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
say() { echo "[collect $(date +%H:%M:%S)] $*"; }

say "bench lines: fp32 default, fp16, fp16-strict, kitti streams"
python3 "$ROOT/bench.py" > "$OUT/bench_b16.json" 2> "$OUT/bench_b16.stderr.log"
python3 "$ROOT/bench.py" --precision fp16 --no-cpu-baseline > "$OUT/f16_bench_b16.json" 2> "$OUT/f16_bench_b16.stderr.log"
python3 "$ROOT/bench.py" --precision fp16-strict --no-cpu-baseline > "$OUT/f16s_bench_b16.json" 2> "$OUT/f16s_bench_b16.stderr.log"
python3 "$ROOT/bench.py" --workload kitti > "$OUT/kitti_bench.json" 2> /dev/null
python3 "$ROOT/bench.py" --workload kitti --precision fp16-strict > "$OUT/kitti_bench_strict.json" 2> /dev/null
python3 "$ROOT/bench.py" --workload kitti --precision fp32 > "$OUT/kitti_bench_fp32.json" 2> /dev/null
for b in 1 4 32; do
  python3 "$ROOT/bench.py" --batch $b --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_b$b.json" 2> /dev/null
  python3 "$ROOT/bench.py" --precision fp16 --batch $b --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/f16_bench_b$b.json" 2> /dev/null
done
say "fp32 with the F(4x4) route off (A/B of the same build)"
PWC_CONV_WINO4=0 python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_b16_wino4_off.json" 2> /dev/null

say "N > 1 rehearsal on the one GPU (self-launching bench.py, gloo, every rank on cuda:0: plumbing, not a measurement)"
PWC_BENCH_REHEARSE=1 python3 "$ROOT/bench.py" --gpus 2 --steps 5 --warmup 2 --batch 4 > "$OUT/rehearse_n2_fp32.json" 2> "$OUT/rehearse_n2_fp32.stderr.log"
PWC_BENCH_REHEARSE=1 python3 "$ROOT/bench.py" --gpus 2 --steps 5 --warmup 2 --workload kitti --batch 4 > "$OUT/rehearse_n2_kitti.json" 2> "$OUT/rehearse_n2_kitti.stderr.log"

say "kernel traces + timelines"
bash "$ROOT/tools/collect_timelines.sh" > /dev/null 2>&1
cp "$ROOT/gpurun_out/tl/timeline_fp32.txt" "$OUT/forward_timeline_b16.txt"
cp "$ROOT/gpurun_out/tl/timeline_fp16.txt" "$OUT/f16_forward_timeline_b16.txt"
cp "$ROOT/gpurun_out/tl/timeline_fp16-strict.txt" "$OUT/f16s_forward_timeline_b16.txt"
cp "$ROOT/gpurun_out/tl/kernel_stats_fp32.csv" "$OUT/kernel_stats_bench_b16.csv"
cp "$ROOT/gpurun_out/tl/kernel_stats_fp16.csv" "$OUT/f16_kernel_stats_bench_b16.csv"

say "dominant kernels alone (rocprofv3 averages)"
rocprofv3 --kernel-trace --stats -d "$OUT/p4" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
cp "$(find "$OUT/p4" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_wino4_dc_conv1.csv"; rm -rf "$OUT/p4"
python3 "$ROOT/tools/bench_wino4.py" all > "$OUT/microbench_wino4.txt" 2>&1
python3 "$ROOT/tools/bench_wino.py" layers > "$OUT/microbench_wino_layers.txt" 2>&1
python3 "$ROOT/tools/bench_warpcorr.py" > "$OUT/microbench_warpcorr.txt" 2>&1
PWC_BENCH_LEVELS=2 rocprofv3 --kernel-trace --stats -d "$OUT/p2" -o p --output-format csv -- python3 "$ROOT/tools/bench_warpcorr.py" > /dev/null 2>&1
cp "$(find "$OUT/p2" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats_warpcorr.csv"; rm -rf "$OUT/p2"
rocprofv3 --kernel-trace --stats -d "$OUT/p3" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv_f16.py" dc_conv1 > /dev/null 2>&1
cp "$(find "$OUT/p3" -name "*kernel_stats.csv" | head -1)" "$OUT/f16_kernel_stats_dc_conv1.csv"; rm -rf "$OUT/p3"
python3 "$ROOT/tools/bench_conv_f16.py" > "$OUT/f16_microbench_conv.txt" 2>&1
PWC_BENCH_SPLIT=1 python3 "$ROOT/tools/bench_conv_f16.py" > "$OUT/f16_microbench_split.txt" 2>&1
python3 "$ROOT/tools/bench_corr.py" > "$OUT/microbench_corr.txt" 2>&1
python3 "$ROOT/tools/bench_bwd.py" > "$OUT/microbench_bwd.txt" 2>&1

say "KITTI stream: share of the GPU time in kernels that are not ours (PyTorch pre/post launches inside the captured graph)"
for p in fp16 fp32; do
  rocprofv3 --kernel-trace --stats -d "$OUT/ks" -o p --output-format csv -- python3 "$ROOT/bench.py" --workload kitti --precision $p --steps 20 --warmup 5 > /dev/null 2>&1
  python3 "$ROOT/tools/kernel_share.py" "$(find "$OUT/ks" -name "*kernel_stats.csv" | head -1)" > "$OUT/kitti_share_$p.txt"; rm -rf "$OUT/ks"
done

say "PMC passes (one counter per run): calibration, dc_conv1 F(4x4) / fp16, fused warp+corr, plain corr"
: > "$OUT/pmc_summary.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/calib_fetch.py" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "calib_dma_read_kernel<4>" $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "calib_dma_read_kernel<16>" $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino4 $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino8r $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
  PWC_BENCH_F16_ONLY=1 rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv_f16.py" dc_conv1 > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_f16 $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
  PWC_BENCH_LEVELS=2 rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_warpcorr.py" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "corr81_dma_kernel<true>" $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "corr81_dma_kernel<false>" $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_wino4.py" pmc > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino4 $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_wino8r $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
done
# fp16 dc_conv1: matrix-pipe busy cycles and the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration), MI355X guide "DVFS give-back"
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA; do
  PWC_BENCH_F16_ONLY=1 rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_conv_f16.py" dc_conv1 > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" conv3x3_f16 $c >> "$OUT/pmc_summary.txt"
  [ "$c" = GRBM_GUI_ACTIVE ] && cp "$(find "$OUT/q" -name "*kernel_trace.csv" | head -1)" "$OUT/f16_grbm_trace.csv"
  rm -rf "$OUT/q"
done
for c in SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM; do
  PWC_BENCH_LEVELS=2 rocprofv3 --kernel-trace --pmc $c -d "$OUT/q" -o p --output-format csv -- python3 "$ROOT/tools/bench_warpcorr.py" > /dev/null 2>&1
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "corr81_dma_kernel<true>" $c >> "$OUT/pmc_summary.txt"
  python3 "$ROOT/tools/pmc_avg.py" "$OUT/q" "corr81_dma_kernel<false>" $c >> "$OUT/pmc_summary.txt"; rm -rf "$OUT/q"
done
say "done"
cat "$OUT/pmc_summary.txt"
ls "$OUT"
