#!/usr/bin/env python3
"""Copy the evidence tools/collect_profiles_r04.sh wrote (default gpurun_out/final_r04) into profiles/r04_* and derive
profiles/r04_pmc_traffic.json (what bench.py reports as roofline.traffic / frac_rocprof) and the batch-sweep table.
usage: tools/install_profiles_r04.py [srcdir]"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "final_r04")
DST = os.path.join(ROOT, "profiles")

COPIES = {
    "bench_b16.json": "r04_bench_b16.json", "bench_b16.stderr.log": "r04_bench_b16.stderr.log",
    "f16_bench_b16.json": "r04_f16_bench_b16.json", "f16_bench_b16.stderr.log": "r04_f16_bench_b16.stderr.log",
    "f16s_bench_b16.json": "r04_f16strict_bench_b16.json", "f16s_bench_b16.stderr.log": "r04_f16strict_bench_b16.stderr.log",
    "kitti_bench.json": "r04_kitti_stream_fp16.json", "kitti_bench_strict.json": "r04_kitti_stream_fp16strict.json", "kitti_bench_fp32.json": "r04_kitti_stream_fp32.json",
    "rehearse_n2_fp32.json": "r04_rehearse_n2_fp32.json", "rehearse_n2_kitti.json": "r04_rehearse_n2_kitti.json",
    "forward_timeline_fp32_b16.txt": "r04_forward_timeline_b16.txt", "forward_timeline_fp32_b1.txt": "r04_forward_timeline_b1.txt",
    "forward_timeline_fp16_b16.txt": "r04_f16_forward_timeline_b16.txt", "forward_timeline_fp16-strict_b16.txt": "r04_f16strict_forward_timeline_b16.txt",
    "forward_timeline_fp32_b16_kernel_stats.csv": "r04_kernel_stats_forward_b16.csv", "forward_timeline_fp16_b16_kernel_stats.csv": "r04_f16_kernel_stats_forward_b16.csv",
    "kernel_stats_wino4_dc_conv1.csv": "r04_kernel_stats_wino4_dc_conv1_alone.csv", "kernel_stats_warpcorr.csv": "r04_kernel_stats_warpcorr_alone.csv",
    "pmc_summary.txt": "r04_pmc_summary.txt", "ab_smallsplit.txt": "r04_ab_smallsplit.txt", "microbench_corr_pipe.txt": "r04_microbench_corr_pipe.txt",
    "ubench_rw_mix.txt": "r04_ubench_rw_mix.txt", "ubench_valu_rate.txt": "r04_ubench_valu_rate.txt",
}
FILTERED = {"microbench_warpcorr.txt": ("r04_microbench_warpcorr.txt", ("level",)),
            "microbench_wino4.txt": ("r04_wino4_layers.txt", ("B", "conv", "dc_", "sum"))}


def pmc(summary, kernel, counter):
    m = re.search(r"^%s %s: launches \d+, mean ([0-9.]+)" % (re.escape(kernel), counter), summary, re.M)
    return float(m.group(1)) if m else None


def avg_ms(stats_csv, needle):
    """average duration (ms) of the kernel whose name contains `needle` from a rocprofv3 --stats CSV"""
    try:
        for r in csv.DictReader(open(os.path.join(SRC, stats_csv))):
            if needle in r["Name"]:
                return float(r["AverageNs"]) / 1e6
    except (OSError, KeyError):
        pass
    return None


def main():
    for s, d in COPIES.items():
        if os.path.exists(os.path.join(SRC, s)):
            shutil.copyfile(os.path.join(SRC, s), os.path.join(DST, d))
    for s, (d, keep) in FILTERED.items():
        if os.path.exists(os.path.join(SRC, s)):
            with open(os.path.join(SRC, s)) as f, open(os.path.join(DST, d), "w") as o:
                o.writelines(line for line in f if line.startswith(keep))
    try:
        summary = open(os.path.join(SRC, "pmc_summary.txt")).read()
    except OSError:
        summary = ""        # part 2 (PMC passes) not collected yet: the traffic records are left out
    kib = 1024.0
    known = 1 << 30                                                     # bytes tools/calib_fetch.py streams per launch
    cal = {}
    for w, k in ((4, "calib_dma_read_kernel<4>"), (16, "calib_dma_read_kernel<16>")):
        v = pmc(summary, k, "FETCH_SIZE")
        cal[w] = (known / (v * kib)) if v else None
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) on one MI355X, per-launch means "
                   "(tools/collect_profiles_r04.sh, tools/pmc_avg.py; counters in KiB).  FETCH_SIZE is multiplied by the factor measured on a "
                   "byte-exact 1 GiB stream through the SAME LDS-DMA instruction (tools/calib_fetch.py -> pwc_calib_lds_dma_read): "
                   "known bytes / reported bytes, per access width.",
           "fetch_size_correction": {"buffer_load_dword_lds": cal[4], "buffer_load_dwordx4_lds": cal[16],
                                     "_note": "MI355X guide: FETCH_SIZE reports 1/2 of a wide coalesced streaming read on gfx950; "
                                              "round 1 ASSUMED the same for dword LDS-DMA, this is the measurement"}}
    f4 = cal[4] or 2.0
    f16 = cal[16] or 2.0
    alg_in = 16 * 565 * 112 * 256 * 4
    alg_out = 16 * 128 * 112 * 256 * 4
    rows = (("conv3x3_wino4_dc_conv1_b16", "conv3x3_wino4", f16, "kernel_stats_wino4_dc_conv1.csv", "conv3x3_wino4", alg_in + alg_out + 565 * 128 * 36 * 4,
             "fp32 dc_conv1 565->128 @112x256 B=16 by Winograd F(4x4,3x3): input and G g Gt filters (10.4 MB, re-read per workgroup from "
             "L2) by 16-byte LDS-DMA; algorithmic = input + filters once + output"),
            ("conv3x3_wino_dc_conv1_b16", "conv3x3_wino8r", f4, "kernel_stats_wino4_dc_conv1.csv", "conv3x3_wino8r", alg_in + alg_out + 565 * 128 * 16 * 4,
             "the same layer by Winograd F(2x2,3x3) (the warm-up launches of tools/bench_wino4.py pmc): input by dword LDS-DMA"),
            ("warp_corr81_level2_b16", "warp_corr81_pipe_kernel<8>", f16, "kernel_stats_warpcorr.csv", "warp_corr81_pipe_kernel<8>", 269746176,
             "fused warp + correlation, level 2, round-4 window kernel on the forward's own up_flow (in1 and the 23x52 source window by 16-byte "
             "LDS-DMA, outside pixels by 4-byte gathers: x2 correction applied to all reads, an upper bound)"),
            ("warp_corr81_round2_level2_b16", "corr81_dma_kernel<true>", f16, "kernel_stats_warpcorr.csv", "corr81_dma_kernel<true>", 269746176,
             "the round-2 fused kernel on the same operands (option warpcorr_window=0)"),
            ("corr81_level2_b16", "corr81_dma_kernel<false>", f16, "kernel_stats_warpcorr.csv", "corr81_dma_kernel<false>", 266076160,
             "correlation alone, level 2 (round-2 kernel: the forward's default)"),
            ("corr81_roll_level2_b16", "corr81_roll_kernel", f16, "kernel_stats_warpcorr.csv", "corr81_roll_kernel", 266076160,
             "correlation alone, level 2, round-4 rolling-window kernel (option corr_pipe=1)"))
    for key, kern, fac, stats, needle, alg, note in rows:
        fe, wr = pmc(summary, kern, "FETCH_SIZE"), pmc(summary, kern, "WRITE_SIZE")
        if fe is None or wr is None:
            continue
        rec = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "fetch_correction": fac, "traffic_bytes": int(fe * kib * fac + wr * kib),
               "algorithmic_bytes": alg, "source": "r04_pmc_summary.txt, FETCH_SIZE x %.3f (calibrated) + WRITE_SIZE" % fac, "_note": note}
        ms = avg_ms(stats, needle)
        if ms is not None:
            rec["rocprof_avg_ms"] = round(ms, 5)
        out[key] = rec
    try:    # the half-precision conv kernel did not change this round: its record is carried over, and says so
        old = json.load(open(os.path.join(DST, "r03_pmc_traffic.json")))["conv3x3_f16_dc_conv1_b16"]
        old["source"] = "carried over from profiles/r03_pmc_traffic.json (kernel unchanged in round 4): " + old.get("source", "")
        out["conv3x3_f16_dc_conv1_b16"] = old
    except (OSError, KeyError, ValueError):
        pass
    with open(os.path.join(DST, "r04_pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(DST, "r04_batch_sweep.md"), "w") as o:
        o.write("# bench.py on one MI355X, 1024x448, HIP graph (round 4, one box, one gpurun call: tools/collect_profiles_r04.sh)\n\n"
                "| run | image-pairs/s | ms/step | whole-forward MFMA util | dc_conv1 probe TF | warp+corr L2 probe GB/s |\n|---|---|---|---|---|---|\n")
        for tag, fn in (("fp32 batch 1", "bench_b1.json"), ("fp32 batch 2", "bench_b2.json"), ("fp32 batch 4", "bench_b4.json"), ("fp32 batch 8", "bench_b8.json"),
                        ("fp32 batch 16 (default run)", "bench_b16.json"), ("fp32 batch 32", "bench_b32.json"), ("fp16 batch 16", "f16_bench_b16.json")):
            try:
                d = json.load(open(os.path.join(SRC, fn)))
            except (OSError, ValueError):
                continue
            o.write("| %s | %.1f | %.3f | %.3f | %.1f | %.0f |\n" % (tag, d["value"], d["ms_per_step"], d["mfma_util_whole_forward"],
                                                                 d["roofline"]["achieved"], d["roofline_corr"]["achieved"]))
        for tag, fn in (("fp16-strict batch 16", "f16s_bench_b16.json"),):
            try:
                d = json.load(open(os.path.join(SRC, fn)))
                o.write("| %s | %.1f | %.3f | %s | %.1f | - |\n" % (tag, d["value"], d["ms_per_step"], d.get("mfma_util_whole_forward", "-"), d["roofline"]["achieved"]))
            except (OSError, ValueError, KeyError):
                pass
        for tag, fn in (("KITTI 375x1242 stream fp16, H2D included", "kitti_bench.json"), ("KITTI 375x1242 stream fp16-strict, H2D included", "kitti_bench_strict.json"),
                        ("KITTI 375x1242 stream fp32, H2D included", "kitti_bench_fp32.json")):
            try:
                d = json.load(open(os.path.join(SRC, fn)))
                o.write("| %s, %d pairs per replay | %.1f | %.3f | - | - | - |\n" % (tag, d["config"]["pairs_per_gpu"], d["value"], d["ms_per_step"]))
            except (OSError, ValueError):
                pass
        try:
            o.write("\nOne-process A/B of the whole-launch Cin split for small F(4x4) launches (tools/bench_ab_option.py w4_smallsplit 0 1):\n\n```\n%s```\n"
                    % open(os.path.join(SRC, "ab_smallsplit.txt")).read())
        except OSError:
            pass
    print(open(os.path.join(DST, "r04_batch_sweep.md")).read())
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
