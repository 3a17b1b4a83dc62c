#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-launch timeline of ONE forward of bench.py.

usage: tools/timeline.py <kernel_trace.csv> --periodic [--full]          (trace of tools/profile_forward.py: forwards of ONE plan only)
       tools/timeline.py <kernel_trace.csv> [--full] [--min-grid=N] [--fp32 [--strict]]      (trace of bench.py, heuristic cut)
--periodic: the trace ends in K identical forwards; the period is found from the kernel-name sequence itself (the shortest L such
that the last three stretches of L launches carry the same names and grids) and exactly the last full forward is reported --
whatever the precision, batch size or kernel selection.
--fp32: the trace also holds fp16 forwards (bench.py's fp16_same_workload leg); pick the fp32 plan's forward.
--strict (with --fp32): the fp16-strict plan, whose forward starts on the fp32 kernels and continues on the half-precision ones.
A forward starts at the two back-to-back launches of conv1a on the two images.
"""
import csv
import re
import sys


def short(n):
    m = re.search(r"conv3x3_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", n)
    if m:
        return "mfma<MT%s,NT%s,S%s,D%s,two%s>" % m.groups()[:5] + ("+splitK" if m.group(6) == "1" else "")
    m = re.search(r"conv3x3_mfma16_kernel<(\d+), (\d+)>", n)
    if m:
        return "mfma16<NT%s,CK%s>" % m.groups()
    m = (re.search(r"conv3x3_f16_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", n) or
         re.search(r"conv3x3_f16_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", n))
    if m:
        return "f16conv<MT%s,NT%s,S%s,D%s,R%s>" % m.groups()
    m = (re.search(r"conv3x3_f16w8_kernel<(\d+), (\d+)", n) or re.search(r"conv3x3_f16w8_kernelILi(\d+)ELi(\d+)E", n))
    if m:
        return "f16w8<MT%s,D%s>" % m.groups()
    m = re.search(r"conv3x3_wino4p?_kernel<(\d+), (\d+)(?:, (\d+))?>", n) or re.search(r"conv3x3_wino4p?_kernelILi(\d+)ELi(\d+)E(?:Li(\d+)E)?", n)
    if m:
        return "wino4<CB%s,TG%s%s>" % (m.group(1), m.group(2), ",GW32" if m.group(3) == "32" else "")
    m = re.search(r"conv3x3_wino(8r?)_kernel<(\d+)>", n) or re.search(r"conv3x3_wino(8r?)_kernelILi(\d+)E", n)
    if m:
        return "wino%s<MT%s>" % m.groups()
    m = re.search(r"stream3x3_kernel<(\d+), (\d+), (\d+)>", n)
    if m:
        return "stream3x3<mode%s,TH%s,KS%s>" % m.groups()
    if "corr81_dma_kernel<true>" in n or "corr81_dma_kernelILb1E" in n:
        return "warp+corr81"
    if "warp_corr81_pipe_kernel" in n:
        return "warp+corr81 (window)"
    if "corr81_roll_kernel" in n or "corr81_pipe_kernel" in n:
        return "corr81 (pipelined)"
    if "wino4_tail_reduce" in n:
        return "wino4_tail_reduce"
    if "kitti_ingest" in n or "flow_upsample" in n:
        return "kitti_pre/post"
    for k in ("pyr1_fused", "corr81_bwd", "corr81_c8", "warp_c8", "nchw_to_c8_hilo", "nchw_to_c8", "c8_to_nchw", "image_conv_s2_f32", "image_conv_s2"):
        if k in n:
            return k
    for k in ("stream3x3_kernel<1>", "stream3x3_kernel<2>", "stream3x3_kernel<3>", "conv3x3_head", "deconv4x4s2", "corr81", "corr_generic", "splitk_reduce", "warp_kernel", "copyBuffer", "elementwise", "pack3x3", "lattice_unsplit"):
        if k in n:
            return k
    return n[:30]


def report(rows, names, s, e):
    t0 = int(rows[s]["Start_Timestamp"])
    agg = {}
    tot = 0.0
    for r, n in zip(rows[s:e], names[s:e]):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
        if "--full" in sys.argv:
            print("%8.1f us  +%8.1f  %-22s grid=(%d,%s)" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, n,
                                                          int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]))
    span = ((int(rows[e]["Start_Timestamp"]) if e < len(rows) else int(rows[e - 1]["End_Timestamp"])) - t0) / 1e3
    print("one forward: %d launches, kernel time %.1f us, span %.1f us" % (e - s, tot, span))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-24s x%3d  %9.1f us  %5.1f %%" % (n, c, d, 100 * d / tot))


def periodic(rows, names):
    key = [(n, r["Grid_Size_X"], r["Workgroup_Size_X"]) for n, r in zip(names, rows)]
    N = len(key)
    for L in range(8, N // 3 + 1):
        if key[N - L:] == key[N - 2 * L:N - L] == key[N - 3 * L:N - 2 * L]:
            # the LAST forward is followed by nothing: report the one before it, whose span ends at the next forward's first launch
            report(rows, names, N - 2 * L, N - L)
            return True
    print("no period found in %d launches" % N)
    return False


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if "--periodic" in sys.argv:
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        periodic(rows, [short(r["Kernel_Name"]) for r in rows])
        return
    min_grid = 256 * 1000
    for a in sys.argv[2:]:
        if a.startswith("--min-grid="):
            min_grid = int(a.split("=")[1])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    first16 = "pyr1_fused" if any(n == "pyr1_fused" for n in names) else "image_conv_s2"
    if "--fp32" not in sys.argv and any(n == first16 for n in names):   # fp16 plan: a forward starts with the first layer on the two images
        starts = [i for i in range(len(rows) - 1) if names[i] == first16 and names[i + 1] == first16
                  and int(rows[i]["Grid_Size_X"]) > min_grid]
        rows_ok = True
    else:
        rows_ok = False
    starts = starts if rows_ok else [i for i in range(len(rows) - 1)
              if ((names[i].startswith("mfma<") and ("S2,D1" in names[i])) or names[i] == "image_conv_s2_f32") and names[i + 1] == names[i]
              and rows[i]["Grid_Size_X"] == rows[i + 1]["Grid_Size_X"] and int(rows[i]["Grid_Size_X"]) > min_grid]
    if len(starts) < 3:
        print("no forward found")
        return
    s, e = starts[-2], starts[-1]
    if "--fp32" in sys.argv:   # the last back-to-back pair of fp32 forwards with nothing else between them
        def span(i):
            return int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"])
        pure = [i for i in range(len(starts) - 1)
                if "--strict" in sys.argv or not any(n.startswith("f16conv") or n.startswith("f16w8") or n in ("image_conv_s2", "pyr1_fused") for n in names[starts[i]:starts[i + 1]])]
        best = min(span(i) for i in pure)
        i = [i for i in pure if span(i) <= 1.2 * best][-1]
        s, e = starts[i], starts[i + 1]
    t0 = int(rows[s]["Start_Timestamp"])
    agg = {}
    tot = 0.0
    for r, n in zip(rows[s:e], names[s:e]):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
        if "--full" in sys.argv:
            print("%8.1f us  +%8.1f  %-22s grid=(%d,%s)" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, n,
                                                          int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]))
    span = (int(rows[e]["Start_Timestamp"]) - t0) / 1e3
    print("one forward: %d launches, kernel time %.1f us, span %.1f us" % (e - s, tot, span))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-24s x%3d  %9.1f us  %5.1f %%" % (n, c, d, 100 * d / tot))


if __name__ == "__main__":
    main()
