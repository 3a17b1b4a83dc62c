#!/usr/bin/env python3
"""Micro-bench of the fp16 MFMA conv (c8 layout) on single layers with HIP events; fp32 kernel beside it.
Default geometry: BASELINE level 2 (B=16, 112x256); PWC_BENCH_GEOM="B,H,W" overrides; cases `name:cin:cout:dil`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops, ops_f16  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = [int(v) for v in os.environ.get("PWC_BENCH_GEOM", "16,112,256").split(",")]
cases = [("dc_conv1", 565, 128, 1), ("conv2_0", 117, 128, 1), ("conv2_2", 373, 96, 1), ("conv2_3", 469, 64, 1),
         ("conv2_4", 533, 32, 1), ("dc_conv2", 128, 128, 2), ("dc_conv3", 128, 128, 4)]
if len(sys.argv) > 1:
    custom = [a for a in sys.argv[1:] if ":" in a]
    cases = [c for c in cases if c[0] in sys.argv[1:]] + [(n, int(ci), int(co), int(d)) for n, ci, co, d in
                                                          (a.split(":") for a in custom)]


def t(fn, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


g = torch.Generator().manual_seed(0)
for name, cin, cout, dil in cases:
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    fl = 2.0 * cout * cin * 9 * H * W * B
    xc = ops_f16.to_c8(x)
    wp16 = ops_f16.pack_conv3x3_f16(w)
    yc = torch.empty(ops_f16.c8_shape(B, cout, H, W), dtype=torch.float16, device=dev)
    ms16 = t(lambda: ops_f16.conv3x3_f16(xc, wp16, b, cin, cout, dilation=dil, out=yc))
    if os.environ.get("PWC_BENCH_SPLIT") == "1":           # split (hi + lo) filters of the strict mode beside the plain kernel
        wps = ops_f16.pack_conv3x3_f16(w, split=True)
        mss = t(lambda: ops_f16.conv3x3_f16(xc, wps, b, cin, cout, dilation=dil, out=yc, split_w=True))
        print("%-9s %4d->%3d d%-2d  fp16 %8.1f us   split filters %8.1f us (x%.2f, %.1f TFLOP/s executed)"
              % (name, cin, cout, dil, ms16 * 1e3, mss * 1e3, mss / ms16, 2 * fl / mss / 1e9), flush=True)
        continue
    if os.environ.get("PWC_BENCH_F16_ONLY") == "1":        # tile sweeps: skip the fp32 comparison
        print("%-9s %4d->%3d d%-2d  fp16 %8.1f us %7.1f TFLOP/s" % (name, cin, cout, dil, ms16 * 1e3, fl / ms16 / 1e9), flush=True)
        continue
    wp32 = ops.pack_conv3x3(w)
    y = torch.empty(B, cout, H, W, device=dev)
    ms32 = t(lambda: ops.conv3x3(x, wp32, b, cout, dilation=dil, out=y), reps=5)
    print("%-9s %4d->%3d d%-2d  fp16 %8.1f us %7.1f TFLOP/s (%.1f%% of 2500)   fp32 %8.1f us %6.1f TFLOP/s   x%.2f"
          % (name, cin, cout, dil, ms16 * 1e3, fl / ms16 / 1e9, fl / ms16 / 1e9 / 25.0, ms32 * 1e3, fl / ms32 / 1e9,
             ms32 / ms16), flush=True)
