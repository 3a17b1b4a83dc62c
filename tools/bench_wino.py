#!/usr/bin/env python3
"""Winograd F(2x2,3x3) convolution against the direct MFMA implicit GEMM: max error vs an fp64 reference on small cases,
then level-2 layer timings at batch 16 (HIP events).  usage: bench_wino.py [check|time|all]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator().manual_seed(0)


def case(B, cin, cout, H, W, leaky=0.1):
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if leaky is not None:
        ref = F.leaky_relu(ref, leaky)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    y = ops.conv3x3_wino(xd, ops.pack_conv3x3_wino(wd), bd, cout, leaky_slope=leaky)
    y0 = ops.conv3x3(xd, ops.pack_conv3x3(wd), bd, cout, leaky_slope=leaky)
    torch.cuda.synchronize()
    e = (y.cpu().double() - ref).abs().max().item()
    e0 = (y0.cpu().double() - ref).abs().max().item()
    print("B%d %3d->%3d %3dx%3d: wino max err %.3e   direct %.3e   (|ref| max %.2f)" % (B, cin, cout, H, W, e, e0, ref.abs().max().item()), flush=True)
    return e


if mode in ("check", "all"):
    for shp in ((1, 4, 32, 4, 32), (1, 8, 32, 8, 32), (2, 5, 7, 9, 13), (1, 16, 128, 12, 64), (2, 37, 96, 17, 70), (1, 64, 64, 20, 40), (1, 130, 128, 16, 33),
                (1, 21, 40, 33, 31)):
        case(*shp)
    case(1, 12, 32, 10, 36, leaky=None)

if mode == "pmc":          # one layer, few launches, for the counter passes of tools/pmc_wino.sh
    B, H, W, cin, cout = 16, 112, 256, 565, 128
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
    b = torch.zeros(cout, device=dev)
    up, wp = ops.pack_conv3x3_wino(w), ops.pack_conv3x3(w)
    y = torch.empty(B, cout, H, W, device=dev)
    for _ in range(40):        # the direct kernel first: ~160 ms that bring the clocks up (a different kernel name in the trace)
        ops.conv3x3(x, wp, b, cout, out=y)
    for _ in range(30):        # back to back, as in the forward's steady state: the profiler's average is over these
        ops.conv3x3_wino(x, up, b, cout, out=y)
    torch.cuda.synchronize()

if mode == "layers":       # every stride-1 / dilation-1 3x3 layer of the forward at batch 16 (pyramid: 32 images)
    def t2(fn, reps=8):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s_.record()
        for _ in range(reps):
            fn()
        e_.record()
        e_.synchronize()
        return s_.elapsed_time(e_) / reps

    B0 = int(os.environ.get("PWC_BENCH_BATCH", "16"))
    layers = []
    pyr = {1: 16, 2: 32, 3: 64, 4: 96, 5: 128, 6: 196}
    for l in range(1, 7):
        layers.append(("conv%daa/b" % l, 2 * B0, pyr[l], pyr[l], 448 >> l, 1024 >> l))
    for l in range(6, 1, -1):
        cin = 81 + (0 if l == 6 else pyr[l] + 4)
        for i, co in enumerate((128, 128, 96, 64, 32)):
            layers.append(("conv%d_%d" % (l, i), B0, cin, co, 448 >> l, 1024 >> l))
            cin += co
    layers.append(("dc_conv1", B0, 565, 128, 112, 256))
    layers.append(("dc_conv6", B0, 64, 32, 112, 256))
    layers = [l + (1,) for l in layers] + [("dc_conv2 d2", B0, 128, 128, 112, 256, 2), ("dc_conv3 d4", B0, 128, 128, 112, 256, 4),
                                           ("dc_conv4 d8", B0, 128, 96, 112, 256, 8), ("dc_conv5 d16", B0, 96, 64, 112, 256, 16)]
    only = os.environ.get("PWC_BENCH_ONLY")
    for name, B, cin, cout, H, W, D in layers:
        if only and only not in name:
            continue
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
        b = torch.zeros(cout, device=dev)
        up, wp = ops.pack_conv3x3_wino(w), ops.pack_conv3x3(w)
        y, y0 = torch.empty(B, cout, H, W, device=dev), torch.empty(B, cout, H, W, device=dev)
        nws = ops.conv3x3_workspace_bytes(B, cin, H, W, cout)
        ws = torch.empty(max(nws, 4) // 4, device=dev)
        nww = ops.conv3x3_wino_workspace_bytes(B, cin, H, W, cout, D)
        wws = torch.empty(max(nww, 4) // 4, device=dev)
        tw = t2(lambda: ops.conv3x3_wino(x, up, b, cout, out=y, dilation=D, workspace=wws if nww else None))
        td = t2(lambda: ops.conv3x3(x, wp, b, cout, dilation=D, out=y0, workspace=ws if (nws and D == 1) else None))
        print("%-12s B%-2d %3d->%3d @%3dx%-3d: wino %7.1f us   direct %7.1f us   x%.2f   preferred=%d   max diff %.1e" %
              (name, B, cin, cout, H, W, tw * 1e3, td * 1e3, td / tw, ops.conv3x3_wino_preferred(B, cin, H, W, cout, D), (y - y0).abs().max().item()), flush=True)

if mode == "kscan":        # time against the number of 4-channel chunks at fixed geometry: slope = per-chunk time, intercept = fixed cost per launch
    B, H, W = 16, 112, 256
    for cout in (128, 64, 32):
        pts = []
        for cin in (16, 32, 64, 128, 256, 512):
            x = torch.randn(B, cin, H, W, device=dev)
            w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
            b = torch.zeros(cout, device=dev)
            up = ops.pack_conv3x3_wino(w)
            y = torch.empty(B, cout, H, W, device=dev)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                ops.conv3x3_wino(x, up, b, cout, out=y)
            torch.cuda.synchronize()
            s_.record()
            for _ in range(20):
                ops.conv3x3_wino(x, up, b, cout, out=y)
            e_.record()
            e_.synchronize()
            pts.append((cin // 4, s_.elapsed_time(e_) / 20 * 1e3))
        (k0, t0), (k1, t1) = pts[2], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print("cout %3d: " % cout + "  ".join("%d:%.0fus" % p_ for p_ in pts) + "   slope %.2f us/chunk, intercept %.0f us per launch" % (slope, t0 - slope * k0), flush=True)

if mode in ("time", "all"):
    B, H, W = 16, 112, 256

    def t(fn, reps=10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / reps

    for name, cin, cout in (("conv2_0", 117, 128), ("conv2_1", 245, 128), ("conv2_2", 373, 96), ("conv2_3", 469, 64), ("conv2_4", 533, 32), ("dc_conv1", 565, 128)):
        x = torch.randn(B, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
        b = torch.zeros(cout, device=dev)
        up, wp = ops.pack_conv3x3_wino(w), ops.pack_conv3x3(w)
        y, y0 = torch.empty(B, cout, H, W, device=dev), torch.empty(B, cout, H, W, device=dev)
        tw = t(lambda: ops.conv3x3_wino(x, up, b, cout, out=y))
        td = t(lambda: ops.conv3x3(x, wp, b, cout, out=y0))
        gf = 2.0 * 9 * cin * cout * B * H * W / 1e9
        print("%-9s %3d->%3d: wino %7.1f us (%.0f TF direct-equivalent)   direct %7.1f us (%.0f TF)   x%.2f   max diff %.2e" %
              (name, cin, cout, tw * 1e3, gf / tw, td * 1e3, gf / td, td / tw, (y - y0).abs().max().item()), flush=True)
