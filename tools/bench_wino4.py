#!/usr/bin/env python3
"""Winograd F(4x4,3x3) (pwc_conv3x3_wino4_fwd) against F(2x2,3x3): max / rms error vs an fp64 reference on small and ragged cases,
then the forward's large layers at batch 16 (HIP-event means).  usage: bench_wino4.py [check|layers|all]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops, _lib  # noqa: E402

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
    y4 = ops.conv3x3_wino4(xd, ops.pack_conv3x3_wino4(wd), bd, cout, leaky_slope=leaky)
    y2 = ops.conv3x3_wino(xd, ops.pack_conv3x3_wino(wd), bd, cout, leaky_slope=leaky)
    torch.cuda.synchronize()
    e4 = (y4.cpu().double() - ref).abs()
    e2 = (y2.cpu().double() - ref).abs()
    print("B%d %3d->%3d %3dx%3d: F(4x4) max %.3e rms %.3e | F(2x2) max %.3e rms %.3e | bound 3e-6*sqrt(9Cin) = %.2e   (|ref| max %.2f)"
          % (B, cin, cout, H, W, e4.max().item(), e4.pow(2).mean().sqrt().item(), e2.max().item(), e2.pow(2).mean().sqrt().item(),
             3e-6 * (cin * 9) ** 0.5, ref.abs().max().item()), flush=True)


if mode in ("check", "all"):
    for shp in ((1, 4, 32, 8, 64), (1, 8, 64, 8, 64), (2, 5, 7, 9, 12), (1, 16, 128, 12, 64), (2, 37, 96, 17, 72), (1, 64, 64, 20, 40),
                (1, 130, 128, 16, 36), (1, 21, 40, 33, 132), (1, 565, 128, 24, 64), (2, 117, 32, 40, 128)):
        case(*shp)
    case(1, 12, 32, 10, 36, leaky=None)


if mode == "pmc":          # one layer, few launches, for the profiler passes (tools/collect_profiles_r03.sh): dc_conv1 at batch 16
    B, H, W, cin, cout = 16, 112, 256, 565, 128
    x = torch.randn(B, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
    b = torch.zeros(cout, device=dev)
    u4, u2 = ops.pack_conv3x3_wino4(w), ops.pack_conv3x3_wino(w)
    y = torch.empty(B, cout, H, W, device=dev)
    for _ in range(40):        # F(2x2) first: ~90 ms that bring the clocks up (a different kernel name in the trace)
        ops.conv3x3_wino(x, u2, b, cout, out=y)
    for _ in range(30):        # back to back, as in the forward's steady state: the profiler's average is over these
        ops.conv3x3_wino4(x, u4, b, cout, out=y)
    torch.cuda.synchronize()


def t(fn, reps=10):
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


if mode in ("layers", "all"):
    B = int(os.environ.get("PWC_BENCH_BATCH", "16"))
    LAYERS = [("conv2aa/b", 2 * B, 32, 32, 112, 256), ("conv3aa/b", 2 * B, 64, 64, 56, 128),
              ("conv3_0", B, 149, 128, 56, 128), ("conv3_1", B, 277, 128, 56, 128), ("conv3_2", B, 405, 96, 56, 128),
              ("conv3_3", B, 501, 64, 56, 128), ("conv3_4", B, 565, 32, 56, 128),
              ("conv2_0", B, 117, 128, 112, 256), ("conv2_1", B, 245, 128, 112, 256), ("conv2_2", B, 373, 96, 112, 256),
              ("conv2_3", B, 469, 64, 112, 256), ("conv2_4", B, 533, 32, 112, 256), ("dc_conv1", B, 565, 128, 112, 256),
              ("dc_conv6", B, 64, 32, 112, 256)]
    tot = [0.0, 0.0]
    for name, b, cin, cout, H, W in LAYERS:
        x = torch.randn(b, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5
        bias = torch.randn(cout, device=dev) * 0.1
        u2, u4 = ops.pack_conv3x3_wino(w), ops.pack_conv3x3_wino4(w)
        y2, y4 = torch.empty(b, cout, H, W, device=dev), torch.empty(b, cout, H, W, device=dev)
        t2 = t(lambda: ops.conv3x3_wino(x, u2, bias, cout, out=y2))
        t4 = t(lambda: ops.conv3x3_wino4(x, u4, bias, cout, out=y4))
        kern = _lib.load().pwc_last_conv_kernel().decode()
        gf4 = 2.0 * 36 * cin * cout * b * ((H + 3) // 4) * ((W + 3) // 4) / 1e9        # executed flops of F(4x4)
        tot[0] += t2
        tot[1] += t4
        print("%-10s %3d->%3d @%3dx%-3d: F(2x2) %7.1f us   F(4x4) %7.1f us (%.2f of the fp32 MFMA peak)   x%.2f   preferred=%d   max diff %.1e   %s"
              % (name, cin, cout, H, W, t2 * 1e3, t4 * 1e3, gf4 / t4 / 157.3, t2 / t4, ops.conv3x3_wino4_preferred(b, cin, H, W, cout),
                 (y2 - y4).abs().max().item(), kern), flush=True)
    print("sum: F(2x2) %.1f us, F(4x4) %.1f us" % (tot[0] * 1e3, tot[1] * 1e3))
