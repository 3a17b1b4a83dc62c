#!/usr/bin/env python3
"""F(4x4,3x3) against inputs that are NOT zero-mean unit gaussians: per-output error of the three fp32 3x3 kernels
(direct MFMA, Winograd F(2x2), Winograd F(4x4)) against an fp64 convolution, absolute and relative to S = sum |x||w| of that output,
at dc_conv1's channel counts (565 -> 128).  Writes the table that profiles/r04_wino4_robustness.txt holds."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflow_amd import ops
dev = torch.device("cuda:0")
torch.set_num_threads(max(8, torch.get_num_threads()))
cin, cout, H, W = 565, 128, 32, 64
g = torch.Generator().manual_seed(11)
w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
b = torch.randn(cout, generator=g) * 0.1


def cases():
    x = torch.randn(1, cin, H, W, generator=g)
    yield "randn (zero mean, unit scale)", x, w
    yield "8 + randn (common-mode offset)", x + 8.0, w
    yield "relu-like: |randn| * 3 + 2", x.abs() * 3 + 2, w
    h = x.clone()
    m = torch.rand(h.shape, generator=g) < 1e-3
    h[m] *= 50.0
    yield "randn with 0.1 % entries x 50 (heavy tail)", h, w
    yield "8 + randn, filters with a common-mode offset too (w + 0.01)", x + 8.0, w + 0.01


print("%-62s %-9s %10s %10s %12s %12s" % ("input", "kernel", "max err", "rms err", "max err/S", "rms err/S"))
for name, x, wt in cases():
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    S = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    xd, wd, bd = x.to(dev), wt.to(dev), b.to(dev)
    outs = {"direct": ops.conv3x3(xd, ops.pack_conv3x3(wd), bd, cout, leaky_slope=None),
            "F(2x2)": ops.conv3x3_wino(xd, ops.pack_conv3x3_wino(wd), bd, cout, leaky_slope=None),
            "F(4x4)": ops.conv3x3_wino4(xd, ops.pack_conv3x3_wino4(wd), bd, cout, leaky_slope=None)}
    for k, o in outs.items():
        e = (o.cpu().double() - ref).abs()
        print("%-62s %-9s %10.2e %10.2e %12.2e %12.2e" % (name, k, e.max().item(), e.pow(2).mean().sqrt().item(),
                                                        (e / S).max().item(), (e / S).pow(2).mean().sqrt().item()), flush=True)
