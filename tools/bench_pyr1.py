#!/usr/bin/env python3
"""Micro-bench of the fused first pyramid level (ops_f16.pyramid1_fused) against its four layers run one by one, batch-16
geometry (32 images of 448x1024), HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops_f16 as F16  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = [int(v) for v in os.environ.get("PWC_BENCH_GEOM", "16,448,1024").split(",")]
g = torch.Generator().manual_seed(0)
x = torch.rand(B, 6, H, W, generator=g).to(dev)
ws = [((torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5).to(dev), torch.zeros(co, device=dev))
      for ci, co in ((3, 16), (16, 16), (16, 16), (16, 32))]
packed, bias = F16.pack_pyramid1(*ws[0], *ws[1], *ws[2], *ws[3])
wp = [F16.pack_conv3x3_f16(w) for w, _ in ws[1:]]
out = torch.empty((2 * B, 4, H // 4, W // 4, 8), dtype=torch.float16, device=dev)
a = torch.empty((2 * B, 2, H // 2, W // 2, 8), dtype=torch.float16, device=dev)
bb = torch.empty_like(a)


def fused():
    F16.pyramid1_fused(x[:, :3], packed, bias, out=out[:B])
    F16.pyramid1_fused(x[:, 3:], packed, bias, out=out[B:])


def layers():
    F16.image_conv_s2(x[:, :3], *ws[0], out=a[:B])
    F16.image_conv_s2(x[:, 3:], *ws[0], out=a[B:])
    F16.conv3x3_f16(a, wp[0], ws[1][1], 16, 16, out=bb)
    F16.conv3x3_f16(bb, wp[1], ws[2][1], 16, 16, out=a)
    F16.conv3x3_f16(a, wp[2], ws[3][1], 16, 32, stride=2, out=out)


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
    return s.elapsed_time(e) / reps * 1e3


print("fused  (2 launches): %7.1f us" % t(fused))
print("layers (5 launches): %7.1f us" % t(layers))
