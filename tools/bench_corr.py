#!/usr/bin/env python3
"""Micro-bench of the level-2 correlation / warp kernels (B=16, C=32, 112x256) with HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops
dev = torch.device('cuda:0')
B, C, H, W = 16, 32, 112, 256
g = torch.Generator().manual_seed(0)
a = torch.randn(B, C, H, W, generator=g).to(dev)
b = torch.randn(B, C, H, W, generator=g).to(dev)
flo = (torch.randn(B, 2, H, W, generator=g) * 2).to(dev)
arena = torch.zeros(B, 565, H, W, device=dev)
out = arena[:, 448:529]
def t(fn, reps=50):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: ops.correlation(a, b, leaky_slope=0.1, out=out))
by = (2 * C + 81) * H * W * 4 * B
print("corr  L2 B16: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80))
w = torch.empty_like(a)
ms = t(lambda: ops.warp(a, flo, 5.0, False, out=w))
by = (2 * C + 2) * H * W * 4 * B
print("warp  L2 B16: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)" % (ms * 1e3, by / ms / 1e6, by / ms / 1e6 / 80))
# plain copy for reference
src = torch.randn(64 * 1024 * 1024, device=dev); dst = torch.empty_like(src)
ms = t(lambda: dst.copy_(src), 20)
print("copy 256MB+256MB: %.1f us  %.0f GB/s" % (ms * 1e3, 2 * src.numel() * 4 / ms / 1e6))
