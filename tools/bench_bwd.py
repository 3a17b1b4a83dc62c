#!/usr/bin/env python3
"""Micro-bench of the backward kernels (training callers of the native boundary): pwc_corr_bwd (tiled PWC path and the
generic gather) and pwc_warp_bwd (deterministic fixed-point / float atomics), level-2 geometry at batch 16, HIP events.
Algorithmic bytes: corr bwd reads in1, in2, grad_out and writes two gradients = (4C + 81) * H*W*4; warp bwd reads x, flo,
grad_out and writes grad_x, grad_flo = (3C + 4) * H*W*4."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, C, H, W = [int(v) for v in os.environ.get("PWC_BENCH_GEOM", "16,32,112,256").split(",")]


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
    return s.elapsed_time(e) / reps * 1e3


g = torch.Generator().manual_seed(0)
a = torch.randn(B, C, H, W, generator=g).to(dev)
b = torch.randn(B, C, H, W, generator=g).to(dev)
go = torch.randn(B, 81, H, W, generator=g).to(dev)
us = t(lambda: ops.correlation_backward(a, b, go))
nbytes = (4 * C + 81) * H * W * 4 * B
flops = 2 * 2 * 81 * C * H * W * B
print("corr81_bwd_kernel   (%d,%d,%d,%d): %8.1f us  %6.1f GB/s algorithmic (%.1f%% of 8 TB/s), %5.1f TFLOP/s fp32 VALU"
      % (B, C, H, W, us, nbytes / us / 1e3, nbytes / us / 1e3 / 80.0, flops / us / 1e6))
us = t(lambda: ops.correlation_backward(a, b, go[:, :25].contiguous(), 4, 1, 4, 1, 2), reps=3)
print("corr_bwd_generic    (stride2=2, 25 channels):  %8.1f us" % us)
gx = torch.randn(B, C, H, W, generator=g).to(dev)
nbytes = (3 * C + 4) * H * W * 4 * B
noise = (torch.randn(B, 2, H, W, generator=g) * 2).to(dev)                       # per-pixel random +-10 px after the x5 scale
smooth = torch.nn.functional.interpolate(torch.randn(B, 2, H // 16, W // 16, generator=g) * 2, size=(H, W), mode="bicubic").to(dev)
for name, flo in (("smooth flow", smooth.contiguous()), ("noise flow", noise)):
    for det in (True, False):
        us = t(lambda: ops.warp_backward(a, flo, gx, 5.0, False, 0.9999, deterministic=det))
        print("warp_bwd %-12s %-14s: %8.1f us  %6.1f GB/s algorithmic (%.1f%% of 8 TB/s)"
              % (name, "fixed-point" if det else "float atomics", us, nbytes / us / 1e3, nbytes / us / 1e3 / 80.0))
