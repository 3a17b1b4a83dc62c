#!/usr/bin/env python3
"""Fused warp + correlation (pwc_warp_corr81_fwd) against the two kernels it replaces, per decoder level at batch 16, HIP events,
three operand sets in rotation (> 256 MiB in total at level 2: nothing is served from the Infinity Cache).  Flows: smooth
(low-resolution noise upsampled, what a decoder produces) scaled like the level."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("PWC_BENCH_B", "16"))
g = torch.Generator().manual_seed(0)


def t(fns, reps=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in fns:
        f()
    torch.cuda.synchronize()
    s.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


LEVELS = [int(v) for v in os.environ.get("PWC_BENCH_LEVELS", "2,3,4,5").split(",")]       # profiler passes look at level 2 only
for lvl, C, H, W, scale in ((2, 32, 112, 256, 5.0), (3, 64, 56, 128, 2.5), (4, 96, 28, 64, 1.25), (5, 128, 14, 32, 0.625)):
    if lvl not in LEVELS:
        continue
    sets = []
    for _ in range(3):
        c1 = torch.randn(B, C, H, W, generator=g).to(dev)
        c2 = torch.randn(B, C, H, W, generator=g).to(dev)
        flo = torch.nn.functional.interpolate(torch.randn(B, 2, max(H // 8, 2), max(W // 8, 2), generator=g) * 0.6, size=(H, W),
                                              mode="bicubic", align_corners=False).contiguous().to(dev)
        sets.append((c1, c2, flo, torch.empty(B, 81, H, W, device=dev), torch.empty(B, C, H, W, device=dev)))
    fused = t([(lambda s=s: ops.warp_correlation(s[0], s[1], s[2], flow_scale=scale, leaky_slope=0.1, out=s[3])) for s in sets])
    warp = t([(lambda s=s: ops.warp(s[1], s[2], flow_scale=scale, out=s[4])) for s in sets])
    corr = t([(lambda s=s: ops.correlation(s[0], s[4], 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=s[3])) for s in sets])
    alg = (2 * C + 81 + 2) * H * W * 4 * B
    print("level %d (C=%3d %3dx%3d): fused %6.1f us = %6.1f GB/s algorithmic (%.1f%% of 8 TB/s) | warp %6.1f us + corr %6.1f us = %6.1f us"
          % (lvl, C, H, W, fused, alg / fused / 1e3, alg / fused / 1e3 / 80.0, warp, corr, warp + corr), flush=True)
