#!/usr/bin/env python3
"""Correlation entry of the small pyramid levels: tiled kernels (option corr_small_tiles = 0: fused warp+correlation below level 6)
against the small-map route (warp, then corr81_small_kernel).  HIP events over 50 launches, 448x1024 geometry.
usage: python tools/bench_corr_small.py [batches, default 1,4,16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops, _lib

dev = torch.device("cuda:0")
LEVELS = {6: (196, 7, 16), 5: (128, 14, 32), 4: (96, 28, 64), 3: (64, 56, 128), 2: (32, 112, 256)}


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16").split(",")]:
    for l, (C, H, W) in LEVELS.items():
        a, b = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
        flo = torch.randn(B, 2, H, W, device=dev)
        out, wrp = torch.empty(B, 81, H, W, device=dev), torch.empty(B, C, H, W, device=dev)
        res = {}
        for v in (0, 1 << 30):
            _lib.set_option("corr_small_tiles", v)
            tag = "tiled" if v == 0 else "small"
            res[tag + " corr"] = t(lambda: ops.correlation(a, b, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=out))
            if v == 0:
                res["fused warp+corr"] = t(lambda: ops.warp_correlation(a, b, flo, flow_scale=1.25, leaky_slope=0.1, out=out))
            else:
                def two():
                    ops.warp(b, flo, flow_scale=1.25, out=wrp)
                    ops.correlation(a, wrp, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=out)
                res["warp, small corr"] = t(two)
        _lib.set_option("corr_small_tiles", 48)
        tiles = B * ((W + 31) // 32) * ((H + 7) // 8)
        print("batch %2d level %d (C=%3d %3dx%3d, %4d tiles): " % (B, l, C, H, W, tiles) + " | ".join("%s %6.1f us" % kv for kv in res.items()), flush=True)
