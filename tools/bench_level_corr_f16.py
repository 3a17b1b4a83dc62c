#!/usr/bin/env python3
"""Half-precision level entry: pwc_level_entry_c8_f16 + pwc_corr81_c8_f16 (two launches, warped features through HBM) against
pwc_level_corr81_c8_f16 (one launch, warped features in LDS).  HIP events, several operand sets in rotation, batch 16, 448x1024."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops_f16 as F16

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for l, (C, H, W) in {5: (128, 14, 32), 4: (96, 28, 64), 3: (64, 56, 128), 2: (32, 112, 256)}.items():
    g = (C + 7) // 8
    nset = 6
    sets = []
    for k in range(nset):
        c1 = (torch.randn(B, g, H, W, 8, device=dev) * 0.5).half()
        c2 = (torch.randn(B, g, H, W, 8, device=dev) * 0.5).half()
        head = torch.zeros(B, 2, H // 2, W // 2, 8, device=dev)
        smooth = torch.nn.functional.interpolate(torch.randn(B, 2, max(H // 16, 1), max(W // 16, 1), device=dev) * 1.5, size=(H // 2, W // 2), mode="bilinear")
        head[:, 0, :, :, 0:2] = smooth.permute(0, 2, 3, 1)
        head[:, 1] = torch.randn(B, H // 2, W // 2, 8, device=dev)
        ar = torch.zeros(B, 11 + g + 1, H, W, 8, device=dev, dtype=torch.float16)
        sets.append((c1, c2, head, ar, torch.zeros_like(c2)))
    dw, db = torch.randn(2, 2, 4, 4, device=dev) * 0.25, torch.zeros(2, device=dev)

    def two(s):
        c1, c2, head, ar, wr = s
        F16.level_entry(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=ar[:, 11:11 + g], flow_group=ar[:, 11 + g:], out=wr, flow_scale=1.25)
        F16.correlation_c8(c1, wr, C, leaky_slope=0.1, out=ar[:, :11])

    def one(s):
        c1, c2, head, ar, wr = s
        F16.level_entry_correlation(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=ar[:, 11:11 + g], flow_group=ar[:, 11 + g:], out=ar[:, :11],
                                    flow_scale=1.25, leaky_slope=0.1)

    def t(fn, reps=30):
        for s in sets:
            fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            fn(sets[r % nset])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t2, t1 = t(two), t(one)
    print("batch %d level %d (C=%3d %3dx%3d): entry + correlation %6.1f us | one kernel %6.1f us" % (B, l, C, H, W, t2, t1), flush=True)
