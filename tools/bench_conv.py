#!/usr/bin/env python3
"""Micro-bench of single conv layers with HIP events.  Default geometry: BASELINE level 2 (B=16, 112x256);
PWC_BENCH_GEOM="B,H,W" overrides it; PWC_BENCH_WS=1 supplies a split-K workspace (then PWC_CONV_SPLIT=k forces k)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops
dev = torch.device('cuda:0')
B, H, W = [int(v) for v in os.environ.get("PWC_BENCH_GEOM", "16,112,256").split(",")]
cases = [("dc_conv1", 565, 128, 1), ("conv2_0", 117, 128, 1), ("conv2_2", 373, 96, 1), ("conv2_3", 469, 64, 1),
         ("conv2_4", 533, 32, 1), ("dc_conv2", 128, 128, 2), ("dc_conv4", 128, 96, 8), ("dc_conv5", 96, 64, 16)]
if len(sys.argv) > 1:
    custom = [a for a in sys.argv[1:] if ":" in a]          # e.g. k8:8:128:1  (name:cin:cout:dil)
    cases = [c for c in cases if c[0] in sys.argv[1:]] + [(n, int(ci), int(co), int(d)) for n, ci, co, d in
                                                          (a.split(":") for a in custom)]
def t(fn, reps=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps
g = torch.Generator().manual_seed(0)
for name, cin, cout, dil in cases:
    x = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = ops.pack_conv3x3(w)
    y = torch.empty(B, cout, H, W, device=dev)
    ws = None
    if os.environ.get("PWC_BENCH_WS") == "1":
        need = ops.conv3x3_workspace_bytes(B, cin, H, W, cout, 1, dil)
        ws = torch.empty((max(need, 4) // 4,), device=dev)
    ms = t(lambda: ops.conv3x3(x, wp, b, cout, dilation=dil, out=y, workspace=ws), reps=30)
    fl = 2.0 * cout * cin * 9 * H * W * B
    print("%-9s %4d->%3d d%-2d  %8.1f us  %6.1f TFLOP/s (%.1f%% of 157.3)" % (name, cin, cout, dil, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 1.573))
