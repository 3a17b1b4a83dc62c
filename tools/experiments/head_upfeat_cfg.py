#!/usr/bin/env python3
"""predict_flowL + upfeatL (pwc_head_upfeat_fwd) alone at the level-4 / level-3 geometries of batch 16, HIP events, three operand sets in
rotation; PWC_STREAM_CFG=81|42|44|41 pins the tile shape (one process each).  python tools/experiments/head_upfeat_cfg.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflow_amd import ops, _lib
dev = torch.device("cuda:0")
for (B, cin, H, W) in [(16, 629, 28, 64), (16, 597, 56, 128), (12, 597, 56, 128), (8, 597, 56, 128), (4, 597, 56, 128), (2, 597, 56, 128), (8, 629, 28, 64), (4, 629, 28, 64)]:
    g = torch.Generator().manual_seed(1)
    xs = [torch.rand(B, cin, H, W, generator=g).to(dev) for _ in range(3)]
    hw = (torch.rand(2, cin, 3, 3, generator=g) * 0.05).to(dev); hb = torch.rand(2, generator=g).to(dev)
    uw = (torch.rand(cin, 2, 4, 4, generator=g) * 0.05).to(dev); ub = torch.rand(2, generator=g).to(dev)
    hp = ops.pack_conv3x3(hw)
    flow = torch.empty(B, 2, H, W, device=dev); up = torch.empty(B, 2, 2 * H, 2 * W, device=dev)
    need = ops.head_upfeat_workspace_bytes(B, cin, H, W)
    ws = torch.empty(max(need, 16) // 4, device=dev)
    for sliced in (0, 1):
        if not sliced and not ops.head_upfeat_supported(B, H, W):
            continue
        def run(i): ops.head_upfeat(xs[i % 3], hp, hb, uw, ub, flow, up, workspace=ws if sliced else None)
        for i in range(6): run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(60): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 60 * 1e3
        print("PWC_STREAM_CFG=%s %s %s: %.1f us per call (eager, launches included)  %.2f TB/s" % (
            os.environ.get("PWC_STREAM_CFG", "default"), (B, cin, H, W), "Cin slices (%d B workspace)" % need if sliced else "one pass",
            us, B * cin * H * W * 4 / us / 1e6), flush=True)

# the same two layers as ONE 10-channel 3x3 convolution on the matrix cores + pwc_upsample_entry_f32 (the small levels' route, option head10)
for (B, cin, H, W) in [(16, 629, 28, 64), (16, 597, 56, 128), (8, 597, 56, 128), (4, 597, 56, 128), (2, 597, 56, 128), (8, 629, 28, 64), (4, 629, 28, 64)]:
    g = torch.Generator().manual_seed(2)
    xs = [torch.rand(B, cin, H, W, generator=g).to(dev) for _ in range(3)]
    w10 = (torch.rand(10, cin, 3, 3, generator=g) * 0.05).to(dev); b10 = torch.rand(10, generator=g).to(dev)
    p10 = ops.pack_conv3x3(w10)
    dw = (torch.rand(2, 2, 4, 4, generator=g) * 0.05).to(dev); db = torch.rand(2, generator=g).to(dev)
    head = torch.empty(B, 10, H, W, device=dev); nxt = torch.empty(B, 4, 2 * H, 2 * W, device=dev)
    ws = torch.empty(max(ops.conv3x3_workspace_bytes(B, cin, H, W, 10), 16) // 4, device=dev)
    def run(i):
        ops.conv3x3(xs[i % 3], p10, b10, 10, leaky_slope=None, out=head, workspace=ws)
        ops.upsample_entry(head, dw, db, nxt)
    for i in range(6): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(60): run(i)
    e1.record(); torch.cuda.synchronize()
    print("%s 10-channel convolution (%s, workspace %d B) + upsample_entry: %.1f us per call pair (eager)" % (
        (B, cin, H, W), _lib.load().pwc_last_conv_kernel().decode(), ws.numel() * 4, e0.elapsed_time(e1) / 60 * 1e3), flush=True)
