#!/usr/bin/env python3
"""Would a lattice-major context network pay in half precision?  The dilated layers as they run today (dilation D on 16 images of 112x256)
against the same layers as dilation-1 convolutions on 16 D^2 images of (112/D)x(256/D) (what a SPLIT2 store of the previous layer
would hand them), plain and split filters; HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflow_amd import ops_f16 as F16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def t(fn, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for name, cin, cout, D in (("dc_conv2", 128, 128, 2), ("dc_conv3", 128, 128, 4), ("dc_conv4", 128, 96, 8), ("dc_conv5", 96, 64, 16)):
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    for split in (False, True):
        wp = F16.pack_conv3x3_f16(w, split=split)
        x = F16.to_c8(torch.nn.functional.leaky_relu(torch.randn(16, cin, 112, 256, generator=g), 0.1).to(dev) * 0.3)
        y = F16.conv3x3_f16(x, wp, b, cin, cout, dilation=D, split_w=split)
        a = t(lambda: F16.conv3x3_f16(x, wp, b, cin, cout, dilation=D, out=y, split_w=split))
        res = [a]
        for Dl in (D // 2, 1) if D > 2 else (1,):          # one lattice level less dilated, or fully dilation 1
            f = D // Dl
            xl = F16.to_c8(torch.nn.functional.leaky_relu(torch.randn(16 * f * f, cin, 112 // f, 256 // f, generator=g), 0.1).to(dev) * 0.3)
            yl = F16.conv3x3_f16(xl, wp, b, cin, cout, dilation=Dl, split_w=split)
            res.append(t(lambda: F16.conv3x3_f16(xl, wp, b, cin, cout, dilation=Dl, out=yl, split_w=split)))
        print("%s %s: dilation %d on 16 x 112x256: %.1f us | on lattices: %s" % (name, "split" if split else "plain", D, res[0], ", ".join("%.1f us" % v for v in res[1:])))
