#!/usr/bin/env python3
"""The fp16 dc_conv1 launch on operands of different statistics (HIP events, 30 launches each): the conv stack is power-limited, so the clock
the chip holds -- and with it the time -- depends on how much the operand bits toggle (MI355X guide, 'DVFS give-back').  This is what
separates bench.py's probe on the plan's own activations from tools/bench_conv_f16.py's unit-scale random operands."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import ops_f16 as F16

dev = torch.device("cuda:0")
B, cin, cout, H, W = 16, 565, 128, 112, 256
g = torch.Generator().manual_seed(0)
w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
b = torch.zeros(cout, device=dev)
wp = F16.pack_conv3x3_f16(w)
base = torch.randn(B, cin, H, W, generator=g).to(dev)
cases = {"unit-scale gaussian (tools/bench_conv_f16.py)": base,
         "LeakyReLU(0.1) of it, x 0.3 (like a dense block's activations)": F.leaky_relu(base, 0.1) * 0.3,
         "90 % zeros": base * (torch.rand(base.shape, generator=g).to(dev) > 0.9),
         "all zeros": torch.zeros_like(base)}
y = None
for name, x in cases.items():
    xc = F16.to_c8(x)
    if y is None:
        y = F16.conv3x3_f16(xc, wp, b, cin, cout)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        F16.conv3x3_f16(xc, wp, b, cin, cout, out=y)
    torch.cuda.synchronize()
    s.record()
    for _ in range(30):
        F16.conv3x3_f16(xc, wp, b, cin, cout, out=y)
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / 30
    print("%-66s %7.1f us  %6.1f TFLOP/s" % (name, ms * 1e3, 2.0 * cout * cin * 9 * H * W * B / ms / 1e9))
