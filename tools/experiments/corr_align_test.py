#!/usr/bin/env python3
"""Does the level-2 correlation's time depend on the relative placement of in1 / in2 / out (HBM channel / bank aliasing)?
Operands are carved out of one big buffer at chosen byte offsets; three sets in rotation (cold)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflow_amd import _lib, ops
dev = torch.device("cuda:0")
B, C, H, W = 16, 32, 112, 256
n_in = B * C * H * W
n_out = B * 81 * H * W
_lib.set_option("corr_pipe_min_tiles", 1)

def t(fns, reps=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in fns: f()
    torch.cuda.synchronize(); s.record()
    for i in range(reps): fns[i % len(fns)]()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3

pool = torch.empty(3 * (2 * n_in + n_out) + 3 * (1 << 22), device=dev)
pool.normal_()
for off1, off2, offo in ((0, 0, 0), (64, 0, 0), (1024, 0, 0), (16384 + 64, 0, 0), (262144 + 1024 + 64, 0, 0), (0, 0, 1024 + 64), (1024 + 64, 0, 2048 + 192)):
    sets = []
    base = 0
    for k in range(3):
        a = pool[base + off1: base + off1 + n_in].view(B, C, H, W); base += n_in + (1 << 20)
        b = pool[base + off2: base + off2 + n_in].view(B, C, H, W); base += n_in + (1 << 20)
        o = pool[base + offo: base + offo + n_out].view(B, 81, H, W); base += n_out + (1 << 20)
        sets.append((a, b, o))
    res = []
    for on in (0, 1):
        _lib.set_option("corr_pipe", on)
        res.append(min(t([(lambda s=s: ops.correlation(s[0], s[1], 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=s[2])) for s in sets]) for _ in range(2)))
    print("float offsets in1 %7d in2 %7d out %7d (addr mod 4096: %4d %4d %4d): old %.1f us  new %.1f us" % (
        off1, off2, offo, sets[0][0].data_ptr() % 4096, sets[0][1].data_ptr() % 4096, sets[0][2].data_ptr() % 4096, res[0], res[1]), flush=True)
