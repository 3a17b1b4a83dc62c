import os, sys, torch
sys.path.insert(0, "/root/repo")
from opticalflow_amd import ops_f16 as F16
dev = torch.device("cuda:0")
def t(fn, reps=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for (C, H, W) in ((196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 56, 128)):
    a = torch.randn(16, (C + 7) // 8, H, W, 8, device=dev).half()
    b = torch.randn(16, (C + 7) // 8, H, W, 8, device=dev).half()
    out = torch.empty(16, 11, H, W, 8, device=dev, dtype=torch.float16)
    print(C, H, W, "%.1f us" % t(lambda: F16.correlation_c8(a, b, C, leaky_slope=0.1, out=out)))
