#!/usr/bin/env python3
"""Does the sliced streaming head's time depend on the plan instance (operand placement)?  Several plans in one process, HIP events
around predict_flow2 alone (eager), sliced and MFMA route, plus dc_conv1 behind it.  python tools/experiments/slice_instances.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflow_amd import PWCDCNet, _lib, ops
from opticalflow_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
nets = []
for i in range(4):
    net = PWCDCNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    x = torch.rand(B, 6, 448, 1024, generator=torch.Generator().manual_seed(1234)).to(dev)
    net(x)
    nets.append((net, net._plan_for(x)))
    junk = torch.empty((7 + 13 * i) * 1024 * 1024 // 4, device=dev)       # shift the next plan's addresses
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for i, (net, plan) in enumerate(nets):
    row = []
    for v in (512, 0):
        _lib.set_option("stream_slice_wgs", v)
        row.append(t(lambda: plan._conv("predict_flow2", plan.arena[2], plan.flow[2], act=False)))
        row.append(t(lambda: (plan._conv("predict_flow2", plan.arena[2], plan.flow[2], act=False), plan._context())))
    print("plan %d (arena %#x, workspace %#x): predict_flow2 sliced %.1f us, + context %.1f | MFMA route %.1f us, + context %.1f" % (
        i, plan.arena[2].data_ptr(), plan.workspace.data_ptr(), row[0], row[1], row[2], row[3]), flush=True)
_lib.set_option("stream_slice_wgs", 512)
