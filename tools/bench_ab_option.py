#!/usr/bin/env python3
"""A/B of one pwc_set_option switch on the whole forward, interleaved rounds in ONE process (cdna guide rule 24):
    python tools/bench_ab_option.py <option> <value A> <value B> [batches, default 1,4,16]      (PWC_AB_PRECISION=fp16|fp16-strict: that plan)
The option is read when a plan is built, so each arm gets its own network, built under its value."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import PWCDCNet, _lib
from opticalflow_amd.weights import synthetic_state_dict
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
batches = [int(b) for b in (sys.argv[4] if len(sys.argv) > 4 else "1,4,16").split(",")]
dev = torch.device("cuda:0")
for B in batches:
    arms = {}
    for v in (va, vb):
        _lib.set_option(opt, v)
        net = PWCDCNet(use_graph=True, precision=os.environ.get("PWC_AB_PRECISION", "fp32")).to(dev).eval()
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
        x = net.graph_input(B, 448, 1024, dev)
        x.copy_(torch.rand(B, 6, 448, 1024, generator=torch.Generator().manual_seed(1234)))
        for _ in range(5):
            net(x)
        arms[v] = (net, x)
    torch.cuda.synchronize()
    res = {va: [], vb: []}
    steps = max(10, 160 // B)
    for r in range(5):
        for v in (va, vb):
            net, x = arms[v]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                net(x)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / steps * 1e3)
    fa, fb = arms[va][0](arms[va][1]), arms[vb][0](arms[vb][1])
    print("batch %2d: %s=%d %.3f ms (min %.3f) = %.0f pairs/s | %s=%d %.3f ms (min %.3f) = %.0f pairs/s | max |flow diff| %.2e" % (
        B, opt, va, sorted(res[va])[2], min(res[va]), B / min(res[va]) * 1e3, opt, vb, sorted(res[vb])[2], min(res[vb]), B / min(res[vb]) * 1e3,
        (fa - fb).abs().max().item()), flush=True)
_lib.set_option(opt, 1)
