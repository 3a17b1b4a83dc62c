#!/usr/bin/env python3
"""Throughput with S independent forwards in flight on S HIP streams (one captured plan each, batch B per forward) against the
single-stream back-to-back replay bench.py measures.  The coarse decoder levels and the small pyramid levels launch 16-128
workgroups on 256 CUs; a second forward's large kernels can fill those holes.  usage: [PWC_BENCH_STREAMS=1,2,4] bench_streams.py [fp32|fp16] [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import PWCDCNet  # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda:0")
H, W = 448, 1024
sd = None
for S in [int(v) for v in os.environ.get("PWC_BENCH_STREAMS", "1,2,3").split(",")]:
    nets, xs, streams = [], [], []
    for s in range(S):
        net = PWCDCNet(use_graph=True, precision=prec).to(dev).eval()
        if sd is None:
            sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
        net.load_state_dict(sd)
        x = net.graph_input(B, H, W, dev)
        x.copy_(torch.rand(B, 6, H, W, generator=torch.Generator().manual_seed(1234 + s)).to(dev))
        nets.append(net)
        xs.append(x)
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            k = i % S
            with torch.cuda.stream(streams[k]):
                nets[k](xs[k])
        torch.cuda.synchronize()

    run(2 * S + 4)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    print("%s batch %d, %d stream(s): %.1f pairs/s (%.3f ms per forward of %d pairs)" % (prec, B, S, B * steps / dt, 1e3 * dt / steps, B), flush=True)
    del nets, xs, streams
    torch.cuda.empty_cache()
