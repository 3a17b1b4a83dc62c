#!/usr/bin/env python3
"""Nothing but forwards of ONE plan, for rocprofv3 (tools/timeline.py --periodic cuts exactly one of them out of the trace):

    rocprofv3 --kernel-trace --stats -d DIR -o p --output-format csv -- python3 tools/profile_forward.py --precision fp16 --batch 16

bench.py also runs spot checks, roofline probes and the other precisions' side measurements in the same process, which is what made
round 3's fp16 timeline merge several forwards."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import PWCDCNet  # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "fp16-strict"])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--height", type=int, default=448)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--no-graph", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
net = PWCDCNet(use_graph=not args.no_graph, precision=args.precision).to(dev).eval()
net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
x = net.graph_input(args.batch, args.height, args.width, dev) if not args.no_graph else torch.empty(args.batch, 6, args.height, args.width, device=dev)
x.copy_(torch.rand(args.batch, 6, args.height, args.width, generator=torch.Generator().manual_seed(1234)))
for _ in range(3):
    net(x)
torch.cuda.synchronize()
for _ in range(args.steps):
    net(x)
torch.cuda.synchronize()
print("profile_forward: %d forwards of %s, batch %d, %dx%d" % (args.steps, args.precision, args.batch, args.width, args.height))
