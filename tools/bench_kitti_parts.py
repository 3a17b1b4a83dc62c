#!/usr/bin/env python3
"""Where a step of the KITTI stream (bench.py --workload kitti) goes: the captured pipeline alone on a resident batch, the host
staging + upload alone, and the two together as ShardedStream runs them.  usage: bench_kitti_parts.py [fp16|fp32] [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import PWCDCNet, kitti  # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
H, W = 375, 1242
net = PWCDCNet(precision=prec)
net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
net = net.to(dev).eval()
g = torch.Generator().manual_seed(100)
pool = [(torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8), torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8))
        for _ in range(2 * B)]
pipe = kitti.GraphedInfer(net, H, W, dev, batch=B)
u8 = torch.stack([torch.stack(p) for p in pool[:B]]).to(dev)


def wall(fn, n):
    fn(max(3, n // 10))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def gpu_only(n):
    for _ in range(n):
        pipe(u8)


def stage_only(n):
    for _ in kitti.BatchStream((pool[i % len(pool)] for i in range(n * B)), dev, B):
        pass


def both(n):
    for b in kitti.BatchStream((pool[i % len(pool)] for i in range(n * B)), dev, B):
        pipe(b)


print("%s, %d pairs per step, 375x1242" % (prec, B))
ms = wall(gpu_only, steps)
print("captured pipeline alone (resident uint8 batch): %.3f ms/step = %.0f pairs/s" % (ms, B / ms * 1e3))
ms = wall(stage_only, steps)
print("host staging + upload alone:                    %.3f ms/step = %.0f pairs/s" % (ms, B / ms * 1e3))
ms = wall(both, steps)
print("together (BatchStream -> GraphedInfer):         %.3f ms/step = %.0f pairs/s" % (ms, B / ms * 1e3))
slot = torch.empty((B, 2, H, W, 3), dtype=torch.uint8).pin_memory()
t0 = time.perf_counter()
for _ in range(50):
    for k in range(B):
        slot[k, 0].copy_(pool[k][0])
        slot[k, 1].copy_(pool[k][1])
print("one-thread memcpy of one batch into pinned memory: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50):
    d = slot.to(dev, non_blocking=True)
    torch.cuda.synchronize()
print("H2D of one batch (%.1f MB), synchronous:         %.3f ms" % (slot.numel() / 1e6, (time.perf_counter() - t0) / 50 * 1e3))

# ---- host cost of one pipe() call: enqueue a few replays on an idle GPU and time the calls only
torch.cuda.synchronize()
costs = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        pipe(u8)
    costs.append((time.perf_counter() - t0) / 3 * 1e3)
    torch.cuda.synchronize()
print("host time of one pipe() call (D2D copy + graph launch), GPU idle: min %.3f ms, median %.3f ms" % (min(costs), sorted(costs)[5]))
t0 = time.perf_counter()
for _ in range(20):
    pipe.graph.replay()
torch.cuda.synchronize()
print("graph.replay() x20 back to back: %.3f ms each" % ((time.perf_counter() - t0) / 20 * 1e3))
