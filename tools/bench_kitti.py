#!/usr/bin/env python3
"""KITTI-shaped stream (inference_kitti.py path, BASELINE configs[4] geometry in fp32): synthetic 375x1242 uint8 pairs
from HOST memory -> double-buffered H2D (kitti.PairStream) -> normalise -> replicate-pad to 384x1280 -> forward ->
unpad / resize -> flow on the device.  Reports pairs/s including the PCIe upload (the hot-path bench excludes it).
usage: python tools/bench_kitti.py [n_pairs] [fp32|fp16]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from opticalflow_amd import PWCDCNet  # noqa: E402
from opticalflow_amd.kitti import BatchStream, GraphedInfer, PairStream, model_infer  # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    dev = torch.device("cuda:0")
    net = PWCDCNet(use_graph=True, precision=precision).to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    g = torch.Generator().manual_seed(0)
    pool = [(torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8),
             torch.randint(0, 256, (375, 1242, 3), generator=g, dtype=torch.uint8)) for _ in range(8)]

    def pairs(k):
        for i in range(k):
            yield pool[i % len(pool)]

    def run(k):
        last = None
        for i1, i2 in PairStream(pairs(k), dev):
            last = model_infer(net, i1, i2)
        torch.cuda.synchronize()
        return last

    run(10)
    t0 = time.perf_counter()
    out = run(n)
    dt = time.perf_counter() - t0
    print("precision %s" % precision)
    print("KITTI stream 375x1242 -> 384x1280, batch 1, H2D included: %d pairs in %.3f s = %.1f pairs/s (%.2f ms/pair); "
          "flow %s" % (n, dt, n / dt, 1e3 * dt / n, tuple(out.shape)), flush=True)
    pipe = GraphedInfer(net, 375, 1242, dev)

    def run_graphed(k):
        last = None
        for u8 in PairStream(pairs(k), dev, raw=True):
            last = pipe(u8)
        torch.cuda.synchronize()
        return last

    ref = run(1)
    got = run_graphed(1)
    err = (ref - got).abs().max().item()
    run_graphed(10)
    t0 = time.perf_counter()
    run_graphed(n)
    dt = time.perf_counter() - t0
    print("same, pre/post + forward captured as one HIP graph (kitti.GraphedInfer): %.1f pairs/s (%.2f ms/pair); "
          "max |diff| vs eager %.2e" % (n / dt, 1e3 * dt / n, err), flush=True)
    for bsz in (4, 16):
        pipe_b = GraphedInfer(net, 375, 1242, dev, batch=bsz)

        def run_batched(k):
            last = None
            for u8 in BatchStream(pairs(k), dev, bsz):
                last = pipe_b(u8)
            torch.cuda.synchronize()
            return last

        run_batched(2 * bsz)
        t0 = time.perf_counter()
        run_batched(n)
        dt = time.perf_counter() - t0
        print("same, %d pairs per graph replay (kitti.BatchStream + GraphedInfer(batch=%d)): %.1f pairs/s (%.2f ms/pair)"
              % (bsz, bsz, n / dt, 1e3 * dt / n), flush=True)
    # compute-only reference point: same padded geometry, inputs resident
    x = torch.rand(1, 6, 384, 1280, device=dev)
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        net(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("forward only at 1x6x384x1280 (resident input, HIP graph): %.2f ms = %.1f pairs/s" % (10 * dt, 100 / dt), flush=True)


if __name__ == "__main__":
    main()
