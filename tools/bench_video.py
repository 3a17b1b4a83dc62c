"""Frames/s of the pyramid-reusing FlowStream vs. the per-pair forward (video loop, pwc_extract_flow_video.py:262-305).
usage: [PWC_PRECISION=fp16] python tools/bench_video.py [H W] [batches...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from opticalflow_amd import PWCDCNet  # noqa: E402
from opticalflow_amd.video import FlowStream  # noqa: E402
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (448, 1024)
    batches = [int(a) for a in sys.argv[3:]] or [1, 4, 16]
    dev = torch.device("cuda:0")
    net = PWCDCNet(use_graph=True, precision=os.environ.get("PWC_PRECISION", "fp32")).to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    for B in batches:
        frames = torch.rand((B + 1, 3, H, W), device=dev)
        pairs = torch.cat([frames[:-1], frames[1:]], 1).contiguous()
        s = FlowStream(net, B, H, W, use_graph=True)
        s.prime(frames[0])
        iters = max(10, 200 // B)
        t_pair = timed(lambda: net(pairs), iters)
        t_strm = timed(lambda: s.push(frames[1:]), iters)
        print("B=%2d %dx%d  pairwise %.3f ms (%.1f pairs/s)   stream %.3f ms (%.1f frames/s)   x%.3f" % (
            B, W, H, t_pair * 1e3, B / t_pair, t_strm * 1e3, B / t_strm, t_pair / t_strm), flush=True)


if __name__ == "__main__":
    main()
