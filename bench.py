#!/usr/bin/env python3
"""Headline benchmark: image-pairs/s of PWC-Net inference at 1024x448 fp32 on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A step = one PWCDCNet.forward over this rank's batch of B synthetic image pairs already resident in
HBM (full HIP path: correlation, warp, MFMA implicit-GEMM convs; HIP-graph replay), plus -- for N>1 --
the gather of the flow fields to rank 0.  N>1 is launched by torch.distributed.run, one process per
GPU over RCCL; work per GPU is fixed (weak scaling), no collective inside a forward.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      -- dominant kernel (the 3x3 MFMA conv, on its heaviest launch dc_conv1) vs fp32-MFMA peak
  roofline_corr -- the level-2 correlation kernel vs HBM peak (north_star's 60 % target)
  cpu_baseline  -- the CPU oracle timed on this host's cores on the same 1024x448 workload
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# dmabuf IPC for RCCL / cross-process device memory on this driver stack (must be set before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md: ~2.5 PF)
GAIN, BIAS_STD = 0.85, 0.02    # synthetic-weight recipe shared with tests/golden (mean |flow2| ~ 1)


def conv_macs_per_pair(H, W):
    """MACs of the 81 conv/deconv calls of one forward (reference models/PWCNet.py:184-268)."""
    from opticalflow_amd.engine import CONTEXT, DENSE_OUT, PYRAMID_CH, level_in_channels
    total = 0
    for l in range(1, 7):
        h, w = H >> l, W >> l
        cin, c = PYRAMID_CH[l - 1], PYRAMID_CH[l]
        total += 2 * (c * cin * 9 + 2 * c * c * 9) * h * w            # both images
    for l in (6, 5, 4, 3, 2):
        h, w = H >> l, W >> l
        cin = level_in_channels(l)
        for co in DENSE_OUT:
            total += co * cin * 9 * h * w
            cin += co
        total += 2 * cin * 9 * h * w                                   # predict_flow
        if l > 2:
            total += (2 * 2 + cin * 2) * 16 * h * w                     # deconv + upfeat (4x4 taps per input px)
    h, w = H >> 2, W >> 2
    cin = level_in_channels(2) + sum(DENSE_OUT)
    for co, _ in CONTEXT:
        total += co * cin * 9 * h * w
        cin = co
    total += 2 * cin * 9 * h * w
    return total


def log(msg):
    """Progress to stderr (the JSON line on stdout stays alone)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def host_cores():
    """CPU threads this process may really use: min(affinity, cgroup quota) -- a GPU box exposes the
    whole host in sched_getaffinity but caps the container's CPU share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("PWC_BENCH_CPU_THREADS", "16"))))


def pmc_traffic(key, applies):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately and
    corrected as the MI355X guide prescribes; see profiles/r01_pmc_traffic.json).  Counters cannot be read
    live, so this is the committed measurement of the same kernel on the same workload, or None."""
    if not applies:
        return None
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
            return int(json.load(f)[key]["traffic_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def event_time_ms(fn, reps, stream):
    """Average duration of fn() over `reps` back-to-back launches, HIP events on the launch stream."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    fn()
    stream.synchronize()
    start.record(stream)
    for _ in range(reps):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def probes_fp32(result, args, plan, B, H, W, h2, w2, stream):
    """roofline / roofline_corr / roofline_warp of the fp32 plan: HIP-event averages of back-to-back launches."""
    from opticalflow_amd import ops
    full = B == 16 and (H, W) == (448, 1024)
    if args.conv_backend == "hip":
        cin = plan.arena[2].shape[1]
        flops = 2.0 * 128 * cin * 9 * h2 * w2 * B
        ms = event_time_ms(lambda: plan._conv("dc_conv1", plan.arena[2], plan.ctx[0], dilation=1), 10, stream)
        ach = flops / (ms * 1e-3) / 1e12
        result["roofline"] = {"kernel": "conv3x3_mfma_kernel<4, 1, 1, 1, 1, 0> = <MT,NT,stride,dilation,two-per-CU,split-K> "
                                        "(dc_conv1 %d->128 @%dx%d, B=%d)" % (cin, w2, h2, B),
                              "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                              "traffic": pmc_traffic("conv3x3_mfma_dc_conv1_b16", full),
                              "avg_launch_ms": round(ms, 4), "algorithmic_flop_per_launch": flops}
    c2 = 32
    off = 448 + 81
    ar = plan.arena[2]
    bytes_corr = (2 * c2 + 81) * h2 * w2 * 4 * B
    ms = event_time_ms(lambda: ops.correlation(ar[:, off:off + c2], plan.warped[2], 4, 1, 4, 1, 1, 1.0,
                                               leaky_slope=0.1, out=ar[:, 448:529]), 20, stream)
    gbs = bytes_corr / (ms * 1e-3) / 1e9
    result["roofline_corr"] = {"kernel": "corr81_dma_kernel (level 2: C=32 @%dx%d, B=%d, fused LeakyReLU, arena write)" % (w2, h2, B),
                               "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_COPY_CEILING_GBS, 4),
                               "traffic": pmc_traffic("corr81_level2_b16", full),
                               "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": bytes_corr}
    if args.conv_backend != "hip":
        result["roofline"] = result["roofline_corr"]
    bytes_warp = (2 * c2 + 2) * h2 * w2 * 4 * B
    ms = event_time_ms(lambda: ops.warp(plan.c2[2], ar[:, off + c2:off + c2 + 2], 5.0, False, out=plan.warped[2]), 20, stream)
    result["roofline_warp"] = {"kernel": "warp_kernel<f32> (level 2)", "bound": "hbm",
                               "achieved": round(bytes_warp / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(bytes_warp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4)}


def probes_fp16(result, plan, B, H, W, h2, w2, stream):
    """The same three probes on the half-precision plan (c8 layout): algorithmic flops use the real 565 input
    channels of dc_conv1 (the arena carries 576 with its zero pad channels); bytes are halves."""
    from opticalflow_amd import ops_f16 as F16
    from opticalflow_amd.engine_f16 import BASE_G, CORR_G
    flops = 2.0 * 128 * 565 * 9 * h2 * w2 * B
    ms = event_time_ms(lambda: plan._conv("dc_conv1", plan.arena[2], plan.ctx[0], dilation=1), 10, stream)
    ach = flops / (ms * 1e-3) / 1e12
    result["roofline"] = {"kernel": "conv3x3_f16_kernel<4, 1, 1, 3> = <MT,stride,dilation,ring> (dc_conv1 565->128 @%dx%d, B=%d, "
                                    "v_mfma_f32_32x32x16_f16)" % (w2, h2, B),
                          "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4),
                          "traffic": pmc_traffic("conv3x3_f16_dc_conv1_b16", B == 16 and (H, W) == (448, 1024)),
                          "avg_launch_ms": round(ms, 4), "algorithmic_flop_per_launch": flops}
    ar = plan.arena[2]
    f0 = BASE_G + CORR_G
    c2 = 32
    bytes_corr = (2 * c2 + 81) * h2 * w2 * 2 * B
    ms = event_time_ms(lambda: F16.correlation_c8(ar[:, f0:f0 + 4], plan.warped[2], c2, leaky_slope=0.1,
                                                  out=ar[:, BASE_G:BASE_G + CORR_G]), 20, stream)
    gbs = bytes_corr / (ms * 1e-3) / 1e9
    result["roofline_corr"] = {"kernel": "corr81_c8_kernel (level 2: C=32 @%dx%d, B=%d, f16, fused LeakyReLU, arena write)" % (w2, h2, B),
                               "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_COPY_CEILING_GBS, 4),
                               "traffic": None, "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": bytes_corr}
    bytes_warp = (2 * c2 + 2) * h2 * w2 * 2 * B
    ms = event_time_ms(lambda: F16.warp_c8(plan.pyr_a[2][B:], ar[:, f0 + 4:f0 + 5], c2, flow_scale=5.0, out=plan.warped[2]), 20, stream)
    result["roofline_warp"] = {"kernel": "warp_c8_kernel (level 2, f16)", "bound": "hbm",
                               "achieved": round(bytes_warp / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(bytes_warp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="image pairs per GPU per step")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--conv-backend", default="hip", choices=["hip", "torch"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp32 = the headline metric (BASELINE configs[2]); fp16 = half activations/filters, fp32 accumulation (configs[3])")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch.distributed as dist
    from opticalflow_amd import PWCDCNet, _lib
    from opticalflow_amd.parallel import broadcast_parameters, gather_flows
    from opticalflow_amd.weights import synthetic_state_dict

    _lib.load()                                   # no HIP library -> no benchmark
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no ROCm device is visible")
    # PWC_BENCH_REHEARSE=1: every rank on cuda:0 over gloo -- lets the N>1 code path run on a one-GPU box
    # (a rehearsal of the plumbing, not a measurement; the JSON says so)
    rehearse = os.environ.get("PWC_BENCH_REHEARSE") == "1"
    dev = torch.device("cuda", 0 if rehearse else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, H, W = args.batch, args.height, args.width
    net = PWCDCNet(conv_backend=args.conv_backend, use_graph=not args.no_graph, precision=args.precision)
    if rank == 0:
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=GAIN, bias_std=BIAS_STD))
    net = net.to(dev).eval()
    bcast_bytes = broadcast_parameters(net, src=0) if world > 1 else 0

    x = torch.rand(B, 6, H, W, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    if not args.no_graph:
        # the pairs live in the captured forward's own input buffer (what an ingest stage would write into), so
        # no device-to-device staging copy is part of a step
        xin = net.graph_input(B, H, W, dev)
        xin.copy_(x)
        x = xin
    counts = [B] * world

    def step():
        flow = net(x)
        if world > 1:
            return gather_flows(flow, counts, dst=0)
        return flow

    log("rank %d/%d: net on %s, batch %d x 6x%dx%d resident; warm-up x%d" % (rank, world, dev, B, H, W, args.warmup))
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        if rank == 0:
            log("warm-up step %d done" % i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        log("timed region: %d steps in %.3f s" % (args.steps, elapsed))
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    fp32 = args.precision == "fp32"
    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        macs = conv_macs_per_pair(H, W)
        result = {
            "metric": "image-pairs/sec at 1024x448 %s" % args.precision, "value": round(value, 3), "unit": "image-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if fp32 else "f16",
            "data": "synthetic (uniform[0,1) image pairs; seeded Kaiming-fan-in weights x0.85, no checkpoint available)",
            "config": {"workload": ("BASELINE configs[2]: batch=%d/GPU %dx%d fp32, full HIP path (corr+warp+MFMA convs)%s"
                                    % (B, W, H, "" if args.conv_backend == "hip" else " [convs on PyTorch-ROCm: configs[1]]")) if fp32 else
                                   ("BASELINE configs[3] per-GPU shard: batch=%d/GPU %dx%d, fp16 activations and filters (c8 layout), "
                                    "fp32 accumulation, fp32 input/output" % (B, W, H)),
                       "pairs_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "conv_backend": args.conv_backend, "hip_graph": not args.no_graph,
                       "parallelism": "batch-shard x%d, weights broadcast %d B, flow gather to rank 0" % (world, bcast_bytes)},
            "conv_gflop_per_pair": round(2 * macs / 1e9, 3),
            "mfma_util_whole_forward": round(2 * macs * value / 1e12 / (MFMA_F32_PEAK_TFLOPS if fp32 else MFMA_F16_PEAK_TFLOPS), 4),
        }

    # ---- per-kernel roofline probes (rank 0; HIP events on the launch stream) -----------------------
    if rank == 0:
        if rehearse:
            result["config"]["rehearsal"] = "all %d ranks on cuda:0 over gloo: plumbing check, NOT a measurement" % world
        plan = net._plan_for(x)
        stream = torch.cuda.current_stream(dev)
        h2, w2 = H >> 2, W >> 2
        if fp32:
            probes_fp32(result, args, plan, B, H, W, h2, w2, stream)
        else:
            probes_fp16(result, plan, B, H, W, h2, w2, stream)

        # ---- parity spot check + CPU baseline (the oracle is the checker / the baseline, never the product)
        log("roofline probes done; parity spot check")
        from oracle import pwc_oracle as O
        sd_cpu = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        xs = torch.rand(1, 6, 64, 128, generator=torch.Generator().manual_seed(7))
        with torch.no_grad():
            ref = O.pwc_forward(sd_cpu, xs)
        result["epe_vs_cpu_oracle_64x128"] = float("%.3e" % O.epe(net(xs.to(dev)).cpu(), ref))
        if world == 1 and fp32 and args.conv_backend == "hip" and not args.no_graph:
            # side measurement, not the metric: the same workload through the half-precision plan (BASELINE configs[3]
            # per-GPU shard), timed the same way after the fp32 region; `python bench.py --precision fp16` is the full line
            log("side measurement: same workload, precision=fp16")
            net16 = PWCDCNet(use_graph=True, precision="fp16").to(dev).eval()
            net16.load_state_dict(net.state_dict())
            x16 = net16.graph_input(B, H, W, dev)
            x16.copy_(x)
            for _ in range(3):
                net16(x16)
            torch.cuda.synchronize()
            t16 = time.perf_counter()
            for _ in range(args.steps):
                out16 = net16(x16)
            torch.cuda.synchronize()
            dt16 = time.perf_counter() - t16
            f32out = net(x)
            result["fp16_same_workload"] = {
                "value": round(B * args.steps / dt16, 3), "unit": "image-pairs/s", "ms_per_step": round(1e3 * dt16 / args.steps, 4),
                "dtype": "f16 activations/filters, f32 accumulation",
                "epe_vs_fp32_plan": float("%.3e" % O.epe(out16.cpu(), f32out.cpu())),
                "mean_abs_flow": float("%.3e" % f32out.abs().mean().item())}
            del net16
        if world == 1 and not args.no_cpu_baseline:
            cores = host_cores()
            torch.set_num_threads(cores)
            log("cpu baseline: oracle forward at 1x6x%dx%d on %d threads for ~%.0f s" % (H, W, cores, args.cpu_seconds))
            xc = torch.rand(1, 6, H, W, generator=torch.Generator().manual_seed(1234))
            with torch.no_grad():
                O.pwc_forward(sd_cpu, xc)
                n, t0 = 0, time.perf_counter()
                while True:
                    O.pwc_forward(sd_cpu, xc)
                    n += 1
                    dt = time.perf_counter() - t0
                    if dt >= args.cpu_seconds or n >= 50:
                        break
            cpu_model = "unknown CPU"
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
            except (OSError, StopIteration):
                pass
            result["cpu_baseline"] = {"value": round(n / dt, 3), "unit": "image-pairs/s", "cores": cores, "kind": "port",
                                      "sample": "%d forwards of 1x6x%dx%d fp32 by oracle/pwc_oracle.py (torch CPU, %d threads on %s) "
                                                "after 1 warm-up" % (n, H, W, cores, cpu_model)}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
