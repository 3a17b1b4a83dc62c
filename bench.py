#!/usr/bin/env python3
"""Headline benchmark: image-pairs/s of PWC-Net inference at 1024x448 fp32 on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a
child process); under torch.distributed.run it is one of the ranks.

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
HBM_CORR_MIX_CEILING_GBS = 5030.0    # streaming float4 kernel moving the correlation's 116 MB in + 150 MB out, cold, 16 384 WGs, nt stores (profiles/r04_corr_notes.md section 1)
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


PMC_FILE = "r04_pmc_traffic.json"


def pmc_record(key, applies):
    """rocprofv3 evidence for a probed kernel, from the COMMITTED profile of the same kernel on the same workload
    (profiles/r02_pmc_traffic.json, written by tools/install_profiles.py from separate --pmc / --kernel-trace passes;
    counters cannot be read from inside this process).  Returns {"traffic", "traffic_source", "rocprof_avg_ms"} or {}."""
    if not applies:
        return {}
    try:
        with open(os.path.join(REPO, "profiles", PMC_FILE)) as f:
            rec = json.load(f)[key]
        out = {"traffic": int(rec["traffic_bytes"]), "traffic_source": "profiles/%s (%s)" % (PMC_FILE, rec.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"))}
        if "rocprof_avg_ms" in rec:
            out["rocprof_avg_ms"] = float(rec["rocprof_avg_ms"])
        return out
    except (OSError, KeyError, ValueError):
        return {}


def event_time_ms(fns, reps, stream):
    """Mean duration of one launch over `reps` back-to-back launches, HIP events on the launch stream.  `fns` is one
    callable or a list that is cycled through (operand sets larger than the 256 MiB Infinity Cache in total, so that an
    HBM-bound kernel cannot be served from it)."""
    if callable(fns):
        fns = [fns]
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    for fn in fns:
        fn()
    stream.synchronize()
    start.record(stream)
    for i in range(reps):
        fns[i % len(fns)]()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def last_conv_kernel():
    from opticalflow_amd import _lib
    return _lib.load().pwc_last_conv_kernel().decode()


PROBE_REPS = 30


def probes_fp32(result, args, plan, B, H, W, h2, w2, stream):
    """roofline / roofline_corr / roofline_warp of the fp32 plan: HIP-event means over PROBE_REPS back-to-back launches.
    The HBM-bound probes rotate over three operand sets (3 x 266 MB at batch 16 > the 256 MiB Infinity Cache)."""
    from opticalflow_amd import ops
    full = B == 16 and (H, W) == (448, 1024)
    if args.conv_backend == "hip":
        cin = plan.arena[2].shape[1]
        direct_flops = 2.0 * 128 * cin * 9 * h2 * w2 * B
        probe_out = torch.empty((B, 128, h2, w2), device=plan.arena[2].device)       # (the plan's own ctx[0] may be in lattice-major layout)
        ms = event_time_ms(lambda: plan._conv("dc_conv1", plan.arena[2], probe_out, dilation=1), PROBE_REPS, stream)
        kern = last_conv_kernel()
        del probe_out
        wino, wino4 = "wino" in kern, "wino4" in kern
        # The roofline that bounds the kernel counts the multiplications it EXECUTES: Winograd F(2x2,3x3) does 16 per 2x2 outputs,
        # F(4x4,3x3) 36 per 4x4 outputs (instead of 36 / 144 for the direct form), per REAL input channel (the zero channels that pad
        # the last 4-channel chunk are not work); the direct-convolution equivalent is reported beside it and exceeds the fp32 MFMA
        # peak -- that is the point of the algorithm, not a measurement error.
        if wino4:
            flops = 2.0 * 36 * 128 * cin * (-(-h2 // 4)) * (-(-w2 // 4)) * B
        elif wino:
            flops = 2.0 * 16 * 128 * cin * (-(-h2 // 2)) * (-(-w2 // 2)) * B
        else:
            flops = direct_flops
        ach = flops / (ms * 1e-3) / 1e12
        rec = pmc_record("conv3x3_wino4_dc_conv1_b16" if wino4 else "conv3x3_wino_dc_conv1_b16" if wino else "conv3x3_mfma_dc_conv1_b16", full)
        result["roofline"] = {"kernel": "%s = %s (dc_conv1 %d->128 @%dx%d, B=%d)"
                                        % (kern, "<cout blocks, tile groups> Winograd F(4x4,3x3), fp32" if wino4 else
                                           "<cout blocks, tile groups> Winograd F(2x2,3x3), fp32" if wino else
                                           "<MT,NT,stride,dilation,two-per-CU,split-K>", cin, w2, h2, B),
                              "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": rec.get("traffic"),
                              "traffic_source": rec.get("traffic_source"),
                              "avg_launch_ms": round(ms, 4), "launches_timed": PROBE_REPS, "algorithmic_flop_per_launch": flops,
                              "direct_conv_flop_per_launch": direct_flops,
                              "direct_conv_equivalent_tflops": round(direct_flops / (ms * 1e-3) / 1e12, 3)}
        if "rocprof_avg_ms" in rec:
            result["roofline"]["frac_rocprof"] = round(flops / (rec["rocprof_avg_ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
    c2 = 32
    off = 448 + 81
    ar = plan.arena[2]
    # three operand sets laid out like the arena's (corr slot | c1 | up_flow) channels + the second image's features each
    sets = [(torch.empty((B, 81 + c2 + 2, h2, w2), device=ar.device), torch.empty((B, c2, h2, w2), device=ar.device),
             torch.empty((B, c2, h2, w2), device=ar.device)) for _ in range(3)]
    for mini, feat2, wrp in sets:
        mini[:, 81:].copy_(ar[:, off:off + c2 + 2])
        feat2.copy_(plan.c2[2])
        wrp.copy_(plan.warped[2])
    # (a) what levels 5..2 of the forward run: warp + correlation + LeakyReLU in one kernel
    bytes_fused = (2 * c2 + 81 + 2) * h2 * w2 * 4 * B
    ms = event_time_ms([(lambda m=m, f=f: ops.warp_correlation(m[:, 81:81 + c2], f, m[:, 81 + c2:], flow_scale=5.0, leaky_slope=0.1,
                                                               out=m[:, :81])) for m, f, _ in sets], PROBE_REPS, stream)
    gbs = bytes_fused / (ms * 1e-3) / 1e9
    from opticalflow_amd import _lib
    window = _lib.get_option("warpcorr_window") > 0      # the round-4 source-window kernel (default) or the round-2 gather kernel
    rec = pmc_record("warp_corr81_level2_b16" if window else "warp_corr81_round2_level2_b16", full)
    result["roofline_corr"] = {"kernel": "%s = warp + 81-channel correlation + LeakyReLU fused (level 2: C=32 @%dx%d, B=%d, arena-strided operands, "
                                         "the forward's own up_flow; 3 operand sets in rotation)"
                                         % ("warp_corr81_pipe_kernel<8>" if window else "corr81_dma_kernel<true>", w2, h2, B),
                               "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_COPY_CEILING_GBS, 4),
                               "traffic": rec.get("traffic"), "traffic_source": rec.get("traffic_source"),
                               "avg_launch_ms": round(ms, 4), "launches_timed": PROBE_REPS, "algorithmic_bytes_per_launch": bytes_fused}
    if "rocprof_avg_ms" in rec:
        result["roofline_corr"]["frac_rocprof"] = round(bytes_fused / (rec["rocprof_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    # (b) the two kernels it replaces (still the C-ABI entries pwc_corr_fwd / pwc_warp_fwd; level 6 runs the plain correlation)
    bytes_corr = (2 * c2 + 81) * h2 * w2 * 4 * B
    ms_c = event_time_ms([(lambda m=m, w_=w_: ops.correlation(m[:, 81:81 + c2], w_, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=m[:, :81]))
                          for m, _, w_ in sets], PROBE_REPS, stream)
    rolled = _lib.get_option("corr_pipe") > 0
    rec = pmc_record("corr81_roll_level2_b16" if rolled else "corr81_level2_b16", full)
    result["roofline_corr_plain"] = {"kernel": "%s (correlation alone, same operands)" % ("corr81_roll_kernel" if rolled else "corr81_dma_kernel<false>"), "bound": "hbm",
                                     "achieved": round(bytes_corr / (ms_c * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(bytes_corr / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": rec.get("traffic"),
                                     "traffic_source": rec.get("traffic_source"), "avg_launch_ms": round(ms_c, 4),
                                     "frac_of_mixed_stream_ceiling": round(bytes_corr / (ms_c * 1e-3) / 1e9 / HBM_CORR_MIX_CEILING_GBS, 4),
                                     "launches_timed": PROBE_REPS, "algorithmic_bytes_per_launch": bytes_corr}
    if args.conv_backend != "hip":
        result["roofline"] = result["roofline_corr"]
    bytes_warp = (2 * c2 + 2) * h2 * w2 * 4 * B
    ms_w = event_time_ms([(lambda m=m, f=f, w_=w_: ops.warp(f, m[:, 81 + c2:], 5.0, False, out=w_)) for m, f, w_ in sets],
                         PROBE_REPS, stream)
    result["roofline_warp"] = {"kernel": "warp_kernel<f32> (warp alone, level 2; 3 operand sets in rotation)", "bound": "hbm",
                               "achieved": round(bytes_warp / (ms_w * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(bytes_warp / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms_w, 4),
                               "launches_timed": PROBE_REPS, "algorithmic_bytes_per_launch": bytes_warp}
    result["roofline_corr"]["replaces_ms"] = round(ms_c + ms_w, 4)
    del sets


def probes_fp16(result, plan, B, H, W, h2, w2, stream):
    """The same three probes on the half-precision plan (c8 layout): algorithmic flops use the real 565 input
    channels of dc_conv1 (the arena carries 576 with its zero pad channels); bytes are halves."""
    from opticalflow_amd import ops_f16 as F16
    from opticalflow_amd.engine_f16 import BASE_G, CORR_G
    full = B == 16 and (H, W) == (448, 1024)
    flops = 2.0 * 128 * 565 * 9 * h2 * w2 * B
    ms = event_time_ms(lambda: plan._conv("dc_conv1", plan.arena[2], plan.ctx[0], dilation=1), PROBE_REPS, stream)
    ach = flops / (ms * 1e-3) / 1e12
    rec = pmc_record("conv3x3_f16_dc_conv1_b16", full)
    result["roofline"] = {"kernel": "%s = <MT,NT,stride,dilation,ring,-> (dc_conv1 565->128 @%dx%d, B=%d, v_mfma_f32_32x32x16_f16)"
                                    % (last_conv_kernel(), w2, h2, B),
                          "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "traffic": rec.get("traffic"),
                          "traffic_source": rec.get("traffic_source"),
                          "avg_launch_ms": round(ms, 4), "launches_timed": PROBE_REPS, "algorithmic_flop_per_launch": flops}
    if "rocprof_avg_ms" in rec:
        result["roofline"]["frac_rocprof"] = round(flops / (rec["rocprof_avg_ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)
        # the conv stack is power-limited: the same launch takes 0.49 ms on the plan's own (post-LeakyReLU) activations -- what `frac`
        # times -- and 0.54-0.59 ms on the unit-scale gaussian operands of tools/bench_conv_f16.py -- what the rocprofv3 average times
        result["roofline"]["frac_rocprof_operands"] = ("unit-scale gaussian (tools/bench_conv_f16.py); `frac` is timed on the plan's own "
                                                       "activations: power-limited kernel, clock depends on the operand statistics "
                                                       "(profiles/r03_f16_data_effect.txt)")
    ar = plan.arena[2]
    f0 = BASE_G + CORR_G
    c2 = 32
    # six operand sets (133 MB each in halves) so that the rotation exceeds the Infinity Cache
    sets = [(torch.zeros((B, CORR_G + 4, h2, w2, 8), device=ar.device, dtype=torch.float16),
             torch.zeros((B, 4, h2, w2, 8), device=ar.device, dtype=torch.float16)) for _ in range(6)]
    for mini, wrp in sets:
        mini[:, CORR_G:].copy_(ar[:, f0:f0 + 4])
        wrp.copy_(plan.warped[2])
    bytes_corr = (2 * c2 + 81) * h2 * w2 * 2 * B
    ms = event_time_ms([(lambda m=m, w_=w_: F16.correlation_c8(m[:, CORR_G:], w_, c2, leaky_slope=0.1, out=m[:, :CORR_G]))
                        for m, w_ in sets], PROBE_REPS, stream)
    gbs = bytes_corr / (ms * 1e-3) / 1e9
    result["roofline_corr"] = {"kernel": "corr81_c8_kernel (level 2: C=32 @%dx%d, B=%d, f16, fused LeakyReLU, arena-strided write; "
                                         "6 operand sets in rotation)" % (w2, h2, B),
                               "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_COPY_CEILING_GBS, 4),
                               "traffic": None, "avg_launch_ms": round(ms, 4), "launches_timed": PROBE_REPS,
                               "algorithmic_bytes_per_launch": bytes_corr}
    bytes_warp = (2 * c2 + 2) * h2 * w2 * 2 * B
    srcs = [plan.pyr_a[2][B:].clone() for _ in range(6)]
    ms = event_time_ms([(lambda s_=s_, w_=w_: F16.warp_c8(s_, ar[:, f0 + 4:f0 + 5], c2, flow_scale=5.0, out=w_))
                        for s_, (_, w_) in zip(srcs, sets)], PROBE_REPS, stream)
    result["roofline_warp"] = {"kernel": "warp_c8_kernel (level 2, f16; 6 operand sets in rotation)", "bound": "hbm",
                               "achieved": round(bytes_warp / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(bytes_warp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4),
                               "launches_timed": PROBE_REPS, "algorithmic_bytes_per_launch": bytes_warp}
    del sets, srcs


def bench_kitti(args, rank, world, dev, rehearse):
    """BASELINE configs[4]: KITTI-shaped stream (inference_kitti.py:227-266,296-314 loop) sharded over the ranks.  A step =
    `batch` uint8 375x1242 pairs per rank taken from HOST memory: pinned double-buffered upload on a copy stream,
    normalise + replicate-pad to 384x1280 + forward + unpad + resize as one HIP graph, gather of the quarter-resolution
    flows to rank 0.  The PCIe upload is INSIDE the timed region (unlike the headline metric)."""
    import torch.distributed as dist
    from opticalflow_amd import PWCDCNet, kitti
    from opticalflow_amd.parallel import broadcast_parameters
    from opticalflow_amd.weights import synthetic_state_dict
    H, W, B = 375, 1242, args.batch
    net = PWCDCNet(precision=args.precision)
    if rank == 0:
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=GAIN, bias_std=BIAS_STD))
    net = net.to(dev).eval()
    bcast_bytes = broadcast_parameters(net, src=0) if world > 1 else 0
    stream = kitti.ShardedStream.for_model(net, H, W, dev, batch=B)
    g = torch.Generator().manual_seed(100 + rank)
    pool = [(torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8),
             torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)) for _ in range(2 * B)]

    class Repeat:                                   # a stream of n pairs cycling over the pool (global index -> pool entry)
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return pool[(i // world) % len(pool)]

    def run(steps):
        last = None
        for _, full, gathered in stream.run(Repeat(steps * B * world)):
            last = gathered if rank == 0 else full
        torch.cuda.synchronize()
        return last

    log("rank %d/%d: KITTI stream %dx%d %s, %d pairs per step and rank; warm-up x%d" % (rank, world, W, H, args.precision, B, args.warmup))
    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = run(args.steps)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        pairs = world * B * args.steps
        ok = last is not None and bool(torch.isfinite(last[1]).all())
        result = {
            "metric": "image-pairs/sec KITTI 1242x375 stream %s (H2D included)" % args.precision, "value": round(pairs / elapsed, 3),
            "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f16",
            "data": "synthetic (uniform uint8 375x1242 pairs in host memory; seeded Kaiming-fan-in weights x0.85)",
            "config": {"workload": "BASELINE configs[4]: KITTI 1242x375 stream (inference_kitti.py path), %s, per-GPU double-buffered "
                                   "H2D, %d pairs per graph replay and rank" % (args.precision, B),
                       "pairs_per_gpu": B, "global_batch": B * world, "height": H, "width": W, "padded": [384, 1280],
                       "parallelism": "round-robin stream shard x%d (kitti.ShardedStream), weights broadcast %d B, quarter-resolution "
                                      "flow gather to rank 0" % (world, bcast_bytes)},
            "outputs_finite": ok,
        }
        if rehearse:
            result["config"]["rehearsal"] = "all %d ranks on cuda:0 over gloo: plumbing check, NOT a measurement" % world
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves, as a CHILD process and before this
    process has touched the GPU (no exec: replacing a process that initialised HIP is not allowed on the pool, and nothing here
    has).  The child's stdout is ours -- rank 0's single JSON line passes through -- and its exit code becomes ours."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    rc = 1
    for attempt in range(2):        # a port found by bind-and-close can be taken before torchrun binds it: one retry with a new one
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("--gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)))
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        out, _ = proc.communicate()
        rc = proc.returncode
        if out:
            sys.stdout.write(out)
            sys.stdout.flush()
        if rc == 0 or out.strip():  # success, or a rank got far enough to print: not a rendezvous failure
            break
        log("launch failed before any rank printed (rc %d)%s" % (rc, ": retrying with another port" if attempt == 0 else ""))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 100 timed, >= 20 warm-up forwards
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="image pairs per GPU per step (default 16)")
    ap.add_argument("--workload", default="pairs", choices=["pairs", "kitti"],
                    help="pairs = the headline metric (resident 1024x448 batch, BASELINE configs[2]/[3]); kitti = configs[4]: "
                         "375x1242 uint8 pairs streamed from host memory through kitti.ShardedStream (H2D included)")
    ap.add_argument("--height", type=int, default=448)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--conv-backend", default="hip", choices=["hip", "torch"])
    ap.add_argument("--precision", default=None, choices=["fp32", "fp16", "fp16-strict"],
                    help="fp32 = the headline metric (BASELINE configs[2]); fp16-strict = the half-precision mode that meets the 1e-3 tolerance "
                         "(default of --workload kitti); fp16 = half activations/filters everywhere, fp32 accumulation (configs[3], misses it)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for each step's flow gather before the next forward (round-3 behaviour, for A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.precision is None:
        # configs[4] names fp16; the half-precision mode that meets north_star's tolerance is the strict one (the fast mode: --precision fp16)
        args.precision = "fp16-strict" if args.workload == "kitti" else "fp32"
    if args.batch is None:
        args.batch = 16

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            return self_launch(args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch.distributed as dist
    from opticalflow_amd import PWCDCNet, _lib
    from opticalflow_amd.parallel import AsyncFlowGather, broadcast_parameters
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

    if args.workload == "kitti":
        return bench_kitti(args, rank, world, dev, rehearse)

    B, H, W = args.batch, args.height, args.width
    # borrow_output: the timed loop consumes each flow before the next forward (single GPU: finiteness check at the end; N > 1: the
    # gather copies it into its staging buffer on the compute stream), so the forward need not hand out a private copy
    net = PWCDCNet(conv_backend=args.conv_backend, use_graph=not args.no_graph, precision=args.precision, borrow_output=True)
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
    # N > 1: the flows of step k travel to rank 0 on a side stream while forward k+1 runs (parallel.AsyncFlowGather); the timed region
    # ends with a synchronize of that stream, so every gather is inside it
    gather = AsyncFlowGather(counts, (2, H // 4, W // 4), torch.float32, dev, dst=0) if world > 1 else None

    def step():
        flow = net(x)
        if world > 1:
            return gather.result(gather.submit(flow)) if args.sync_gather else gather.submit(flow)
        return flow

    log("rank %d/%d: net on %s, batch %d x 6x%dx%d resident; warm-up x%d" % (rank, world, dev, B, H, W, args.warmup))
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        if rank == 0 and (i < 3 or i == args.warmup - 1):
            log("warm-up step %d done" % i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        gather.synchronize()
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
                                    "fp32 accumulation, fp32 input/output%s" % (B, W, H, "" if args.precision == "fp16" else
                                                                                 " [strict mode: fp32 pyramid / levels 6-3 / warps, split filters]")),
                       "pairs_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "conv_backend": args.conv_backend, "hip_graph": not args.no_graph,
                       "parallelism": "batch-shard x%d, weights broadcast %d B, flow gather to rank 0" % (world, bcast_bytes)},
            "conv_gflop_per_pair": round(2 * macs / 1e9, 3),
            "mfma_util_whole_forward": round(2 * macs * value / 1e12 / (MFMA_F32_PEAK_TFLOPS if fp32 else MFMA_F16_PEAK_TFLOPS), 4),
        }
        cm = getattr(net._plan_for(x), "conv_macs", None)
        if fp32 and cm and cm["direct"]:
            # with Winograd layers the matrix cores execute fewer multiplications than the direct-convolution count above:
            # utilisation = executed / peak; the direct-equivalent rate (what a direct conv would need) is kept beside it
            exe = macs - (cm["direct"] - cm["executed"]) / B
            result["conv_gflop_executed_per_pair"] = round(2 * exe / 1e9, 3)
            result["conv_tflops_direct_equivalent"] = round(2 * macs * value / 1e12, 2)
            result["mfma_util_whole_forward"] = round(2 * exe * value / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)

    # ---- per-kernel roofline probes (rank 0; HIP events on the launch stream) -----------------------
    if rank == 0:
        if rehearse:
            result["config"]["rehearsal"] = "all %d ranks on cuda:0 over gloo: plumbing check, NOT a measurement" % world
        plan = net._plan_for(x)
        stream = torch.cuda.current_stream(dev)
        h2, w2 = H >> 2, W >> 2
        if fp32:
            probes_fp32(result, args, plan, B, H, W, h2, w2, stream)
        elif args.precision == "fp16-strict":
            # dominant kernel of the strict mode: dc_conv1 in half with split filters = twice the MFMA passes of the plain layer
            flops = 2.0 * 2 * 128 * 565 * 9 * h2 * w2 * B
            ms = event_time_ms(lambda: plan._conv("dc_conv1", plan.arena, plan.ctx[0], dilation=1), PROBE_REPS, stream)
            ach = flops / (ms * 1e-3) / 1e12
            result["roofline"] = {"kernel": "%s (dc_conv1 565->128 @%dx%d, B=%d, split hi+lo filters: 256 MFMA rows)" % (last_conv_kernel(), w2, h2, B),
                                  "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(ms, 4),
                                  "launches_timed": PROBE_REPS, "algorithmic_flop_per_launch": flops}
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
        if (H, W) == (448, 1024) and args.conv_backend == "hip":
            # ... and item 0 of the MEASURED batch (the graph replay's own output at the headline geometry: the routes a 64x128
            # input never takes -- F(4x4), lattice-major context network, split-K) against the oracle on the same pair
            torch.set_num_threads(host_cores())
            with torch.no_grad():
                ref0 = O.pwc_forward(sd_cpu, x[:1].cpu())
            result["epe_vs_cpu_oracle_headline_item0"] = float("%.3e" % O.epe(net(x)[:1].cpu(), ref0))
        if world == 1 and fp32 and args.conv_backend == "hip" and not args.no_graph:
            # side measurements, not the metric: the same workload (BASELINE configs[3]'s per-GPU shard) through the two half-precision
            # plans, timed the same way after the fp32 region.  The STRICT mode first: it is the one that meets north_star's 1e-3 mean
            # EPE (fp32 pyramid / levels 6-3 / warps, level 2 + context network in half with split filters); the fast mode
            # (half everywhere, ~1.2e-3 x mean |flow|) does not.  `python bench.py --precision fp16-strict | fp16` give full lines.
            f32out = net(x).clone()
            for prec, key, dtype in (("fp16-strict", "fp16_strict_same_workload",
                                      "f32 pyramid / levels 6-3 / warps; f16 activations with split (hi+lo) filters at level 2 and in the context network"),
                                     ("fp16", "fp16_same_workload", "f16 activations/filters, f32 accumulation")):
                log("side measurement: same workload, precision=%s" % prec)
                neth = PWCDCNet(use_graph=True, precision=prec).to(dev).eval()
                neth.load_state_dict(net.state_dict())
                xh = neth.graph_input(B, H, W, dev)
                xh.copy_(x)
                for _ in range(3):
                    neth(xh)
                torch.cuda.synchronize()
                th = time.perf_counter()
                for _ in range(args.steps):
                    outh = neth(xh)
                torch.cuda.synchronize()
                dth = time.perf_counter() - th
                result[key] = {
                    "value": round(B * args.steps / dth, 3), "unit": "image-pairs/s", "ms_per_step": round(1e3 * dth / args.steps, 4),
                    "dtype": dtype, "meets_1e-3_mean_epe": prec == "fp16-strict",
                    "epe_vs_fp32_plan": float("%.3e" % O.epe(outh.cpu(), f32out.cpu())),
                    "mean_abs_flow": float("%.3e" % f32out.abs().mean().item())}
                del neth, xh
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
