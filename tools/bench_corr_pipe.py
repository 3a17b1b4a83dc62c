#!/usr/bin/env python3
"""Round-4 correlation kernels (pwc_corr_pipe.hip) against the round-2 kernels they replace on the large levels: bit-equality
on a set of shapes (ragged tiles, ragged channel chunks, arena-strided operands), then HIP-event timings at level 2 / 3 of the
benchmark batch with three operand sets in rotation (> 256 MiB at level 2: nothing is served from the Infinity Cache).
The kernels are selected in ONE process through pwc_set_option("corr_pipe" / "warpcorr_window").

usage: python tools/bench_corr_pipe.py [check|time|all] [smooth|noise]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflow_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("PWC_BENCH_B", "16"))
g = torch.Generator().manual_seed(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
flow_kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"


def make_flow(b, h, w, amp):
    if flow_kind == "noise":       # what the benchmark's random-weight decoder produces: neighbouring pixels differ by ~amp
        return (torch.randn(b, 2, h, w, generator=g) * amp).to(dev)
    return torch.nn.functional.interpolate(torch.randn(b, 2, max(h // 8, 2), max(w // 8, 2), generator=g) * amp, size=(h, w),
                                           mode="bicubic", align_corners=False).contiguous().to(dev)


def both(fn, shape_out):
    """fn(out) with the old and the new kernel into NaN-filled buffers (an output the kernel does not write must not pass
    because the allocator handed back a block that still holds the other kernel's result)"""
    res = []
    for on in (0, 1):
        _lib.set_option("corr_pipe", on)
        _lib.set_option("warpcorr_window", 2 * on)
        out = torch.full(shape_out, float("nan"), device=dev)
        r = fn(out)
        res.append(None if r is None else out)
    return res


def describe(o, n):
    bad = ~((o == n) | (torch.isnan(o) & torch.isnan(n)))
    idx = bad.nonzero()
    ch = sorted(set(idx[:, 1].tolist()))
    return "%d bad (%d NaN in new), batch %s, channels %s..%s (%d), rows %d..%d, cols %d..%d" % (
        int(bad.sum()), int(torch.isnan(n).sum()), sorted(set(idx[:, 0].tolist()))[:4], ch[:3], ch[-3:], len(ch),
        int(idx[:, 2].min()), int(idx[:, 2].max()), int(idx[:, 3].min()), int(idx[:, 3].max()))


def check():
    _lib.set_option("corr_pipe_min_tiles", 1)
    ok = True
    for shape in ((2, 32, 24, 64), (1, 64, 56, 128), (3, 33, 20, 44), (2, 128, 14, 32), (5, 32, 112, 256), (2, 196, 7, 16),
                  (16, 32, 112, 256), (7, 96, 28, 64), (1, 37, 9, 36)):
        b, c, h, w = shape
        c1 = torch.rand(shape, generator=g).to(dev) * 2 - 1
        c2 = torch.rand(shape, generator=g).to(dev) * 2 - 1
        arena = torch.full((b, 81 + c + 2, h, w), 7.0, device=dev)
        arena[:, 81:81 + c].copy_(c1)
        for leaky in (None, 0.1):
            for norm in (False, True):
                o, n = both(lambda out: ops.correlation(arena[:, 81:81 + c], c2, 4, 1, 4, 1, 1, 1.0, normalize=norm, leaky_slope=leaky,
                                                        out=out), (b, 81, h, w))
                same = torch.equal(o, n)
                ok &= same
                if not same:
                    print("PLAIN MISMATCH", shape, leaky, norm, describe(o, n), flush=True)
        flo = make_flow(b, h, w, 0.6)
        flo[0, :, : h // 3] *= 4.0
        for scale, align in ((5.0, False), (1.25, True)):
            o, n = both(lambda out: ops.warp_correlation(c1, c2, flo, flow_scale=scale, align_corners=align, leaky_slope=0.1, out=out),
                        (b, 81, h, w))
            if o is None or n is None:
                continue
            same = torch.equal(o, n)
            ok &= same
            if not same:
                print("FUSED MISMATCH", shape, scale, align, describe(o, n), flush=True)
        print("checked", shape, flush=True)
    _lib.set_option("corr_pipe_min_tiles", 1024)
    print("bit-equality old vs new:", "OK" if ok else "FAILED", flush=True)
    return ok


def t(fns, reps=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in fns:
        f()
    torch.cuda.synchronize()
    s.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def plan_operands():
    """c1 / c2 / up_flow of decoder levels 2 and 3 as the benchmark's forward leaves them (synthetic weights of bench.py)"""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    net = net.to(dev).eval()
    x = torch.rand(B, 6, 448, 1024, generator=torch.Generator().manual_seed(1234)).to(dev)
    net(x)
    plan = net._plan_for(x)
    out = {}
    for lvl, C in ((2, 32), (3, 64)):
        ar = plan.arena[lvl]
        off = plan.arena_base[lvl] + 81 if hasattr(plan, "arena_base") else 448 + 81
        out[lvl] = (ar[:, off:off + C].clone(), plan.c2[lvl].clone(), ar[:, off + C:off + C + 2].clone())
    return out


def timing():
    levels = [int(v) for v in os.environ.get("PWC_BENCH_LEVELS", "2,3").split(",")]
    po = plan_operands() if flow_kind == "plan" else None
    for lvl, C, H, W, scale in ((2, 32, 112, 256, 5.0), (3, 64, 56, 128, 2.5), (4, 96, 28, 64, 1.25), (5, 128, 14, 32, 0.625)):
        if lvl not in levels:
            continue
        sets = []
        for _ in range(3):
            if po is not None and lvl in po:
                c1, c2, flo = (v.clone() for v in po[lvl])
                if _ == 0:
                    print("level %d plan flow: mean |f| %.2f px, mean |df/dx| %.2f" % (lvl, (flo * scale).abs().mean().item(),
                          ((flo[..., 1:] - flo[..., :-1]) * scale).abs().mean().item()), flush=True)
            else:
                c1 = torch.randn(B, C, H, W, generator=g).to(dev)
                c2 = torch.randn(B, C, H, W, generator=g).to(dev)
                flo = make_flow(B, H, W, 0.6 if flow_kind == "smooth" else 0.4)
            sets.append((c1, c2, flo, torch.empty(B, 81, H, W, device=dev)))
        _lib.set_option("corr_pipe_min_tiles", 1)
        res = {}
        for name, on in (("old", 0), ("new", 1), ("old", 0), ("new", 1)):
            _lib.set_option("corr_pipe", on)
            _lib.set_option("warpcorr_window", 2 * on)
            plain = t([(lambda s=s: ops.correlation(s[0], s[1], 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=s[3])) for s in sets])
            fused = t([(lambda s=s: ops.warp_correlation(s[0], s[1], s[2], flow_scale=scale, leaky_slope=0.1, out=s[3])) for s in sets])
            res.setdefault(name, []).append((plain, fused))
        _lib.set_option("corr_pipe_min_tiles", 1024)
        algp = (2 * C + 81) * H * W * 4 * B
        algf = (2 * C + 81 + 2) * H * W * 4 * B
        for name in ("old", "new"):
            p = min(v[0] for v in res[name])
            f = min(v[1] for v in res[name])
            print("level %d (C=%3d %3dx%3d, %s flow) %s: plain %6.1f us = %.3f of 8 TB/s | fused %6.1f us = %.3f" %
                  (lvl, C, H, W, flow_kind, name, p, algp / p / 8e6, f, algf / f / 8e6), flush=True)


if __name__ == "__main__":
    rc = 0
    if what in ("check", "all"):
        rc = 0 if check() else 1
    if what in ("time", "all"):
        timing()
    sys.exit(rc)
