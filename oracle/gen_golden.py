#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on the CPU.

Runs only in the build container (needs /root/reference, read-only).  The reference's source never
leaves that container: what is committed is data -- inputs (or their seeds) and the outputs the
reference produced -- plus this script.

Recipe: two process-local shims, no reference file is edited:
  * ``sys.modules['correlation_cuda']`` = empty module (reference correlation.py:4 imports the unbuilt
    CUDA extension at import time; it is never called because USE_ONNX_CORRELATION is set);
  * ``torch.Tensor.cuda`` = identity (reference PWCNet.py:167 calls .cuda() unconditionally).

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [old|wino]
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PWC_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
GAIN, BIAS_STD, WSEED = 0.85, 0.02, 0

sys.dont_write_bytecode = True
sys.modules["correlation_cuda"] = types.ModuleType("correlation_cuda")
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self

import warnings  # noqa: E402

warnings.filterwarnings("ignore")

from models.PWCNet import PWCDCNet as RefNet  # noqa: E402  (the reference's class)
from models.correlation_package import correlation as refcorr  # noqa: E402

refcorr.USE_ONNX_CORRELATION = True

# our own modules must come from the repo, not from the reference's `models` package
sys.path.insert(0, REPO)
from opticalflow_amd.weights import synthetic_state_dict  # noqa: E402


def rand(shape, seed, lo=0.0, hi=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo).to(dtype)


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def gen_corr():
    """Inputs are regenerated from (shape, seed) by the tests -- `rand` below is the recipe and the
    stored sha256 pins it -- so that the fixture holds only what the reference produced."""
    cases = {}
    corr = refcorr.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    shapes = [(2, 196, 7, 16), (1, 128, 14, 32), (1, 96, 10, 64), (1, 64, 20, 24), (2, 32, 12, 40),
              (1, 5, 3, 5), (1, 8, 9, 3), (2, 7, 13, 11), (1, 3, 1, 1), (1, 33, 8, 32), (1, 16, 9, 70)]
    for i, shp in enumerate(shapes):
        a = rand(shp, 100 + i, -1, 1)
        b = rand(shp, 200 + i, -1, 1)
        cases["shape_%d" % i] = np.array(shp)
        cases["digest_%d" % i] = np.array(digest(a) + digest(b))
        cases["out_%d" % i] = corr(a, b).numpy()               # reference fallback: raw sum (correlation.py:35-36)
    cases["n"] = np.array(len(shapes))
    # another operator configuration of the fallback (stride2, corr_multiply)
    a = rand((1, 6, 10, 12), 300, -1, 1)
    b = rand((1, 6, 10, 12), 301, -1, 1)
    c2 = refcorr.Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=2, corr_multiply=3)
    cases["s2_out"] = c2(a, b).numpy()
    np.savez_compressed(os.path.join(OUT, "g1_corr.npz"), **cases)
    print("g1_corr: %d cases" % len(shapes))


def gen_warp(net):
    cases = {}
    specs = []
    H, W = 12, 20
    x = rand((2, 5, H, W), 400, -1, 1)
    specs.append(("zero", x, torch.zeros(2, 2, H, W)))
    specs.append(("subpix", x, rand((2, 2, H, W), 401, -1.5, 1.5)))
    specs.append(("oob", x, rand((2, 2, H, W), 402, -30, 30)))
    # flows chosen so that sample points land exactly on pixel centres / borders of the
    # align_corners=False mapping x_src = (x+u)*W/(W-1) - 0.5
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    ys = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    u = (xs + 0.5) * (W - 1) / W - xs
    v = (ys + 0.5) * (H - 1) / H - ys
    specs.append(("centres", x, torch.cat((u.expand(2, 1, H, W), v.expand(2, 1, H, W)), 1).contiguous()))
    specs.append(("intflow", x, torch.cat((torch.full((2, 1, H, W), 2.0), torch.full((2, 1, H, W), -1.0)), 1)))
    x2 = rand((1, 32, 28, 64), 410, -1, 1)
    specs.append(("level", x2, rand((1, 2, 28, 64), 411, -4, 4)))
    x3 = rand((1, 3, 7, 5), 412, -1, 1)
    specs.append(("tiny", x3, rand((1, 2, 7, 5), 413, -2, 2)))
    x4 = rand((1, 2, 1, 1), 414, -1, 1)
    specs.append(("one", x4, torch.zeros(1, 2, 1, 1)))
    for name, xx, ff in specs:
        with torch.no_grad():
            out = net.warp(xx, ff.clone())
            torch.set_default_dtype(torch.float64)     # PWCNet.py:167 builds the ones-mask in the default dtype
            try:
                out64 = RefNet.warp(net, xx.double(), ff.double().clone())
            finally:
                torch.set_default_dtype(torch.float32)
        cases["x_" + name] = xx.numpy()
        cases["flo_" + name] = ff.numpy()
        cases["out_" + name] = out.numpy()
        cases["out64_" + name] = out64.numpy()
    cases["names"] = np.array([s[0] for s in specs])
    np.savez_compressed(os.path.join(OUT, "g2_warp.npz"), **cases)
    print("g2_warp: %d cases" % len(specs))


def gen_forward():
    net = RefNet().eval()
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synthetic_state_dict(manifest, seed=WSEED, gain=GAIN, bias_std=BIAS_STD)
    net.load_state_dict(sd, strict=True)
    cases = {"gain": np.array(GAIN), "bias_std": np.array(BIAS_STD), "wseed": np.array(WSEED)}
    cases["weights_digest"] = np.array(hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k, _ in manifest)).hexdigest())
    for tag, shape, seed in (("s", (1, 6, 64, 64), 1234), ("m", (2, 6, 128, 192), 1235)):
        x = rand(shape, seed)
        cases["xseed_" + tag] = np.array(seed)
        cases["xshape_" + tag] = np.array(shape)
        cases["xdigest_" + tag] = np.array(digest(x))
        with torch.no_grad():
            net.eval()
            f2 = net(x)
            net.train()
            outs = net(x)
            net.eval()
        cases["flow2_" + tag] = f2.numpy()
        for lvl, o in zip((2, 3, 4, 5, 6), outs):
            cases["train_flow%d_%s" % (lvl, tag)] = o.numpy()
        torch.set_default_dtype(torch.float64)
        try:
            net64 = RefNet().double().eval()
            net64.load_state_dict({k: v.double() for k, v in sd.items()})
            with torch.no_grad():
                f2d = net64(x.double())
        finally:
            torch.set_default_dtype(torch.float32)
        cases["flow2_f64_" + tag] = f2d.numpy()
        e = torch.sqrt(((f2.double() - f2d) ** 2).sum(1)).mean().item()
        print("g3_forward[%s]: mean|flow2| %.4f, fp32-vs-fp64 EPE %.3e" % (tag, f2.abs().mean().item(), e))
    np.savez_compressed(os.path.join(OUT, "g3_forward.npz"), **cases)
    return net, manifest


def gen_forward_wino():
    """g7 (round 3): inputs large enough that the build's fp32 plan takes its Winograd / fused warp+correlation / split-K routes
    (4x6x256x512: level 2 = 64x128 -> 256 workgroups per 128-cout launch) and the headline geometry 1x6x448x1024, so that the
    REFERENCE's own outputs -- not only the oracle's -- pin the kernels that carry the benchmark step.  Same synthetic-weight
    recipe as g3; inputs are regenerated from (shape, seed) by the tests and pinned by sha256."""
    net = RefNet().eval()
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synthetic_state_dict(manifest, seed=WSEED, gain=GAIN, bias_std=BIAS_STD)
    net.load_state_dict(sd, strict=True)
    cases = {"gain": np.array(GAIN), "bias_std": np.array(BIAS_STD), "wseed": np.array(WSEED)}
    cases["weights_digest"] = np.array(hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k, _ in manifest)).hexdigest())
    torch.set_default_dtype(torch.float64)
    try:
        net64 = RefNet().double().eval()
        net64.load_state_dict({k: v.double() for k, v in sd.items()})
    finally:
        torch.set_default_dtype(torch.float32)
    for tag, shape, seed in (("w", (4, 6, 256, 512), 1236), ("full", (1, 6, 448, 1024), 1237)):
        x = rand(shape, seed)
        cases["xseed_" + tag] = np.array(seed)
        cases["xshape_" + tag] = np.array(shape)
        cases["xdigest_" + tag] = np.array(digest(x))
        with torch.no_grad():
            net.eval()
            f2 = net(x)
            net.train()
            outs = net(x)
            net.eval()
            torch.set_default_dtype(torch.float64)
            try:
                f2d = net64(x.double())
            finally:
                torch.set_default_dtype(torch.float32)
        cases["flow2_" + tag] = f2.numpy()
        for lvl, o in zip((3, 4, 5, 6), outs[1:]):
            cases["train_flow%d_%s" % (lvl, tag)] = o.numpy()
        assert torch.equal(outs[0], f2)
        cases["flow2_f64_" + tag] = f2d.numpy()
        e = torch.sqrt(((f2.double() - f2d) ** 2).sum(1)).mean().item()
        print("g7_forward_wino[%s]: mean|flow2| %.4f, fp32-vs-fp64 EPE %.3e" % (tag, f2.abs().mean().item(), e))
    np.savez_compressed(os.path.join(OUT, "g7_forward_wino.npz"), **cases)


def gen_manifest(manifest):
    # default-initialised reference network: per-tensor sums pin the init recipe (PWCNet.py:134-138)
    torch.manual_seed(0)
    net = RefNet()
    sums = np.array([float(v.double().abs().sum()) for v in net.state_dict().values()])
    np.savez_compressed(os.path.join(OUT, "g5_manifest.npz"),
                        keys=np.array([k for k, _ in manifest]),
                        shapes=np.array([",".join(map(str, s)) for _, s in manifest]),
                        seed0_abs_sums=sums)
    print("g5_manifest: %d keys, %d params" % (len(manifest), sum(int(np.prod(s)) for _, s in manifest)))


def gen_flo():
    # The reference's writer (script_pwc.py:12-27) lives in a script that runs the whole CLI at import
    # time and needs cv2/imageio (absent), so its byte layout is restated here from the source text:
    # tag float32 202021.25, int32 W, int32 H, then the HxWx2 float32 array in C order.
    uv = (np.arange(3 * 5 * 2, dtype=np.float32).reshape(3, 5, 2) - 7.25) * 0.5
    blob = (np.array(202021.25, dtype=np.float32).tobytes() + np.array(5, dtype=np.int32).tobytes() +
            np.array(3, dtype=np.int32).tobytes() + uv.tobytes())
    np.savez_compressed(os.path.join(OUT, "g4_flo.npz"), uv=uv, blob=np.frombuffer(blob, dtype=np.uint8))
    print("g4_flo: %d bytes" % len(blob))


def gen_old():
    """PWCDCNet_old (PWCNet.py:277-491): 116-key variant without the *aa pyramid convs, mixed concat order,
    warp mask threshold 0.999.  Same synthetic-weight recipe as g3."""
    from models.PWCNet import PWCDCNet_old as RefOld
    net = RefOld().eval()
    manifest = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = synthetic_state_dict(manifest, seed=WSEED, gain=GAIN, bias_std=BIAS_STD)
    net.load_state_dict(sd, strict=True)
    cases = {"gain": np.array(GAIN), "bias_std": np.array(BIAS_STD), "wseed": np.array(WSEED),
             "keys": np.array([k for k, _ in manifest]),
             "shapes": np.array([",".join(map(str, s)) for _, s in manifest])}
    cases["weights_digest"] = np.array(hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k, _ in manifest)).hexdigest())
    for tag, shape, seed in (("s", (1, 6, 64, 64), 2234), ("m", (2, 6, 128, 192), 2235)):
        x = rand(shape, seed)
        cases["xseed_" + tag] = np.array(seed)
        cases["xshape_" + tag] = np.array(shape)
        cases["xdigest_" + tag] = np.array(digest(x))
        with torch.no_grad():
            f2 = net(x)
            net.train()
            outs = net(x)
            net.eval()
        cases["flow2_" + tag] = f2.numpy()
        for lvl, o in zip((2, 3, 4, 5, 6), outs):
            cases["train_flow%d_%s" % (lvl, tag)] = o.numpy()
        torch.set_default_dtype(torch.float64)
        try:
            net64 = RefOld().double().eval()
            net64.load_state_dict({k: v.double() for k, v in sd.items()})
            with torch.no_grad():
                f2d = net64(x.double())
        finally:
            torch.set_default_dtype(torch.float32)
        cases["flow2_f64_" + tag] = f2d.numpy()
        e = torch.sqrt(((f2.double() - f2d) ** 2).sum(1)).mean().item()
        print("g6_old[%s]: mean|flow2| %.4f, fp32-vs-fp64 EPE %.3e" % (tag, f2.abs().mean().item(), e))
    # warp with the old threshold (PWCNet.py:400: mask < 0.999)
    xx = rand((2, 5, 12, 20), 2400, -1, 1)
    ff = rand((2, 2, 12, 20), 2401, -3, 3)
    with torch.no_grad():
        cases["warp_x"], cases["warp_flo"] = xx.numpy(), ff.numpy()
        cases["warp_out"] = net.warp(xx, ff.clone()).numpy()
    np.savez_compressed(os.path.join(OUT, "g6_old.npz"), **cases)
    print("g6_old: %d keys" % len(manifest))


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["old"]:          # only the PWCDCNet_old fixture (leaves g1..g5 untouched)
        gen_old()
        return
    if sys.argv[1:] == ["wino"]:         # only the large-geometry forward fixture (round 3)
        gen_forward_wino()
        return
    gen_corr()
    net, manifest = gen_forward()
    gen_warp(net)
    gen_manifest(manifest)
    gen_flo()
    gen_old()
    gen_forward_wino()


if __name__ == "__main__":
    main()
