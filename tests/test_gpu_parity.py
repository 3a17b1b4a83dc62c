"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden fixtures.  Needs an MI355X: run with ``-m gpu``.

Tolerances (fp32 kernels, fp32 accumulation; summation order differs from the CPU):
  * correlation / conv: |err| <= 2e-6 * sqrt(K) * scale, K = reduction length
  * warp: 2e-6 absolute on O(1) data, identical mask decisions
  * full forward: mean EPE on raw flow2 <= 1e-3 vs the reference's fp32 AND fp64 outputs
    (BASELINE.json north_star bar); observed values are printed.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, seeded_rand
from oracle import pwc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu_device):
    from opticalflow_amd import _lib
    _lib.load()          # fail loudly if the HIP library is missing
    return gpu_device


# ------------------------------------------------------------------ correlation

@pytest.fixture
def corr_options():
    """flip the library's kernel-selection switches for one test and restore the defaults afterwards"""
    from opticalflow_amd import _lib
    names = ("corr_pipe", "corr_roll", "corr_pipe_min_tiles", "warpcorr_window")
    saved = {n: _lib.get_option(n) for n in names}
    yield _lib.set_option
    for n, v in saved.items():
        _lib.set_option(n, v)


@pytest.mark.parametrize("shape", [(2, 32, 24, 64), (1, 64, 56, 128), (5, 32, 112, 256), (3, 29, 20, 44), (16, 32, 112, 256), (2, 61, 17, 36)])
def test_corr_round4_kernels_bit_equal_round2_kernels(dev, corr_options, shape):
    """The round-4 correlation kernels (pwc_corr_pipe.hip: one workgroup per CU, LDS-DMA ring / rolling in2 window, output leaving
    plane by plane; fused form: warp taps sampled from an LDS window, gathers for the pixels that leave it) against the round-2
    kernels on the same operands: bit-identical, also on ragged tiles, ragged channel chunks, arena-strided operands, runs of tiles
    that wrap to a second segment, flows that leave the window and the image; NaN-filled outputs (an unwritten element must not pass)."""
    from opticalflow_amd import ops
    B, C, H, W = shape
    c1 = seeded_rand(shape, 170, -1, 1).to(dev)
    c2 = seeded_rand(shape, 171, -1, 1).to(dev)
    flo = seeded_rand((B, 2, H, W), 172, -2, 2)
    flo[0, :, : H // 3] *= 4.0                                        # part of image 0 samples far outside the window / the image
    flo = flo.to(dev)
    arena = torch.full((B, 81 + C + 2, H, W), 7.0, device=dev)
    arena[:, 81:81 + C].copy_(c1)
    corr_options("corr_pipe_min_tiles", 1)

    def run(new, roll, fn):
        corr_options("corr_pipe", new)
        corr_options("corr_roll", roll)
        corr_options("warpcorr_window", 2 * new)
        out = torch.full((B, 81, H, W), float("nan"), device=dev)
        assert fn(out) is not None
        return out

    for leaky, norm in ((None, False), (0.1, False), (0.1, True)):
        fn = lambda out: ops.correlation(arena[:, 81:81 + C], c2, 4, 1, 4, 1, 1, 1.0, normalize=norm, leaky_slope=leaky, out=out)  # noqa: E731
        old = run(0, 0, fn)
        assert not torch.isnan(old).any()
        assert torch.equal(old, run(1, 1, fn)), ("rolling / ring form", leaky, norm)
        assert torch.equal(old, run(1, 0, fn)), ("ring form", leaky, norm)
    for scale, align in ((5.0, False), (1.25, True)):
        fn = lambda out: ops.warp_correlation(c1, c2, flo, flow_scale=scale, align_corners=align, leaky_slope=0.1, out=out)       # noqa: E731
        old = run(0, 0, fn)
        assert not torch.isnan(old).any()
        assert torch.equal(old, run(1, 0, fn)), ("fused window form", scale, align)
    assert (arena[:, 81:81 + C] == c1).all()


def test_corr_window_kernel_more_than_64_tiles_per_workgroup(dev, corr_options):
    """The fused window kernel keeps the window origins of a workgroup's first 64 tiles in a per-wave table (OrgTable, lane n =
    tile n) and takes scalar loads for the tiles behind them: 16 640 one-tile images = 65 tiles for each of the 256 workgroups, a
    ragged channel chunk, flows that leave the window and the image -- bit-identical to the round-2 kernel."""
    from opticalflow_amd import ops
    B, C, H, W = 16640, 29, 8, 32
    g = torch.Generator(device=dev).manual_seed(173)
    c1 = torch.rand((B, C, H, W), device=dev, generator=g) * 2 - 1
    c2 = torch.rand((B, C, H, W), device=dev, generator=g) * 2 - 1
    flo = torch.rand((B, 2, H, W), device=dev, generator=g) * 4 - 2
    flo[::7] *= 6.0
    flo[16384:] += 1.5                                                 # the tiles past the table have a flow of their own
    corr_options("corr_pipe_min_tiles", 1)
    outs = []
    for new in (0, 1):
        corr_options("warpcorr_window", 2 * new)
        out = torch.full((B, 81, H, W), float("nan"), device=dev)
        assert ops.warp_correlation(c1, c2, flo, flow_scale=1.25, leaky_slope=0.1, out=out) is not None
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0], outs[1])
    bad = ~(outs[0][16384:] == outs[1][16384:])
    assert not bad.any()


def test_set_option_rejects_unknown_names(dev):
    from opticalflow_amd import _lib
    with pytest.raises(_lib.PwcHipError):
        _lib.set_option("no_such_option", 1)
    assert _lib.get_option("corr_pipe_min_tiles") >= 1

def test_corr_golden_fast_and_scalar_paths(dev):
    from opticalflow_amd import ops
    g = load_golden("g1_corr.npz")
    for i in range(int(g["n"])):
        shp = g["shape_%d" % i]
        a = seeded_rand(shp, 100 + i, -1, 1)
        b = seeded_rand(shp, 200 + i, -1, 1)
        ref = torch.from_numpy(g["out_%d" % i])
        got = ops.correlation(a.to(dev), b.to(dev), 4, 1, 4, 1, 1, 1.0).cpu()
        tol = 2e-6 * float(shp[1]) ** 0.5 * max(1.0, ref.abs().max().item())
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= tol, (tuple(shp), (got - ref).abs().max().item())
        gotn = ops.correlation(a.to(dev), b.to(dev), 4, 1, 4, 1, 1, 1.0, normalize=True).cpu()
        assert torch.allclose(gotn, ref / float(shp[1]), rtol=1e-5, atol=1e-6)


def test_corr_generic_path_stride2_multiply(dev):
    from opticalflow_amd import ops
    g = load_golden("g1_corr.npz")
    a = seeded_rand((1, 6, 10, 12), 300, -1, 1).to(dev)
    b = seeded_rand((1, 6, 10, 12), 301, -1, 1).to(dev)
    got = ops.correlation(a, b, 4, 1, 4, 1, 2, 3.0).cpu()
    assert torch.allclose(got, torch.from_numpy(g["s2_out"]), rtol=1e-5, atol=1e-5)
    # kernel_size 3 / stride1 2 against the oracle's statement of the CUDA semantics
    a = seeded_rand((2, 4, 15, 17), 310, -1, 1)
    b = seeded_rand((2, 4, 15, 17), 311, -1, 1)
    ref = O.correlation(a, b, 3, 3, 6, 2, 2, 1, normalize=True)
    got = ops.correlation(a.to(dev), b.to(dev), 3, 3, 6, 2, 2, 1.0, normalize=True).cpu()
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)


def test_corr_leaky_fused_into_arena_slot(dev):
    from opticalflow_amd import ops
    B, C, H, W = 2, 32, 24, 64
    a = seeded_rand((B, C, H, W), 1, -1, 1)
    b = seeded_rand((B, C, H, W), 2, -1, 1)
    arena = torch.full((B, 200, H, W), 7.0, device=dev)
    arena[:, 100:100 + C] = a.to(dev)                     # in1 is itself an arena slice (batch-strided)
    ops.correlation(arena[:, 100:100 + C], b.to(dev), 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=arena[:, 10:91])
    ref = O.leaky_relu(O.correlation(a, b, 4, 1, 4, 1, 1, 1))
    assert (arena[:, 10:91].cpu() - ref).abs().max().item() < 2e-5
    assert (arena[:, :10] == 7).all() and (arena[:, 91:100] == 7).all() and (arena[:, 132:] == 7).all()


def test_corr_properties_full_size(dev):
    """Size-independent properties at the BASELINE level-2 geometry (B=4, C=32, 112x256)."""
    from opticalflow_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(4, 32, 112, 256, generator=g).to(dev)
    b = torch.randn(4, 32, 112, 256, generator=g).to(dev)
    c = ops.correlation(a, b)
    # bilinearity
    assert torch.allclose(ops.correlation(2 * a, b), 2 * c, rtol=1e-5, atol=1e-4)
    assert torch.allclose(ops.correlation(a, b + b), 2 * c, rtol=1e-5, atol=1e-4)
    # centre channel == plain channel dot product
    assert torch.allclose(c[:, 40], (a * b).sum(1), rtol=1e-4, atol=1e-3)
    # symmetry: corr(a,b)[d](p) == corr(b,a)[-d](p+d): check d=(dy=-4,dx=+4) <-> (+4,-4)
    cba = ops.correlation(b, a)
    ch, chm = 0 * 9 + 8, 8 * 9 + 0
    assert torch.allclose(c[:, ch, 4:, :-4], cba[:, chm, :-4, 4:], rtol=1e-4, atol=1e-3)
    # out-of-image displacements are exactly zero
    assert (c[:, 0, :4, :] == 0).all() and (c[:, 80, -4:, :] == 0).all()


@pytest.mark.parametrize("shape,scale,align", [((2, 32, 24, 64), 5.0, False), ((1, 64, 56, 128), 2.5, True), ((3, 13, 20, 44), 1.25, False),
                                               ((2, 128, 14, 32), 0.625, False), ((16, 32, 112, 256), 5.0, False)])
def test_warp_correlation_fused_equals_two_kernels(dev, request, shape, scale, align):
    """pwc_warp_corr81_fwd (warp producer waves + LDS-DMA loader + nine fma waves in one persistent kernel) is BIT-IDENTICAL to
    pwc_warp_fwd followed by pwc_corr_fwd -- flows that leave the image, ragged tiles, ragged channel chunks, arena-strided
    operands, both scale modes -- and agrees with the oracle."""
    from opticalflow_amd import ops, _lib
    B, C, H, W = shape
    # maps of a few tiles would send pwc_corr_fwd to its small-map kernel (another summation order: test_corr_small_map_kernel)
    request.addfinalizer(lambda: _lib.set_option("corr_small_tiles", 48))
    _lib.set_option("corr_small_tiles", 0)
    c1 = seeded_rand(shape, 70, -1, 1).to(dev)
    c2 = seeded_rand(shape, 71, -1, 1).to(dev)
    flo = seeded_rand((B, 2, H, W), 72, -3, 3)
    flo[0, :, : H // 3] *= 4.0                                        # part of image 0 samples far outside
    flo = flo.to(dev)
    arena = torch.full((B, 81 + C + 2, H, W), 7.0, device=dev)         # [corr slot | c1 | up_flow] like the plan's arena
    arena[:, 81:81 + C].copy_(c1)
    arena[:, 81 + C:].copy_(flo)
    got = ops.warp_correlation(arena[:, 81:81 + C], c2, arena[:, 81 + C:], flow_scale=scale, align_corners=align,
                               leaky_slope=0.1, out=arena[:, :81])
    assert got is not None
    warped = ops.warp(c2, flo, flow_scale=scale, align_corners=align)
    two = ops.correlation(c1, warped, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1)
    assert torch.equal(arena[:, :81], two)
    assert torch.equal(arena[:, 81:81 + C], c1) and torch.equal(arena[:, 81 + C:], flo)            # neighbours untouched
    gotn = ops.warp_correlation(c1, c2, flo, flow_scale=scale, align_corners=align, normalize=True)
    assert torch.equal(gotn, ops.correlation(c1, warped, 4, 1, 4, 1, 1, 1.0, normalize=True))
    if B * H * W <= 100000:
        ref = O.leaky_relu(O.correlation(c1.cpu(), O.warp(c2.cpu(), flo.cpu() * scale, align_corners=align), 4, 1, 4, 1, 1, 1))
        assert (arena[:, :81].cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(ops.warp_correlation(c1, c2, flo, flow_scale=scale, align_corners=align, leaky_slope=0.1), two)   # repeatable
    # outside the fused kernel's geometry: nothing launched, caller falls back
    assert ops.warp_correlation(c1[..., :-1].contiguous(), c2[..., :-1].contiguous(), flo[..., :-1].contiguous()) is None


def test_corr_backward_matches_autograd_of_oracle(dev):
    from opticalflow_amd import ops
    a = seeded_rand((2, 6, 9, 11), 20, -1, 1).requires_grad_(True)
    b = seeded_rand((2, 6, 9, 11), 21, -1, 1).requires_grad_(True)
    go = seeded_rand((2, 81, 9, 11), 22, -1, 1)
    O.correlation(a, b, 4, 1, 4, 1, 1, 1).backward(go)
    g1, g2 = ops.correlation_backward(a.detach().to(dev), b.detach().to(dev), go.to(dev))
    assert torch.allclose(g1.cpu(), a.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(g2.cpu(), b.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 13, 20, 45), (1, 32, 24, 64), (3, 5, 7, 9)])
def test_corr_backward_tiled_pwc_config(dev, shape):
    """corr81_bwd_kernel (PWC-Net's pad 4 / k 1 / d 4 / strides 1, fp32: LDS-tiled gather) vs autograd through the oracle:
    ragged tiles, widths that are not multiples of 4, both scale modes; bit-reproducible (no atomics)."""
    from opticalflow_amd import ops
    B, C, H, W = shape
    a = seeded_rand(shape, 23, -1, 1).requires_grad_(True)
    b = seeded_rand(shape, 24, -1, 1).requires_grad_(True)
    go = seeded_rand((B, 81, H, W), 25, -1, 1)
    O.correlation(a, b, 4, 1, 4, 1, 1, 1).backward(go)
    ad, bd, god = a.detach().to(dev), b.detach().to(dev), go.to(dev)
    g1, g2 = ops.correlation_backward(ad, bd, god)
    assert torch.allclose(g1.cpu(), a.grad, rtol=1e-4, atol=2e-5) and torch.allclose(g2.cpu(), b.grad, rtol=1e-4, atol=2e-5)
    h1, h2 = ops.correlation_backward(ad, bd, god)
    assert torch.equal(g1, h1) and torch.equal(g2, h2)
    n1, n2 = ops.correlation_backward(ad, bd, god, normalize=True)
    assert torch.allclose(n1.cpu(), a.grad / C, rtol=1e-4, atol=2e-5) and torch.allclose(n2.cpu(), b.grad / C, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("cfg", [(3, 3, 6, 2, 2), (20, 3, 20, 1, 2), (2, 1, 2, 1, 1), (4, 1, 4, 2, 1), (1, 3, 4, 1, 2), (0, 1, 2, 1, 2)])
def test_corr_backward_general_configs(dev, cfg):
    """pwc_corr_bwd accepts any (pad, kernel_size, max_displacement, stride1, stride2) like the reference's backward
    (correlation_cuda_kernel.cu:150-334); checked against autograd through the oracle's general correlation, fp32 and fp16."""
    from opticalflow_amd import ops
    pad, k, d, s1, s2 = cfg
    C, H, W = 4, 15 + 2 * max(0, (k - 1) // 2 + d - pad), 17 + 2 * max(0, (k - 1) // 2 + d - pad)
    nch, oh, ow = O.corr_output_shape(C, H, W, pad, k, d, s1, s2)
    a = seeded_rand((2, C, H, W), 26, -1, 1).requires_grad_(True)
    b = seeded_rand((2, C, H, W), 27, -1, 1).requires_grad_(True)
    go = seeded_rand((2, nch, oh, ow), 28, -1, 1)
    O.correlation(a, b, pad, k, d, s1, s2, 1.5).backward(go)
    g1, g2 = ops.correlation_backward(a.detach().to(dev), b.detach().to(dev), go.to(dev), pad, k, d, s1, s2, 1.5)
    assert torch.allclose(g1.cpu(), a.grad, rtol=1e-4, atol=2e-5) and torch.allclose(g2.cpu(), b.grad, rtol=1e-4, atol=2e-5)
    h1, h2 = ops.correlation_backward(a.detach().half().to(dev), b.detach().half().to(dev), go.half().to(dev), pad, k, d, s1, s2, 1.5)
    assert h1.dtype == torch.float16
    assert (h1.float().cpu() - a.grad).abs().max().item() <= 2e-2 * max(1.0, a.grad.abs().max().item())
    assert (h2.float().cpu() - b.grad).abs().max().item() <= 2e-2 * max(1.0, b.grad.abs().max().item())
    # the forward of the same configuration, for completeness of the pair
    y = ops.correlation(a.detach().to(dev), b.detach().to(dev), pad, k, d, s1, s2, 1.5)
    with torch.no_grad():
        assert torch.allclose(y.cpu(), O.correlation(a, b, pad, k, d, s1, s2, 1.5), rtol=1e-4, atol=1e-5)


def test_warp_backward_deterministic_fixed_point(dev):
    """pwc_warp_bwd with a workspace accumulates the scatter into grad_x as 64-bit fixed-point integers: bit-identical run
    after run (float atomics are not), equal to the float-atomic path and to autograd through the oracle within fp32
    rounding; convergent flows (many output pixels sampling the same source pixel) stress the accumulation."""
    from opticalflow_amd import ops
    B, C, H, W = 2, 6, 24, 40
    x = seeded_rand((B, C, H, W), 60, -1, 1)
    flo = seeded_rand((B, 2, H, W), 61, -3, 3)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flo[1, 0] = (W / 2 - xx) * 0.9 + 0.3                    # image 1: everything samples near the centre column / row
    flo[1, 1] = (H / 2 - yy) * 0.9 + 0.2
    go = seeded_rand((B, C, H, W), 62, -100, 100)
    xd, fd, gd = x.to(dev), flo.to(dev), go.to(dev)
    gx, gf = ops.warp_backward(xd, fd, gd, 1.25, False, 0.9999)
    for _ in range(5):
        gx2, gf2 = ops.warp_backward(xd, fd, gd, 1.25, False, 0.9999)
        assert torch.equal(gx, gx2) and torch.equal(gf, gf2)
    ax, af = ops.warp_backward(xd, fd, gd, 1.25, False, 0.9999, deterministic=False)
    assert torch.equal(af, gf)
    assert (ax - gx).abs().max().item() <= 1e-5 * max(1.0, gx.abs().max().item())
    xr, fr = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    O.warp(xr, fr * 1.25).backward(go)
    assert (gx.cpu() - xr.grad).abs().max().item() <= 1e-5 * max(1.0, xr.grad.abs().max().item())
    assert (gf.cpu() - fr.grad).abs().max().item() <= 1e-4 * max(1.0, fr.grad.abs().max().item())
    z, _ = ops.warp_backward(xd, fd, torch.zeros_like(gd), 1.25, False, 0.9999)          # all-zero grad_out: scale degenerates safely
    assert (z == 0).all()
    # a non-finite grad_out has no fixed-point form -- the call falls back (on the device) to float atomics, so Inf / NaN
    # reach grad_x where the float path and torch's grid_sample backward put them, and nowhere else
    gi = gd.clone()
    gi[0, 2, 5, 7] = float("inf")
    gi[0, 3, 9, 11] = float("nan")
    dx, df = ops.warp_backward(xd, fd, gi, 1.25, False, 0.9999)
    fx, ff = ops.warp_backward(xd, fd, gi, 1.25, False, 0.9999, deterministic=False)
    bad, bad_f = ~torch.isfinite(dx), ~torch.isfinite(fx)
    assert bool(bad.any()) and torch.equal(bad, bad_f)
    assert bool(bad[1].logical_not().all()) and int(bad.sum()) <= 8          # two pixels x four taps, image 0 only
    assert bool(bad[0, [0, 1, 4, 5]].logical_not().all())                     # other channels untouched
    assert (dx[~bad] - gx[~bad]).abs().max().item() <= 1e-4 * max(1.0, gx.abs().max().item())


def test_correlation_module_and_pybind_shim(dev):
    import correlation_cuda
    from opticalflow_amd import Correlation
    a = seeded_rand((1, 16, 16, 32), 30, -1, 1).to(dev)
    b = seeded_rand((1, 16, 16, 32), 31, -1, 1).to(dev)
    m = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    y = m(a, b)
    ref = O.correlation(a.cpu(), b.cpu(), 4, 1, 4, 1, 1, 1)
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=1e-5)
    # pybind-compatible entry point: empty output tensor, normalised (CUDA kernel) semantics
    out = a.new_empty(0)
    assert correlation_cuda.forward(a, b, a.new_empty(0), b.new_empty(0), out, 4, 1, 4, 1, 1, 1) == 1
    assert torch.allclose(out.cpu(), ref / 16.0, rtol=1e-5, atol=1e-6)
    # autograd through the module
    a2 = a.clone().requires_grad_(True)
    m(a2, b).sum().backward()
    ar = a.cpu().clone().requires_grad_(True)
    O.correlation(ar, b.cpu(), 4, 1, 4, 1, 1, 1).sum().backward()
    assert torch.allclose(a2.grad.cpu(), ar.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_forward_native_semantics_vs_oracle(dev, precision):
    """What an unchanged script gets from models.pwc_dc_net(path): the native (normalised, /C) correlation
    (correlation_cuda_kernel.cu:104,143), here with align_corners on too (the behaviour published weights were trained
    under).  No reference-generated golden exists for this mode (the reference's CUDA extension cannot run here), so the
    oracle -- whose normalised correlation is its golden-pinned un-normalised one divided by C -- is the checker."""
    import models
    from opticalflow_amd.weights import synthetic_state_dict
    import tempfile, os
    net0 = models.pwc_dc_net()
    # normalised cost volumes are ~C times smaller: gain 1.05 instead of 0.85 keeps mean |flow| ~ 1.1 (0.85 gives 0.07)
    sd = synthetic_state_dict(net0.manifest(), seed=5, gain=1.05, bias_std=0.02)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.pth.tar")
        torch.save({"state_dict": sd}, path)
        net = models.pwc_dc_net(path, align_corners=True, precision=precision).to(dev).eval()
    assert net.normalize_corr is True
    x = seeded_rand((2, 6, 128, 192), 77)
    with torch.no_grad():
        ref = O.pwc_forward(sd, x, normalize_corr=True, align_corners=True)
    got = net(x.to(dev)).cpu()
    epe, scale = O.epe(got, ref), ref.abs().mean().item()
    print("native-semantics forward [%s]: EPE %.3e, mean|flow| %.3f" % (precision, epe, scale))
    assert 0.5 < scale < 3.0
    # fp16: these gain-1.05 weights amplify rounding noise 2.5x more than the golden (gain 0.85) ones -- the CPU emulation of
    # nothing but the half roundings (tests/f16_error_budget.py, normalised mode) gives 3.8e-3 here; measured 3.78e-3
    assert epe < (1e-3 if precision == "fp32" else 4.5e-3 * scale)
    # the drop-in Correlation module is the native operator too
    from models.correlation_package.correlation import Correlation
    a, b = seeded_rand((1, 16, 16, 32), 30, -1, 1), seeded_rand((1, 16, 16, 32), 31, -1, 1)
    y = Correlation(4, 1, 4, 1, 1, 1)(a.to(dev), b.to(dev)).cpu()
    assert torch.allclose(y, O.correlation(a, b, 4, 1, 4, 1, 1, 1, normalize=True), rtol=1e-5, atol=1e-6)


def test_shim_classes_native_default_and_onnx_flag_through_shim(dev):
    """ADVICE r2 / VERDICT r2 weak #11.  (1) The reference's scripts do `from models.PWCNet import PWCDCNet; PWCDCNet();
    load_state_dict(ckpt)` (inference_kitti.py:301): that path must give the native (/C) cost volumes -- checked against the
    oracle with normalize_corr=True.  (2) `corr_mod.USE_ONNX_CORRELATION = True` set the way pth2onnx.py:44-46 sets it (on
    models.correlation_package.correlation) switches both the drop-in Correlation and the net to the un-normalised
    expression, as in the reference (correlation.py:103-110)."""
    from models.PWCNet import PWCDCNet
    import models.correlation_package.correlation as corr_mod
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet()
    sd = synthetic_state_dict(net.manifest(), seed=5, gain=1.05, bias_std=0.02)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    x = seeded_rand((1, 6, 64, 128), 78)
    with torch.no_grad():
        ref_native = O.pwc_forward(sd, x, normalize_corr=True)
        ref_raw = O.pwc_forward(sd, x, normalize_corr=False)
    assert O.epe(ref_native, ref_raw) > 0.05                      # the two semantics are far apart with these weights
    got = net(x.to(dev)).cpu()
    assert O.epe(got, ref_native) < 1e-3 * max(1.0, ref_native.abs().mean().item())
    m = corr_mod.Correlation(4, 1, 4, 1, 1, 1)
    a, b = seeded_rand((1, 16, 16, 32), 30, -1, 1).to(dev), seeded_rand((1, 16, 16, 32), 31, -1, 1).to(dev)
    corr_mod.USE_ONNX_CORRELATION = True
    try:
        y = m(a, b).cpu()
        assert torch.allclose(y, O.correlation(a.cpu(), b.cpu(), 4, 1, 4, 1, 1, 1), rtol=1e-5, atol=1e-5)
        got_flag = net(x.to(dev)).cpu()
    finally:
        corr_mod.USE_ONNX_CORRELATION = False
    assert O.epe(got_flag, ref_raw) < 1e-3 * max(1.0, ref_raw.abs().mean().item())
    assert torch.allclose(m(a, b).cpu(), O.correlation(a.cpu(), b.cpu(), 4, 1, 4, 1, 1, 1, normalize=True), rtol=1e-5, atol=1e-6)
    assert O.epe(net(x.to(dev)).cpu(), ref_native) < 1e-3 * max(1.0, ref_native.abs().mean().item())


def test_plan_cache_is_bounded_and_training_mode_warns(dev):
    """plans/graphs are cached per input geometry in an LRU of `max_cached_plans`; a training-mode call with
    grad enabled says that the returned flows are detached instead of silently feeding a training loop."""
    import warnings
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet(use_graph=True).to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    net.max_cached_plans = 2
    xs = [seeded_rand((1, 6, h, w), 90 + i).to(dev) for i, (h, w) in enumerate(((64, 64), (64, 128), (128, 128)))]
    first = net(xs[0])
    for x in xs[1:]:
        net(x)
    assert len(net._plans) == 2 and len(net._graphs) == 2 and net._key(xs[0]) not in net._plans
    assert torch.equal(net(xs[0]), first)                       # rebuilt on demand, same bits
    assert len(net._plans) == 2 and net._key(xs[1]) not in net._plans
    net.train()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        outs = net(xs[0])
        assert len(outs) == 5 and not outs[0].requires_grad
        assert any("inference-only" in str(m.message) for m in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            net(xs[0])
        assert not w


def test_corr_and_warp_fp16_storage(dev):
    """PWC_F16 tensors (fp32 accumulation inside): compare with the oracle on the SAME fp16-rounded inputs.
    Tolerance = half-precision rounding of the result (2^-10 relative) + accumulation noise."""
    from opticalflow_amd import ops
    for shp in ((2, 32, 24, 64), (1, 7, 13, 11)):                      # vector path / scalar path
        a = seeded_rand(shp, 120, -1, 1).half()
        b = seeded_rand(shp, 121, -1, 1).half()
        ref = O.correlation(a.float(), b.float(), 4, 1, 4, 1, 1, 1)
        got = ops.correlation(a.to(dev), b.to(dev)).float().cpu()
        assert got.dtype == torch.float32 and ops.correlation(a.to(dev), b.to(dev)).dtype == torch.float16
        assert (got - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
        gotn = ops.correlation(a.to(dev), b.to(dev), normalize=True, leaky_slope=0.1).float().cpu()
        assert (gotn - O.leaky_relu(ref / shp[1])).abs().max().item() <= 2e-3
    x = seeded_rand((2, 9, 14, 32), 122, -1, 1).half()
    flo = seeded_rand((2, 2, 14, 32), 123, -3, 3).half()
    ref = O.warp(x.float(), flo.float() * 1.25)
    got = ops.warp(x.to(dev), flo.to(dev), flow_scale=1.25).float().cpu()
    # coordinates are fp32 in the kernel; values are rounded to half on store
    assert ((got == 0) == (ref == 0)).float().mean().item() > 0.995
    assert (got - ref).abs().max().item() < 2e-3
    with pytest.raises(TypeError):
        ops.correlation(a.double().to(dev), b.double().to(dev))


# ------------------------------------------------------------------ warp
def test_warp_golden(dev):
    from opticalflow_amd import ops
    g = load_golden("g2_warp.npz")
    for name in g["names"]:
        x = torch.from_numpy(g["x_" + name])
        flo = torch.from_numpy(g["flo_" + name])
        ref = torch.from_numpy(g["out_" + name])
        got = ops.warp(x.to(dev), flo.to(dev)).cpu()
        assert ((got == 0) == (ref == 0)).float().mean().item() > 0.999, name
        assert (got - ref).abs().max().item() < 3e-6, (name, (got - ref).abs().max().item())


def test_warp_scale_align_and_strided_operands(dev):
    from opticalflow_amd import ops
    B, C, H, W = 2, 9, 14, 32
    x = seeded_rand((B, C, H, W), 40, -1, 1)
    flo = seeded_rand((B, 2, H, W), 41, -3, 3)
    for ac in (False, True):
        ref = O.warp(x, flo * 2.5, align_corners=ac)
        arena = torch.zeros((B, 20, H, W), device=dev)
        arena[:, 5:7] = flo.to(dev)
        out = torch.zeros((B, 30, H, W), device=dev)
        ops.warp(x.to(dev), arena[:, 5:7], flow_scale=2.5, align_corners=ac, out=out[:, 3:3 + C])
        assert (out[:, 3:3 + C].cpu() - ref).abs().max().item() < 3e-6
        assert (out[:, :3] == 0).all() and (out[:, 3 + C:] == 0).all()
    # zero flow with align_corners=True is the identity
    idt = ops.warp(x.to(dev), torch.zeros(B, 2, H, W, device=dev), align_corners=True).cpu()
    assert (idt - x).abs().max().item() < 1e-5      # (x*2/(W-1)-1+1)/2*(W-1) is x only up to fp32 rounding


@pytest.mark.parametrize("align,thr,scale", [(False, 0.9999, 1.0), (True, 0.999, 2.5)])
def test_warp_backward_matches_autograd_of_oracle(dev, align, thr, scale):
    """pwc_warp_bwd vs torch autograd through the oracle's explicit-bilinear warp (mask constant, PWCNet.py:174-175).
    Flows keep sample points away from integer coordinates by construction of the check: positions whose
    fractional part is within 1e-3 of a pixel boundary are excluded (the derivative is discontinuous there)."""
    from opticalflow_amd import ops
    B, C, H, W = 2, 5, 11, 14
    x = seeded_rand((B, C, H, W), 140, -1, 1)
    flo = seeded_rand((B, 2, H, W), 141, -3, 3)
    go = seeded_rand((B, C, H, W), 142, -1, 1)
    xr, fr = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    O.warp(xr, fr * scale, align_corners=align, mask_threshold=thr).backward(go)
    gx, gf = ops.warp_backward(x.to(dev), flo.to(dev), go.to(dev), scale, align, thr)
    assert torch.allclose(gx.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    sx = (W if not align else W - 1) / (W - 1)
    sy = (H if not align else H - 1) / (H - 1)
    ix = (torch.arange(W).view(1, 1, W) + flo[:, 0] * scale) * sx - (0.0 if align else 0.5)
    iy = (torch.arange(H).view(1, H, 1) + flo[:, 1] * scale) * sy - (0.0 if align else 0.5)
    smooth = (((ix - ix.round()).abs() > 1e-3) & ((iy - iy.round()).abs() > 1e-3)).unsqueeze(1)
    assert smooth.float().mean().item() > 0.99
    assert torch.allclose(gf.cpu() * smooth, fr.grad * smooth, rtol=1e-4, atol=1e-5)
    assert gf.abs().max().item() > 0.1
    # autograd registration: model.warp is differentiable end to end on the device
    from opticalflow_amd import PWCDCNet
    net = PWCDCNet(align_corners=align)
    xd, fd = x.to(dev).requires_grad_(True), flo.to(dev).requires_grad_(True)
    net.warp(xd, fd).backward(go.to(dev))
    xr2, fr2 = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    O.warp(xr2, fr2, align_corners=align).backward(go)
    assert torch.allclose(xd.grad.cpu(), xr2.grad, rtol=1e-4, atol=1e-5)


def test_model_warp_method(dev):
    from opticalflow_amd import PWCDCNet
    net = PWCDCNet()
    x = seeded_rand((1, 4, 8, 16), 50, -1, 1)
    flo = seeded_rand((1, 2, 8, 16), 51, -2, 2)
    assert (net.warp(x.to(dev), flo.to(dev)).cpu() - O.warp(x, flo)).abs().max().item() < 3e-6


# ------------------------------------------------------------------ conv / deconv
CONV_CASES = [
    # (B, Cin, Cout, H, W, stride, dilation)
    (2, 3, 16, 64, 64, 2, 1),
    (1, 16, 16, 32, 96, 1, 1),
    (2, 21, 9, 20, 45, 1, 1),          # Cout <= 16: 16x16x4 MFMA kernel, ragged Cin / Cout / edges
    (3, 16, 16, 120, 200, 1, 1),       # ... 4-row tiles
    (4, 16, 16, 224, 512, 1, 1),       # ... 8-row tiles (conv1aa geometry)
    (10, 16, 16, 224, 500, 1, 1),      # ... 16-row tiles, ragged right edge
    (2, 32, 64, 16, 32, 2, 1),
    (1, 128, 196, 14, 32, 2, 1),
    (1, 196, 196, 7, 16, 1, 1),
    (1, 81, 128, 7, 16, 1, 1),
    (1, 565, 128, 16, 32, 1, 1),
    (1, 128, 128, 24, 40, 1, 2),
    (1, 128, 128, 24, 40, 1, 4),
    (1, 128, 96, 24, 40, 1, 8),
    (1, 96, 64, 40, 72, 1, 16),
    (1, 64, 32, 9, 33, 1, 1),
    (2, 597, 2, 14, 32, 1, 1),
    (1, 32, 2, 16, 32, 1, 1),
]


@pytest.mark.gpu
def test_conv1a_image_kernel_vs_fp64(gpu_device):
    """conv1a (3 -> 16, stride 2, PWCNet.py:52,184) has its own kernel (csrc/pwc_conv_image.hip): full tiles (448x1024), ragged
    tiles (100x152 -> 50x76), the two images as channel slices of one [B,6,H,W] tensor exactly as the plan passes them, no activation,
    and a width the kernel does not take (W % 4 != 0 -> the generic kernel) -- all against fp64 conv2d."""
    from opticalflow_amd import ops, _lib
    g = torch.Generator().manual_seed(77)
    w = torch.randn(16, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    b = torch.randn(16, generator=g) * 0.1
    wp, bd = ops.pack_conv3x3(w.to(gpu_device)), b.to(gpu_device)
    for B, H, W, slope in ((2, 448, 1024, 0.1), (3, 100, 152, 0.1), (1, 64, 128, None), (2, 20, 46, 0.1)):
        x6 = torch.rand(B, 6, H, W, generator=g)
        xd = x6.to(gpu_device)
        for lo in (0, 3):
            got = ops.conv3x3(xd[:, lo:lo + 3], wp, bd, 16, stride=2, leaky_slope=slope).cpu()
            kern = _lib.load().pwc_last_conv_kernel().decode()
            assert ("image_conv_s2_f32" in kern) == (W % 4 == 0), (kern, W)
            ref = F.conv2d(x6[:, lo:lo + 3].double(), w.double(), b.double(), stride=2, padding=1)
            if slope is not None:
                ref = F.leaky_relu(ref, slope)
            assert got.shape == ref.shape
            assert (got.double() - ref).abs().max().item() <= 3e-6 * 27 ** 0.5, (B, H, W, lo)
    # a non-finite pixel poisons exactly the outputs whose 3x3 window contains it (the kernel pads K from 27 to 28 and the
    # pad step used to multiply a REAL image value three columns to the left by a zero filter -- 0 x Inf = NaN outside the window)
    x = torch.rand(1, 3, 64, 128, generator=g)
    x[0, 1, 21, 40] = float("inf")
    x[0, 2, 40, 87] = float("nan")
    got = ops.conv3x3(x.to(gpu_device), wp, bd, 16, stride=2, leaky_slope=0.1).cpu()
    assert "image_conv_s2_f32" in _lib.load().pwc_last_conv_kernel().decode()
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1), 0.1)
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref)) and (~torch.isfinite(ref)).sum() > 0
    ok = torch.isfinite(ref)
    assert (got.double()[ok] - ref[ok]).abs().max().item() <= 3e-6 * 27 ** 0.5


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_vs_torch_cpu(dev, case):
    from opticalflow_amd import ops
    B, cin, cout, H, W, stride, dil = case
    x = seeded_rand((B, cin, H, W), 60, -1, 1)
    w = seeded_rand((cout, cin, 3, 3), 61, -1, 1) * (2.0 / (cin * 9)) ** 0.5
    bias = seeded_rand((cout,), 62, -0.5, 0.5)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=dil, dilation=dil), 0.1)
    wp = ops.pack_conv3x3(w.to(dev))
    got = ops.conv3x3(x.to(dev), wp, bias.to(dev), cout, stride=stride, dilation=dil, leaky_slope=0.1).cpu()
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item()
    assert err <= 3e-6 * (cin * 9) ** 0.5, (case, err)


@pytest.mark.parametrize("case", [(1, 529, 128, 7, 16, True), (2, 565, 32, 14, 32, True), (1, 1010, 2, 14, 32, False),
                                  (3, 213, 96, 9, 33, True), (16, 661, 64, 7, 16, True)])
def test_conv3x3_split_k_route(dev, case):
    """Layers with few output tiles and a long Cin go through the split-K kernel + fixed-order reduction when a
    workspace is supplied (pwc_conv2d_workspace_bytes > 0): same result as torch fp64 within fp32 rounding, equal
    (to summation order) to the unsplit route, bit-identical run to run, arena slices / residual honoured."""
    from opticalflow_amd import ops
    B, cin, cout, H, W, act = case
    need = ops.conv3x3_workspace_bytes(B, cin, H, W, cout)
    assert need > 0, "case is meant to qualify for split-K"
    assert ops.conv3x3_workspace_bytes(16, 565, 112, 256, 128) == 0        # the big level-2 layers never split
    x = seeded_rand((B, cin, H, W), 160, -1, 1)
    w = seeded_rand((cout, cin, 3, 3), 161, -1, 1) * (2.0 / (cin * 9)) ** 0.5
    bias = seeded_rand((cout,), 162, -0.5, 0.5)
    res = seeded_rand((B, cout, H, W), 163, -1, 1)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    if act:
        ref = F.leaky_relu(ref, 0.1)
    ref = ref + res.double()
    ws = torch.empty((need // 4,), device=dev)
    wp = ops.pack_conv3x3(w.to(dev))
    arena = torch.full((B, cout + 5, H, W), 7.0, device=dev)                 # output is a channel slice
    kw = dict(leaky_slope=0.1 if act else None, residual=res.to(dev))
    ops.conv3x3(x.to(dev), wp, bias.to(dev), cout, out=arena[:, 3:3 + cout], workspace=ws, **kw)
    got = arena[:, 3:3 + cout].cpu()
    assert (arena[:, :3] == 7).all() and (arena[:, 3 + cout:] == 7).all()
    tol = 3e-6 * (cin * 9) ** 0.5
    assert (got.double() - ref).abs().max().item() <= tol
    unsplit = ops.conv3x3(x.to(dev), wp, bias.to(dev), cout, **kw).cpu()
    assert (got - unsplit).abs().max().item() <= tol
    again = ops.conv3x3(x.to(dev), wp, bias.to(dev), cout, workspace=ws, **kw).cpu()
    assert torch.equal(again, got)
    small = ops.conv3x3(x.to(dev), wp, bias.to(dev), cout, workspace=ws[:16], **kw).cpu()   # too small -> unsplit
    assert torch.equal(small, unsplit)


def test_conv3x3_arena_slices_residual_no_act(dev):
    from opticalflow_amd import ops
    B, H, W = 2, 16, 32
    arena = torch.zeros((B, 300, H, W), device=dev)
    xin = seeded_rand((B, 200, H, W), 70, -1, 1)
    arena[:, 100:] = xin.to(dev)
    w = seeded_rand((64, 200, 3, 3), 71, -1, 1) * 0.03
    bias = seeded_rand((64,), 72, -0.5, 0.5)
    ops.conv3x3(arena[:, 100:], ops.pack_conv3x3(w.to(dev)), bias.to(dev), 64, out=arena[:, 36:100])
    ref = F.leaky_relu(F.conv2d(xin, w, bias, padding=1), 0.1)
    assert (arena[:, 36:100].cpu() - ref).abs().max().item() < 1e-4
    assert (arena[:, :36] == 0).all()
    assert torch.equal(arena[:, 100:].cpu(), xin)
    # 2-channel head, no activation, residual add (flow2 + dc_conv7(...), PWCNet.py:268)
    w2 = seeded_rand((2, 64, 3, 3), 73, -1, 1) * 0.05
    b2 = seeded_rand((2,), 74, -0.5, 0.5)
    res = seeded_rand((B, 2, H, W), 75, -1, 1)
    got = ops.conv3x3(arena[:, 36:100], ops.pack_conv3x3(w2.to(dev)), b2.to(dev), 2, leaky_slope=None,
                      residual=res.to(dev)).cpu()
    ref2 = F.conv2d(ref, w2, b2, padding=1) + res
    assert (got - ref2).abs().max().item() < 1e-4


@pytest.mark.parametrize("cin,geom", [(2, (2, 7, 16)), (529, (2, 7, 16)), (597, (2, 7, 16)), (101, (9, 28, 60))])
def test_deconv_vs_torch_cpu(dev, cin, geom):
    """small grids take the 16-wave split-Cin kernel, grids > 256 tiles the 8-wave one (last case, ragged edges)"""
    from opticalflow_amd import ops
    B, H, W = geom
    x = seeded_rand((B, cin, H, W), 80, -1, 1)
    w = seeded_rand((cin, 2, 4, 4), 81, -1, 1) * (2.0 / (2 * 16)) ** 0.5 * (0.1 if cin > 2 else 1.0)
    b = seeded_rand((2,), 82, -0.5, 0.5)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    out = torch.zeros((B, 6, 2 * H, 2 * W), device=dev)
    ops.deconv4x4s2(x.to(dev), w.to(dev), b.to(dev), out=out[:, 2:4])
    assert (out[:, 2:4].cpu().double() - ref).abs().max().item() < 3e-6 * (cin * 4) ** 0.5 * 4
    assert (out[:, :2] == 0).all() and (out[:, 4:] == 0).all()


@pytest.mark.parametrize("geom", [(8, 149, 58, 128),     # 64 8-row tiles -> <TH4,KS4> (4 channel groups + LDS sum), ragged rows
                                  (32, 149, 60, 128),    # 256 8-row tiles -> <TH8,KS1>, ragged rows (60 = 7.5 tiles)
                                  (8, 37, 34, 132)])     # two tile columns, the second 4 px wide; ragged rows
def test_streaming_head_deconv_and_fused_entry(dev, geom):
    """Wide levels take the LDS-ring streaming kernel (pwc_stream3x3.hip): head, upfeat and the fused call.
    Every case has a ragged last channel chunk (cin % 4 == 1)."""
    from opticalflow_amd import ops
    B, cin, H, W = geom
    assert ops.head_upfeat_supported(B, H, W)
    x = seeded_rand((B, cin, H, W), 110, -1, 1)
    hw = seeded_rand((2, cin, 3, 3), 111, -1, 1) * 0.05
    hb = seeded_rand((2,), 112, -0.5, 0.5)
    uw = seeded_rand((cin, 2, 4, 4), 113, -1, 1) * 0.05
    ub = seeded_rand((2,), 114, -0.5, 0.5)
    res = seeded_rand((B, 2, H, W), 115, -1, 1)
    torch.set_num_threads(8)
    ref_h = F.conv2d(x, hw, hb, padding=1)
    ref_u = F.conv_transpose2d(x, uw, ub, stride=2, padding=1)
    arena = torch.zeros((B, cin + 7, H, W), device=dev)
    arena[:, 7:] = x.to(dev)
    xin = arena[:, 7:]                                         # batch-strided operand
    hp = ops.pack_conv3x3(hw.to(dev))
    got_h = ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=None, residual=res.to(dev)).cpu()
    assert (got_h - (ref_h + res)).abs().max().item() < 2e-4
    got_l = ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=0.1).cpu()
    assert (got_l - F.leaky_relu(ref_h, 0.1)).abs().max().item() < 2e-4
    got_u = ops.deconv4x4s2(xin, uw.to(dev), ub.to(dev)).cpu()
    assert (got_u - ref_u).abs().max().item() < 2e-4
    nxt = torch.zeros((B, 9, 2 * H, 2 * W), device=dev)
    flow = torch.zeros((B, 2, H, W), device=dev)
    for _ in range(2):
        ops.head_upfeat(xin, hp, hb.to(dev), uw.to(dev), ub.to(dev), flow, nxt[:, 3:5])
    assert (flow.cpu() - ref_h).abs().max().item() < 2e-4
    assert (nxt[:, 3:5].cpu() - ref_u).abs().max().item() < 2e-4
    assert (nxt[:, :3] == 0).all() and (nxt[:, 5:] == 0).all()
    assert not ops.head_upfeat_supported(16, 7, 16)


@pytest.mark.parametrize("geom", [(1, 565, 112, 256),    # predict_flow2 of ONE pair: 28 8-row tiles -> 10 slices of 60 channels, the last one 25
                                  (2, 565, 112, 256),    # two pairs: 56 tiles -> 5 slices of 116 (the last: 101, a ragged chunk)
                                  (3, 149, 58, 128),     # 24 tiles, ragged rows -> 5 slices of 32 (the last: 21)
                                  (1, 53, 34, 132)])     # two tile columns, the second 4 px wide: 10 tiles -> 2 slices of 32 / 21
def test_streaming_head_cin_slices(dev, geom):
    """The 2-channel flow head on a map of 8..63 tiles (predict_flow2 of one or two pairs, PWCNet.py:263) runs the streaming kernel
    on Cin slices (option stream_slice_wgs; partial sums in the caller's workspace, fixed-order reduction) instead of the split-K
    MFMA kernel: against fp64 conv2d, against the MFMA route (option 0), with bias, LeakyReLU and residual, NaN-filled outputs and
    workspace (an unwritten element must not pass), a batch-strided operand, bit-repeatable; without a workspace the MFMA route."""
    from opticalflow_amd import ops, _lib
    B, cin, H, W = geom
    assert not ops.head_upfeat_supported(B, H, W)                   # under the one-pass kernel's 64 tiles
    x = seeded_rand((B, cin, H, W), 310, -1, 1)
    hw = seeded_rand((2, cin, 3, 3), 311, -1, 1) * 0.05
    hb = seeded_rand((2,), 312, -0.5, 0.5)
    res = seeded_rand((B, 2, H, W), 315, -1, 1)
    torch.set_num_threads(8)
    ref_h = F.conv2d(x.double(), hw.double(), hb.double(), padding=1)
    arena = torch.zeros((B, cin + 7, H, W), device=dev)
    arena[:, 7:] = x.to(dev)
    xin = arena[:, 7:]
    hp = ops.pack_conv3x3(hw.to(dev))
    saved = _lib.get_option("stream_slice_wgs")
    bound = 3e-6 * (9 * cin) ** 0.5
    try:
        out = {}
        for mode in (saved or 512, 0):
            _lib.set_option("stream_slice_wgs", mode)
            need = ops.conv3x3_workspace_bytes(B, cin, H, W, 2)
            if mode:
                assert need >= 2 * B * 2 * H * W * 4                # at least two slices' partial sums
            ws = torch.full((max(need, 16) // 4,), float("nan"), device=dev)
            y = torch.full((B, 2, H, W), float("nan"), device=dev)
            ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=None, residual=res.to(dev), out=y, workspace=ws)
            y2 = torch.full((B, 2, H, W), float("nan"), device=dev)
            ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=None, residual=res.to(dev), out=y2, workspace=ws)
            assert torch.equal(y, y2)
            yl = torch.full((B, 2, H, W), float("nan"), device=dev)
            ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=0.1, out=yl, workspace=ws)
            e = (y.cpu().double() - (ref_h + res.double())).abs().max().item()
            el = (yl.cpu().double() - F.leaky_relu(ref_h, 0.1)).abs().max().item()
            print("stream_slice_wgs=%d %s: head max err %.2e / %.2e (bound %.2e)" % (mode, geom, e, el, bound))
            assert e < bound and el < bound
            out[mode] = (y, yl)
        yn = ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=None, residual=res.to(dev))      # no workspace: not sliced
        _lib.set_option("stream_slice_wgs", saved or 512)
        yn2 = ops.conv3x3(xin, hp, hb.to(dev), 2, leaky_slope=None, residual=res.to(dev))
        assert torch.equal(yn, yn2)
        a, b = out[saved or 512], out[0]
        assert not torch.equal(a[0], b[0])                          # the two routes sum in different orders: really two kernels
        for i in range(2):
            assert (a[i] - b[i]).abs().max().item() < 2 * bound
    finally:
        _lib.set_option("stream_slice_wgs", saved)


@pytest.mark.parametrize("geom", [(16, 629, 28, 64),     # level 4 at batch 16: 112 half-filled 4-row tiles -> 5 slices of 128 channels (the last: 117)
                                  (8, 149, 58, 128),     # 120 tiles, ragged rows -> 5 slices of 32 (the last: 21, a ragged chunk)
                                  (10, 37, 34, 132),     # two tile columns, the second 4 px wide; 2 slices of 32 / 5
                                  (2, 597, 56, 128)])    # level 3 of two pairs: 14 tiles, under the one-pass kernel's 64 -> slices only (15 of 40)
def test_streaming_head_upfeat_cin_slices(dev, geom):
    """predict_flowL + upfeatL in one pass (pwc_head_upfeat_ws_fwd) on launches of fewer tiles than the chip has CUs: Cin slices with
    the caller's workspace against fp64 conv2d / conv_transpose2d (PWCNet.py:32-36, 207-209), against the one-pass form (no
    workspace), NaN-filled outputs and workspace, batch-strided operand and output, bit-repeatable; option 0 = one pass."""
    from opticalflow_amd import ops, _lib
    from opticalflow_amd._lib import PwcHipError
    B, cin, H, W = geom
    one_pass = ops.head_upfeat_supported(B, H, W)                   # 64 tiles; below: only with the workspace (at least 4 tiles)
    assert ops.head_upfeat_supported(B, H, W, min_tiles=4)
    x = seeded_rand((B, cin, H, W), 320, -1, 1)
    hw = seeded_rand((2, cin, 3, 3), 321, -1, 1) * 0.05
    hb = seeded_rand((2,), 322, -0.5, 0.5)
    uw = seeded_rand((cin, 2, 4, 4), 323, -1, 1) * 0.05
    ub = seeded_rand((2,), 324, -0.5, 0.5)
    torch.set_num_threads(8)
    ref_h = F.conv2d(x.double(), hw.double(), hb.double(), padding=1)
    ref_u = F.conv_transpose2d(x.double(), uw.double(), ub.double(), stride=2, padding=1)
    arena = torch.zeros((B, cin + 7, H, W), device=dev)
    arena[:, 7:] = x.to(dev)
    xin = arena[:, 7:]
    hp = ops.pack_conv3x3(hw.to(dev))
    need = ops.head_upfeat_workspace_bytes(B, cin, H, W)
    assert need >= 2 * B * 10 * H * W * 4                           # at least two slices of the 2 + 8 planes
    bound = 3e-6 * (9 * cin) ** 0.5
    got = {}
    for tag, ws in (("sliced", torch.full((need // 4,), float("nan"), device=dev)), ("one pass", None),
                    ("short workspace", torch.full((need // 4 - 4,), float("nan"), device=dev))):
        nxt = torch.full((B, 9, 2 * H, 2 * W), float("nan"), device=dev)
        flow = torch.full((B, 2, H, W), float("nan"), device=dev)
        if tag != "sliced" and not one_pass:
            with pytest.raises(PwcHipError):                       # nothing launched: the caller takes its other kernels
                ops.head_upfeat(xin, hp, hb.to(dev), uw.to(dev), ub.to(dev), flow, nxt[:, 3:5], workspace=ws)
            assert torch.isnan(flow).all() and torch.isnan(nxt).all()
            continue
        ops.head_upfeat(xin, hp, hb.to(dev), uw.to(dev), ub.to(dev), flow, nxt[:, 3:5], workspace=ws)
        flow2, nxt2 = torch.empty_like(flow), torch.empty_like(nxt)
        ops.head_upfeat(xin, hp, hb.to(dev), uw.to(dev), ub.to(dev), flow2, nxt2[:, 3:5], workspace=ws)
        assert torch.equal(flow, flow2) and torch.equal(nxt[:, 3:5], nxt2[:, 3:5])
        assert torch.isnan(nxt[:, :3]).all() and torch.isnan(nxt[:, 5:]).all()
        eh, eu = (flow.cpu().double() - ref_h).abs().max().item(), (nxt[:, 3:5].cpu().double() - ref_u).abs().max().item()
        print("head + upfeat %s %s: max err %.2e / %.2e (bound %.2e)" % (geom, tag, eh, eu, bound))
        assert eh < bound and eu < bound
        got[tag] = (flow, nxt[:, 3:5].clone())
    if one_pass:
        assert not torch.equal(got["sliced"][1], got["one pass"][1])    # other summation order: the slices really ran
        assert torch.equal(got["short workspace"][0], got["one pass"][0]) and torch.equal(got["short workspace"][1], got["one pass"][1])
    saved = _lib.get_option("stream_slice_wgs")
    try:
        _lib.set_option("stream_slice_wgs", 0)
        assert ops.head_upfeat_workspace_bytes(B, cin, H, W) == 0
    finally:
        _lib.set_option("stream_slice_wgs", saved)


# ------------------------------------------------------------------ full forward
def _golden_net(dev, **kw):
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    g = load_golden("g3_forward.npz")
    net = PWCDCNet(**kw)
    sd = synthetic_state_dict(net.manifest(), seed=int(g["wseed"]), gain=float(g["gain"]), bias_std=float(g["bias_std"]))
    net.load_state_dict(sd, strict=True)
    return net.to(dev).eval(), g


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_forward_golden_epe(dev, backend):
    net, g = _golden_net(dev, conv_backend=backend)
    for tag in ("s", "m"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f2 = net(x).cpu()
        ref32 = torch.from_numpy(g["flow2_" + tag])
        ref64 = torch.from_numpy(g["flow2_f64_" + tag])
        e32, e64 = O.epe(f2, ref32), O.epe(f2, ref64)
        print("forward[%s,%s]: EPE vs ref fp32 %.3e, vs ref fp64 %.3e (oracle fp32-vs-fp64 %.3e)"
              % (backend, tag, e32, e64, O.epe(ref32, ref64)))
        assert f2.shape == ref32.shape
        assert e32 < 1e-3 and e64 < 1e-3


def test_forward_golden_large_winograd_route(dev):
    """VERDICT r2 missing #5: the REFERENCE's own outputs (golden g7, generated by oracle/gen_golden.py from models/PWCNet.py:180-273)
    at sizes where the plan takes the Winograd, fused warp+correlation and split-K routes: 4x6x256x512 (level 2 = 64x128) and the
    headline geometry 1x6x448x1024.  Bar 1e-4 mean EPE (observed ~3e-6: fp32 rounding), all five training-mode flows included."""
    net, _ = _golden_net(dev)
    g = load_golden("g7_forward_wino.npz")
    for tag in ("w", "full"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        net.eval()
        f2 = net(x).cpu()
        macs = net._plan_for(x).conv_macs
        assert macs["executed"] < macs["direct"], "the Winograd route was not taken at %s" % (tuple(x.shape),)
        e32, e64 = O.epe(f2, torch.from_numpy(g["flow2_" + tag])), O.epe(f2, torch.from_numpy(g["flow2_f64_" + tag]))
        print("forward[g7 %s]: EPE vs reference fp32 %.3e, fp64 %.3e; %.1f of %.1f GMAC executed"
              % (tag, e32, e64, macs["executed"] / 1e9, macs["direct"] / 1e9))
        assert e32 < 1e-4 and e64 < 1e-4
        net.train()
        with torch.no_grad():
            outs = net(x)
        net.eval()
        assert torch.equal(outs[0].cpu(), f2)
        for lvl, o in zip((3, 4, 5, 6), outs[1:]):
            assert O.epe(o.cpu(), torch.from_numpy(g["train_flow%d_%s" % (lvl, tag)])) < 1e-4, (tag, lvl)


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_forward_old_variant_golden_epe(dev, backend):
    """PWCDCNet_old through the same kernels (filters re-mapped to the arena order) vs the reference's output
    (golden g6) -- bar 1e-3 mean EPE, observed ~1e-6."""
    from opticalflow_amd import pwcnet
    from opticalflow_amd.weights import synthetic_state_dict
    g = load_golden("g6_old.npz")
    net = pwcnet.PWCDCNet_old(conv_backend=backend).to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=int(g["wseed"]), gain=float(g["gain"]),
                                             bias_std=float(g["bias_std"])))
    for tag in ("s", "m"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f2 = net(x).cpu()
        ref = torch.from_numpy(g["flow2_" + tag])
        assert f2.shape == ref.shape
        assert O.epe(f2, ref) < 1e-3, (tag, O.epe(f2, ref))
        assert O.epe(f2, torch.from_numpy(g["flow2_f64_" + tag])) < 1e-3
    net.train()
    outs = net(x)
    for lvl, o in zip((2, 3, 4, 5, 6), outs):
        assert O.epe(o.cpu(), torch.from_numpy(g["train_flow%d_m" % lvl])) < 1e-3, lvl
    xx, ff = torch.from_numpy(g["warp_x"]).to(dev), torch.from_numpy(g["warp_flo"]).to(dev)
    assert (net.warp(xx, ff).cpu() - torch.from_numpy(g["warp_out"])).abs().max().item() < 1e-5


def test_forward_training_tuple_and_graph(dev):
    net, g = _golden_net(dev)
    x = seeded_rand(g["xshape_m"], g["xseed_m"]).to(dev)
    net.train()
    outs = net(x)
    assert len(outs) == 5
    for lvl, o in zip((2, 3, 4, 5, 6), outs):
        assert O.epe(o.cpu(), torch.from_numpy(g["train_flow%d_m" % lvl])) < 1e-3, lvl
    net.eval()
    eager = net(x)
    net.use_graph = True
    a = net(x)
    b = net(x * 0.5)
    c = net(x)
    assert torch.equal(a, eager) and torch.equal(a, c) and not torch.equal(a, b)
    # filling the captured forward's own input buffer in place skips the staging copy and gives the same flow
    xin = net.graph_input(x.shape[0], x.shape[2], x.shape[3])
    assert xin.shape == x.shape and xin.data_ptr() != x.data_ptr()
    xin.copy_(x * 0.5)
    assert torch.equal(net(xin), b)
    net.use_graph = False
    with pytest.raises(RuntimeError):
        net.graph_input(1, 64, 64)


def test_forward_rejects_bad_inputs(dev):
    from opticalflow_amd import PWCDCNet, PwcHipError
    net = PWCDCNet().to(dev).eval()
    with pytest.raises(ValueError):
        net(torch.zeros(1, 6, 100, 128, device=dev))          # not a multiple of 64
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 64, 64, device=dev))
    with pytest.raises(PwcHipError):
        net(torch.zeros(1, 6, 64, 64))                         # CPU tensor: no fallback


# ------------------------------------------------------------------ BASELINE-size checks
def test_conv_level2_geometry_repeatable_and_exact(dev):
    """dc_conv1-like layer (565->128 @112x256) at batch 4: same tile configuration family as the bench;
    repeated launches must be bit-identical (no DMA/barrier race) and match the CPU."""
    from opticalflow_amd import ops
    B, cin, cout, H, W = 4, 565, 128, 112, 256
    x = seeded_rand((B, cin, H, W), 90, -1, 1)
    w = seeded_rand((cout, cin, 3, 3), 91, -1, 1) * (2.0 / (cin * 9)) ** 0.5
    bias = seeded_rand((cout,), 92, -0.5, 0.5)
    xd, wp, bd = x.to(dev), ops.pack_conv3x3(w.to(dev)), bias.to(dev)
    outs = [ops.conv3x3(xd, wp, bd, cout) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    torch.set_num_threads(8)
    ref = F.leaky_relu(F.conv2d(x[:1], w, bias, padding=1), 0.1)
    assert (outs[0][:1].cpu() - ref).abs().max().item() < 3e-6 * (cin * 9) ** 0.5


def test_forward_full_size_batch16(dev):
    """1024x448, batch 16 (the bench workload): items 0 and 15 against the CPU oracle, and a batch made of
    16 copies of one pair must give 16 identical flows (tile/race independence), twice in a row."""
    net, g = _golden_net(dev)
    x = seeded_rand((16, 6, 448, 1024), 1234)
    f = net(x.to(dev)).cpu()
    from opticalflow_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(net.manifest(), seed=int(g["wseed"]), gain=float(g["gain"]), bias_std=float(g["bias_std"]))
    torch.set_num_threads(8)
    for i in (0, 15):
        with torch.no_grad():
            ref = O.pwc_forward(sd, x[i:i + 1])
        e = O.epe(f[i:i + 1], ref)
        print("full-size item %d: mean|flow2| %.3f EPE vs CPU oracle %.3e" % (i, ref.abs().mean().item(), e))
        assert e < 1e-4                 # observed ~3e-6 (fp32 rounding); a localised tile bug of the Winograd route would not hide under this
    same = x[:1].expand(16, -1, -1, -1).contiguous().to(dev)
    for _ in range(2):
        fs = net(same)
        assert all(torch.equal(fs[0], fs[i]) for i in range(1, 16))
        assert torch.equal(fs[0].cpu(), f[0])


# ---- Winograd F(2x2,3x3) route of the 3x3 convolution (pwc_conv3x3_wino_fwd) ------------------------------------------
WINO_CASES = [  # B, Cin, Cout, H, W
    (1, 4, 32, 4, 32),        # one tile group, one chunk
    (2, 5, 7, 9, 13),         # ragged everything: Cin % 4, Cout < 32, odd H and W (scalar stores)
    (1, 16, 128, 12, 64),     # MT=4
    (2, 37, 96, 17, 70),      # MT=1 x 3 cout groups, ragged chunk
    (1, 64, 64, 20, 40),      # MT=2
    (1, 130, 128, 16, 33),    # long K, odd W
    (3, 21, 40, 33, 31),      # CoutP = 64 with 24 padded rows
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd_vs_fp64(gpu_device, case):
    """Same operator as conv3x3 (nn.Conv2d 3x3 pad 1 + LeakyReLU, PWCNet.py:26-33); the bound is the direct kernel's own:
    3e-6 * sqrt(9 Cin) of unit-scale data (the Winograd transforms only add, the accumulation is 2.25x shorter)."""
    from opticalflow_amd import ops
    B, cin, cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    xd, wd, bd = x.to(gpu_device), w.to(gpu_device), b.to(gpu_device)
    up = ops.pack_conv3x3_wino(wd)
    got = ops.conv3x3_wino(xd, up, bd, cout).cpu()
    tol = 3e-6 * (cin * 9) ** 0.5
    assert (got.double() - ref).abs().max().item() <= tol
    # no activation, and agreement with the direct MFMA kernel to the same bound
    lin = ops.conv3x3_wino(xd, up, bd, cout, leaky_slope=None).cpu()
    direct = ops.conv3x3(xd, ops.pack_conv3x3(wd), bd, cout, leaky_slope=None).cpu()
    assert (lin - direct).abs().max().item() <= 2 * tol


@pytest.mark.gpu
def test_conv3x3_winograd_arena_views_and_errors(gpu_device):
    """Channel-slice views of a wider arena on both sides (free batch strides), as the decoder uses them; argument errors."""
    from opticalflow_amd import ops
    g = torch.Generator().manual_seed(5)
    arena = torch.randn(2, 50, 12, 36, generator=g).to(gpu_device)
    w = (torch.randn(32, 20, 3, 3, generator=g) * 0.1).to(gpu_device)
    b = torch.zeros(32, device=gpu_device)
    x = arena[:, 30:50]
    out_arena = torch.full((2, 40, 12, 36), 7.0, device=gpu_device)
    up = ops.pack_conv3x3_wino(w)
    ops.conv3x3_wino(x, up, b, 32, out=out_arena[:, 4:36])
    ref = F.leaky_relu(F.conv2d(x.contiguous(), w, b, padding=1), 0.1)
    assert torch.allclose(out_arena[:, 4:36], ref, rtol=1e-5, atol=2e-5)
    assert bool((out_arena[:, :4] == 7.0).all()) and bool((out_arena[:, 36:] == 7.0).all())      # neighbours untouched
    with pytest.raises(ValueError):
        ops.conv3x3_wino(x, up[:-4], b, 32)
    with pytest.raises(ValueError):
        ops.conv3x3_wino(x, up, b[:-1], 32)
    assert ops.conv3x3_wino_preferred(16, 565, 112, 256, 128) and not ops.conv3x3_wino_preferred(16, 497, 7, 16, 32)
    assert ops.conv3x3_wino_preferred(16, 128, 112, 256, 128, 4) and not ops.conv3x3_wino_preferred(16, 96, 112, 256, 64, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 8, 32, 16, 40, 2), (2, 20, 40, 23, 37, 2), (1, 16, 128, 40, 72, 4), (1, 12, 64, 33, 50, 8), (1, 9, 32, 20, 36, 3)])
def test_conv3x3_winograd_dilated_vs_fp64(gpu_device, case):
    """Dilated layers (dc_conv2..4, PWCNet.py:120-125: padding = dilation) run as D*D lattice convolutions; ragged lattices
    (H, W not multiples of D) and every residue class are covered."""
    from opticalflow_amd import ops
    B, cin, cout, H, W, D = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=D, dilation=D), 0.1)
    got = ops.conv3x3_wino(x.to(gpu_device), ops.pack_conv3x3_wino(w.to(gpu_device)), b.to(gpu_device), cout, dilation=D).cpu()
    assert (got.double() - ref).abs().max().item() <= 3e-6 * (cin * 9) ** 0.5


@pytest.mark.gpu
def test_forward_winograd_route_matches_direct_route(gpu_device, monkeypatch):
    """The whole fp32 forward with the Winograd layers against the same forward on the direct kernels (PWC_CONV_WINO=0)."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    x = torch.rand(4, 6, 256, 512, generator=torch.Generator().manual_seed(11)).to(gpu_device)     # big enough for the rule to say yes
    flows = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PWC_CONV_WINO", flag)
        net = PWCDCNet()
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
        net = net.to(gpu_device).eval()
        flows[flag] = net(x).clone()
        plan = net._plan_for(x)
        assert bool(plan.wino) == (flag == "1")
        # the plan counts the multiplications of its 3x3 layers: fewer than a direct convolution only on the Winograd route
        assert (plan.conv_macs["executed"] < plan.conv_macs["direct"]) == (flag == "1")
    epe = (flows["1"] - flows["0"]).pow(2).sum(1).sqrt().mean().item()
    assert epe < 2e-5, epe


@pytest.mark.gpu
def test_conv3x3_winograd_split_k(gpu_device):
    """A launch of 32..159 workgroups is cut along Cin (levels 5-4 at batch 16); partial sums through the caller's workspace and
    a fixed-order reduction.  Same bound as the unsplit kernel; without a workspace the layer runs unsplit."""
    from opticalflow_amd import ops
    B, cin, cout, H, W = 4, 264, 128, 32, 128
    need = ops.conv3x3_wino_workspace_bytes(B, cin, H, W, cout)
    assert need == 3 * B * cout * H * W * 4                      # 128 workgroups, 66 chunks -> 3 slices of 22
    assert ops.conv3x3_wino_workspace_bytes(B, 80, H, W, cout) == 0      # a short K (20 chunks) is not worth splitting (round 3)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    xd, bd = x.to(gpu_device), b.to(gpu_device)
    up = ops.pack_conv3x3_wino(w.to(gpu_device))
    ws = torch.empty(need // 4, device=gpu_device)
    split = ops.conv3x3_wino(xd, up, bd, cout, workspace=ws).cpu()
    again = ops.conv3x3_wino(xd, up, bd, cout, workspace=ws).cpu()
    unsplit = ops.conv3x3_wino(xd, up, bd, cout).cpu()
    tol = 3e-6 * (cin * 9) ** 0.5
    assert (split.double() - ref).abs().max().item() <= tol and (unsplit.double() - ref).abs().max().item() <= tol
    assert torch.equal(split, again)                              # fixed-order reduction: bit-reproducible
    small = torch.empty(16, device=gpu_device)                    # too small a workspace: unsplit, same result as without one
    assert torch.equal(ops.conv3x3_wino(xd, up, bd, cout, workspace=small).cpu(), unsplit)


# ---- the layers that carry the benchmark step, at their BASELINE shapes ---------------------------------------
# (name, Cin, Cout, H, W, dilation, arena channels, input channel offset, output channel offset or None = own tensor)
BASELINE_LAYERS = [
    ("dc_conv1", 565, 128, 112, 256, 1, 565, 0, None),       # 142 chunks, MT4, 16-byte-store epilogue, XCD remap (grid % 8 == 0)
    ("conv2_2", 373, 96, 112, 256, 1, 565, 192, 96),         # 96 couts = a 64-wide + a 32-wide launch, arena slice in and out
    ("conv2_4", 533, 32, 112, 256, 1, 565, 32, 0),           # MT1: four tile groups per workgroup
    ("dc_conv4", 128, 96, 112, 256, 8, 128, 0, None),        # dilation 8: 64 pixel lattices of 14x32
    ("conv4_3", 533, 64, 28, 64, 1, 629, 32 + 64, 32),       # level 4: 128 workgroups -> split-K through the workspace
    ("conv3_1", 277, 128, 56, 128, 1, 597, 320, 192),        # level 3, 128 couts, ragged last chunk (277 = 69 * 4 + 1)
]


@pytest.mark.gpu
@pytest.mark.parametrize("layer", BASELINE_LAYERS, ids=[l[0] for l in BASELINE_LAYERS])
def test_conv3x3_baseline_shapes_vs_fp64(gpu_device, layer):
    """Each layer is launched exactly as the batch-16 plan launches it (channel-suffix view of the level's arena in, channel slice
    out, shared split-K workspace) and items 0 and 15 are compared with torch's fp64 conv2d on the CPU under the per-element bound
    of test_conv3x3_winograd_vs_fp64; the route the library chose is read back (pwc_last_conv_kernel) and must be the Winograd kernel."""
    from opticalflow_amd import ops, _lib
    name, cin, cout, H, W, D, ctot, in_off, out_off = layer
    B = 16
    gen = torch.Generator(device=gpu_device).manual_seed(1000 + cin + cout)
    arena = torch.randn(B, ctot, H, W, generator=gen, device=gpu_device)
    g = torch.Generator().manual_seed(cin * 7 + cout)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = arena[:, in_off:in_off + cin]
    out = torch.full((B, cout, H, W), 7.0, device=gpu_device) if out_off is None else arena[:, out_off:out_off + cout]
    assert out_off is None or out_off + cout <= in_off            # producer slice precedes the suffix it reads (engine.DENSE_OFF)
    xi = {i: x[i:i + 1].cpu().double() for i in (0, B - 1)}       # inputs captured before the in-place arena write
    assert ops.conv3x3_wino_preferred(B, cin, H, W, cout, D)
    need = ops.conv3x3_wino_workspace_bytes(B, cin, H, W, cout, D)
    ws = torch.empty(max(need, 4) // 4, device=gpu_device)
    ops.conv3x3_wino(x, ops.pack_conv3x3_wino(w.to(gpu_device)), b.to(gpu_device), cout, out=out, dilation=D, workspace=ws)
    kern = _lib.load().pwc_last_conv_kernel().decode()
    assert "wino8" in kern, kern
    if name == "conv4_3":
        assert need > 0                                           # the split-K form was planned (and the workspace given)
    torch.set_num_threads(max(8, torch.get_num_threads()))
    tol = 3e-6 * (cin * 9) ** 0.5
    for i, xd in xi.items():
        ref = F.leaky_relu(F.conv2d(xd, w.double(), b.double(), padding=D, dilation=D), 0.1)
        err = (out[i:i + 1].cpu().double() - ref).abs().max().item()
        print("%s item %d: %s max err %.2e (bound %.2e)" % (name, i, kern, err, tol))
        assert err <= tol, (name, i, err)
    if out_off is not None and out_off > 0:                       # the slice below the output was not touched
        assert bool(torch.isfinite(arena[:, :out_off]).all())


# ---- Winograd F(4x4,3x3) route (pwc_conv3x3_wino4_fwd, round 3): gated by an error budget ------------------------------------------
WINO4_CASES = [  # B, Cin, Cout, H, W
    (1, 4, 32, 8, 64),         # one workgroup of each kind, one chunk
    (2, 5, 7, 9, 12),          # ragged everything: Cin % 4, Cout < 16, H % 4, one partial tile column
    (1, 16, 128, 12, 64),      # two 64-cout workgroups
    (2, 37, 96, 17, 72),       # 64 + 32 couts (both kernel configurations), ragged chunk, ragged rows
    (1, 64, 64, 20, 40),
    (1, 130, 128, 16, 36),     # long K, ragged tile group
    (3, 21, 40, 33, 132),      # CoutP = 64 with 24 padded couts, three column groups
    (1, 565, 128, 24, 64),     # dc_conv1's channel counts
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WINO4_CASES)
def test_conv3x3_winograd4_vs_fp64(gpu_device, case):
    """Same operator as conv3x3 / conv3x3_wino (nn.Conv2d 3x3 pad 1 + LeakyReLU, PWCNet.py:26-33) by F(4x4,3x3).  Error budget
    (DESIGN.md 4b): per element within 1e-6 * sqrt(9 Cin) of an fp64 convolution on unit-scale data -- a third of the bound the
    other fp32 kernels are held to, ~5x what F(2x2) measures -- and within 8.1e-5 of the F(2x2) result (VERDICT r2 item 7's gate:
    3x the 2.7e-5 that F(2x2) differs from the direct kernel on dc_conv1)."""
    from opticalflow_amd import ops
    B, cin, cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    xd, wd, bd = x.to(gpu_device), w.to(gpu_device), b.to(gpu_device)
    got = ops.conv3x3_wino4(xd, ops.pack_conv3x3_wino4(wd), bd, cout).cpu()
    err = (got.double() - ref).abs().max().item()
    f2 = ops.conv3x3_wino(xd, ops.pack_conv3x3_wino(wd), bd, cout).cpu()
    print("F(4x4) %s: max err vs fp64 %.2e (budget %.2e), vs F(2x2) %.2e" % (case, err, 1e-6 * (cin * 9) ** 0.5, (got - f2).abs().max().item()))
    assert err <= 1e-6 * (cin * 9) ** 0.5
    assert (got - f2).abs().max().item() <= 8.1e-5
    lin = ops.conv3x3_wino4(xd, ops.pack_conv3x3_wino4(wd), bd, cout, leaky_slope=None).cpu()
    assert (lin.double() - F.conv2d(x.double(), w.double(), b.double(), padding=1)).abs().max().item() <= 1e-6 * (cin * 9) ** 0.5


def _wino4_rel_errors(gpu_device, x, w, b):
    """max over outputs of |kernel - fp64| / S, S = sum |x||w| + |b| of that output, for F(4x4), F(2x2) and the direct kernel"""
    from opticalflow_amd import ops
    cout = w.shape[0]
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    S = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1) + b.double().abs().view(1, -1, 1, 1)
    xd, wd, bd = x.to(gpu_device), w.to(gpu_device), b.to(gpu_device)
    outs = (ops.conv3x3_wino4(xd, ops.pack_conv3x3_wino4(wd), bd, cout, leaky_slope=None),
            ops.conv3x3_wino(xd, ops.pack_conv3x3_wino(wd), bd, cout, leaky_slope=None),
            ops.conv3x3(xd, ops.pack_conv3x3(wd), bd, cout, leaky_slope=None))
    return [((o.cpu().double() - ref).abs() / S).max().item() for o in outs]


WINO4_REL_BAR = 2.0e-6        # |err| <= 2e-6 x sum|x||w| per output: ~33 ulp of fp32 on a 5 085-term sum (measured <= 1.2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["offset", "relu_like", "heavy_tail"])
def test_conv3x3_winograd4_robust_to_offsets_and_outliers(gpu_device, kind):
    """F(4,3)'s error scales with max |x||w| per tile, not with the rms of the result, and the other F(4x4) tests
    feed zero-mean unit gaussians.  dc_conv1's channel counts (565 -> 128) on (i) x = 8 + randn, a map with a large common-mode offset,
    (ii) a post-LeakyReLU-like positive map, (iii) a gaussian map with 0.1 % of its entries x 50.  Bound: per output relative to
    S = sum |x||w| against fp64 (WINO4_REL_BAR), and no more than 5x the larger of F(2x2)'s figure and the direct kernel's -- measured
    (profiles/r04_wino4_robustness.txt): F(4x4) 1.1e-7 / 2.9e-7 / 1.2e-6, F(2x2) 0.7e-7 / 1.0e-7 / 2.8e-7, direct 1.8e-7 / 2.0e-7 / 5.9e-7:
    no guard (mean subtraction) is needed, the rebalanced interpolation points keep F(4x4) within 2x of the direct fp32 kernel."""
    cin, cout, H, W = 565, 128, 32, 64
    g = torch.Generator().manual_seed(11)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(1, cin, H, W, generator=g)
    if kind == "offset":
        x = x + 8.0
    elif kind == "relu_like":
        x = x.abs() * 3 + 2
    else:
        x[torch.rand(x.shape, generator=g) < 1e-3] *= 50.0
    torch.set_num_threads(max(8, torch.get_num_threads()))
    e4, e2, ed = _wino4_rel_errors(gpu_device, x, w, b)
    print("F(4x4) robustness [%s]: max |err| / sum|x||w| = %.2e (F(2x2) %.2e, direct %.2e)" % (kind, e4, e2, ed))
    assert e4 <= WINO4_REL_BAR
    assert e4 <= 5.0 * max(e2, ed)


@pytest.mark.gpu
def test_conv3x3_winograd4_on_the_plans_own_level2_arena(gpu_device):
    """... and (iv) the real thing: the level-2 arena a forward of the benchmark network leaves behind (565 channels: four LeakyReLU'd
    dense-block outputs, the LeakyReLU'd cost volume, c1, up_flow, up_feat -- very different scales per channel group) through dc_conv1's
    own filters by F(4x4), 32 rows of it against fp64."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet().to(gpu_device).eval()
    sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
    net.load_state_dict(sd)
    xin = torch.rand(1, 6, 448, 1024, generator=torch.Generator().manual_seed(1234)).to(gpu_device)
    net(xin)
    plan = net._plan_for(xin)
    arena = plan.arena[2][:, :565, 40:72].contiguous().cpu()                   # [1, 565, 32, 256] as dc_conv1 reads it
    groups = {"dense": arena[:, :448], "corr": arena[:, 448:529], "c1": arena[:, 529:561], "flow": arena[:, 561:565]}
    print("level-2 arena: " + ", ".join("%s mean|x| %.3f max %.1f" % (k, v.abs().mean().item(), v.abs().max().item()) for k, v in groups.items()))
    w, b = sd["dc_conv1.0.weight"], sd["dc_conv1.0.bias"]
    torch.set_num_threads(max(8, torch.get_num_threads()))
    e4, e2, ed = _wino4_rel_errors(gpu_device, arena, w, b)
    print("F(4x4) on the plan's level-2 arena: max |err| / sum|x||w| = %.2e (F(2x2) %.2e, direct %.2e)" % (e4, e2, ed))
    assert e4 <= WINO4_REL_BAR
    assert e4 <= 5.0 * max(e2, ed)


@pytest.mark.gpu
def test_conv3x3_winograd4_arena_views_rule_and_errors(gpu_device):
    """Channel-slice views of a wider arena on both sides, as the decoder uses them; the measured rule; unsupported geometry is an
    error of the ABI, not a silent fallback."""
    from opticalflow_amd import ops
    from opticalflow_amd._lib import PwcHipError
    g = torch.Generator().manual_seed(6)
    arena = torch.randn(2, 72, 16, 64, generator=g).to(gpu_device)
    w = (torch.randn(32, 40, 3, 3, generator=g) * 0.05).to(gpu_device)
    b = torch.randn(32, generator=g).to(gpu_device) * 0.1
    x = arena[:, 32:72]
    out = arena[:, 0:32]
    ref = F.leaky_relu(F.conv2d(x.contiguous().cpu().double(), w.cpu().double(), b.cpu().double(), padding=1), 0.1)
    keep = arena[:, 32:].clone()
    ops.conv3x3_wino4(x, ops.pack_conv3x3_wino4(w), b, 32, out=out)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-6 * (40 * 9) ** 0.5
    assert torch.equal(arena[:, 32:], keep)                                   # the input slice next to the output was not touched
    with pytest.raises(ValueError):
        ops.conv3x3_wino4(x, ops.pack_conv3x3_wino4(w)[:-4], b, 32)
    with pytest.raises(PwcHipError):                                          # W % 4 != 0 has no 16-byte row pieces
        ops.conv3x3_wino4(torch.zeros(1, 40, 8, 30, device=gpu_device), ops.pack_conv3x3_wino4(w), b, 32)
    pref = ops.conv3x3_wino4_preferred
    assert pref(16, 565, 112, 256, 128) and pref(16, 533, 112, 256, 32) and pref(16, 373, 112, 256, 96) and pref(16, 277, 56, 128, 128)
    assert not pref(16, 128, 112, 256, 128, 2) and not pref(16, 16, 224, 512, 16)
    # launches smaller than the chip count with their input-channel slices since round 4 (option "w4_smallsplit"): conv3_2 / conv3_4's 32-cout
    # launch (128 workgroups -> 2 slices) and level 4 (64 workgroups -> 4 slices) are taken; with the option off the round-3 rule is back
    from opticalflow_amd import _lib
    assert pref(16, 405, 56, 128, 96) and pref(16, 565, 56, 128, 32) and pref(16, 533, 28, 64, 64)
    _lib.set_option("w4_smallsplit", 0)
    try:
        assert not pref(16, 405, 56, 128, 96) and not pref(16, 565, 56, 128, 32) and not pref(16, 533, 28, 64, 64)
    finally:
        _lib.set_option("w4_smallsplit", 1)


@pytest.mark.gpu
def test_conv3x3_winograd4_baseline_shapes_vs_fp64(gpu_device):
    """dc_conv1 (565->128) and conv2_2 (373->96: a 64-cout and a 32-cout launch) at 16 x 112 x 256 exactly as the plan launches
    them (arena views), items 0 and 15 against fp64 conv2d; the kernel the library launched is read back."""
    from opticalflow_amd import ops, _lib
    B, H, W = 16, 112, 256
    for name, cin, cout, in_off, out_off in (("dc_conv1", 565, 128, 0, None), ("conv2_2", 373, 96, 192, 96)):
        gen = torch.Generator(device=gpu_device).manual_seed(2000 + cin)
        arena = torch.randn(B, 565, H, W, generator=gen, device=gpu_device)
        g = torch.Generator().manual_seed(cin * 3 + cout)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        x = arena[:, in_off:in_off + cin]
        out = torch.empty(B, cout, H, W, device=gpu_device) if out_off is None else arena[:, out_off:out_off + cout]
        xi = {i: x[i:i + 1].cpu().double() for i in (0, B - 1)}
        assert ops.conv3x3_wino4_preferred(B, cin, H, W, cout)
        ops.conv3x3_wino4(x, ops.pack_conv3x3_wino4(w.to(gpu_device)), b.to(gpu_device), cout, out=out)
        kern = _lib.load().pwc_last_conv_kernel().decode()
        assert "wino4p" in kern, kern
        torch.set_num_threads(max(8, torch.get_num_threads()))
        for i, xd in xi.items():
            ref = F.leaky_relu(F.conv2d(xd, w.double(), b.double(), padding=1), 0.1)
            err = (out[i:i + 1].cpu().double() - ref).abs().max().item()
            print("%s item %d: %s max err %.2e (budget %.2e)" % (name, i, kern, err, 1e-6 * (cin * 9) ** 0.5))
            assert err <= 1e-6 * (cin * 9) ** 0.5


@pytest.mark.gpu
def test_conv3x3_winograd4_tail_split_vs_fp64(gpu_device):
    """With a workspace, an F(4x4) launch whose last round of workgroups is partial runs that round's tiles as input-channel slices
    (csrc/pwc_conv_wino4.hip launch_wino4): conv2_3 (469 -> 64 @16x112x256: 896 workgroups = 3.5 rounds, the last 128 tiles as two
    slices -- the last two tile rows of every image) and conv2_2 (373 -> 96: its 64-cout launch) exactly as the plan launches them;
    conv2_4 (448 workgroups = 1.75 rounds) does not split.  Items 0 and 15 against fp64 under the F(4x4) budget; the unsplit tile rows are
    bit-identical to a launch without workspace, the split ones differ only by the summation order; bit-repeatable; independent of
    the batch slot."""
    from opticalflow_amd import ops, _lib
    B, H, W = 16, 112, 256
    assert ops.conv3x3_wino4_workspace_bytes(B, 533, H, W, 32) == 0
    for name, cin, cout in (("conv2_3", 469, 64), ("conv2_2", 373, 96)):
        need = max(ops.conv3x3_wino4_workspace_bytes(B, cin, H, W, cout), ops.conv3x3_wino4_workspace_bytes(B, cin, H, W, min(cout, 64)))
        assert need > 0, name
        ws = torch.empty(need // 4, device=gpu_device)
        gen = torch.Generator(device=gpu_device).manual_seed(3000 + cin)
        arena = torch.randn(B, 565, H, W, generator=gen, device=gpu_device)
        g = torch.Generator().manual_seed(cin * 5 + cout)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        x = arena[:, 565 - cin:]
        out = arena[:, 565 - cin - cout:565 - cin]                     # the slice in front of its input, as in the dense block
        up, bd = ops.pack_conv3x3_wino4(w.to(gpu_device)), b.to(gpu_device)
        ops.conv3x3_wino4(x, up, bd, cout, out=out, workspace=ws)
        kern = _lib.load().pwc_last_conv_kernel().decode()
        assert "wino4p" in kern, kern
        got = out.clone()
        ops.conv3x3_wino4(x, up, bd, cout, out=out, workspace=ws)
        assert torch.equal(out, got)                                     # fixed slice order: repeatable
        plain = ops.conv3x3_wino4(x, up, bd, cout)
        # the tail is the same tile positions of EVERY image (here the last two 8-row tile rows): rows above are bit-identical to the
        # unsplit launch, rows below differ by the summation order of the Winograd-domain products
        assert torch.equal(plain[:, :, :96], got[:, :, :96])
        d = (plain - got).abs().amax(dim=(1, 2, 3))
        assert d.min().item() > 0 and d.max().item() <= 8.1e-5
        # ... so an item's result does not depend on its slot in the batch
        same = x[:1].expand(B, -1, -1, -1).contiguous()
        r = ops.conv3x3_wino4(same, up, bd, cout, workspace=ws)
        assert all(torch.equal(r[0], r[i]) for i in range(1, B))
        torch.set_num_threads(max(8, torch.get_num_threads()))
        for i in (0, B - 1):
            ref = F.leaky_relu(F.conv2d(x[i:i + 1].cpu().double(), w.double(), b.double(), padding=1), 0.1)
            err = (got[i:i + 1].cpu().double() - ref).abs().max().item()
            print("%s item %d: %s max err %.2e (budget %.2e), split vs unsplit %.2e" % (name, i, kern, err, 1e-6 * (cin * 9) ** 0.5, d[i].item()))
            assert err <= 1e-6 * (cin * 9) ** 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 565, 128, 112, 256, False), (1, 469, 64, 112, 256, False), (2, 533, 32, 112, 256, False),
                                  (1, 565, 128, 112, 256, True), (4, 128, 128, 56, 128, True)])
def test_conv3x3_winograd4_small_batch_whole_launch_split(gpu_device, case):
    """Small batches (script_pwc.py and inference_kitti.py run ONE pair): an F(4x4) launch with fewer workgroups than CUs is cut along
    Cin into ksplit slices per tile -- tiles x cout groups x slices cover the chip once -- and wino4_tail_reduce_kernel adds the slices
    in a fixed order (the same machinery as the partial-last-round split, main launch empty).  dc_conv1 at batch 1
    (112 workgroups -> 2 slices), conv2_3 (56 -> 4), conv2_4 at batch 2 (56 -> 4), and the PWC_CONV_SPLIT2 (lattice) store of the context
    network through the reduce kernel: against fp64 under the F(4x4) budget, against the unsplit launch (summation order only),
    bit-repeatable, independent of the batch slot, and the rule now takes these layers."""
    from opticalflow_amd import ops, _lib
    B, cin, cout, H, W, split2 = case
    need = ops.conv3x3_wino4_workspace_bytes(B, cin, H, W, cout)
    assert need > 0 and ops.conv3x3_wino4_preferred(B, cin, H, W, cout)
    _lib.set_option("w4_smallsplit", 0)
    try:
        assert ops.conv3x3_wino4_workspace_bytes(B, cin, H, W, cout) == 0 and not ops.conv3x3_wino4_preferred(B, cin, H, W, cout)
    finally:
        _lib.set_option("w4_smallsplit", 1)
    ws = torch.empty(need // 4, device=gpu_device)
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xd, up, bd = x.to(gpu_device), ops.pack_conv3x3_wino4(w.to(gpu_device)), b.to(gpu_device)
    got = ops.conv3x3_wino4(xd, up, bd, cout, workspace=ws, split2=split2)
    kern = _lib.load().pwc_last_conv_kernel().decode()
    assert "wino4p" in kern and kern.split(",")[3].strip() != "1", kern           # <CB, TG, GW, ksplit, ...>: it did split
    assert torch.equal(ops.conv3x3_wino4(xd, up, bd, cout, workspace=ws, split2=split2), got)
    plain = ops.conv3x3_wino4(xd, up, bd, cout, split2=split2)                 # no workspace: the unsplit launch
    d = (plain - got).abs().max().item()
    assert 0 < d <= 8.1e-5
    nat = ops.lattice_unsplit(got, B, 1) if split2 else got                     # back to NCHW for the comparison
    torch.set_num_threads(max(8, torch.get_num_threads()))
    ref = F.leaky_relu(F.conv2d(x[:1].double(), w.double(), b.double(), padding=1), 0.1)
    err = (nat[:1].cpu().double() - ref).abs().max().item()
    print("small-batch split %s: %s max err %.2e (budget %.2e), split vs unsplit %.2e" % (case, kern, err, 1e-6 * (cin * 9) ** 0.5, d))
    assert err <= 1e-6 * (cin * 9) ** 0.5
    if B > 1:
        same = xd[:1].expand(B, -1, -1, -1).contiguous()
        r = ops.conv3x3_wino4(same, up, bd, cout, workspace=ws, split2=split2)
        n = r.shape[0] // B
        assert all(torch.equal(r[:n], r[i * n:(i + 1) * n]) for i in range(1, B))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 4])
def test_plan_workspace_covers_every_launch(gpu_device, monkeypatch, B):
    """the plan's shared scratch must cover the split-K / tail / whole-launch split of EVERY convolution it launches (the C side
    falls back to the unsplit launch when the buffer is too small -- correct, and slow without anyone noticing).  Every conv call of one
    forward is recorded and its workspace demand compared with what the plan holds; at these batch sizes the lattice-major context
    network and the F(4x4) route with input-channel slices are in use."""
    from opticalflow_amd import PWCDCNet, ops
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet().to(gpu_device).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
    x = torch.rand(B, 6, 448, 1024, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    ref = net(x).clone()
    plan = net._plan_for(x)
    have = plan.workspace.numel() * 4 if plan.workspace is not None else 0
    calls = []
    real = {n: getattr(ops, n) for n in ("conv3x3", "conv3x3_wino", "conv3x3_wino4")}

    def spy(name):
        def f(xx, wp, bias, cout, *a, **kw):
            dil = kw.get("dilation", 1)
            n, cin, h, w = xx.shape
            fn = {"conv3x3": lambda: ops.conv3x3_workspace_bytes(n, cin, h, w, cout, kw.get("stride", 1), dil),
                  "conv3x3_wino": lambda: ops.conv3x3_wino_workspace_bytes(n, cin, h, w, cout, dil),
                  "conv3x3_wino4": lambda: ops.conv3x3_wino4_workspace_bytes(n, cin, h, w, cout)}[name]
            calls.append((name, (n, cin, h, w, cout, dil), fn(), kw.get("workspace") is not None))
            return real[name](xx, wp, bias, cout, *a, **kw)
        return f
    for n in real:
        monkeypatch.setattr(ops, n, spy(n))
    out = plan.run(x) if hasattr(plan, "run") else net(x)
    assert torch.equal(out, ref)
    assert len(calls) > 40 and any(c[0] == "conv3x3_wino4" for c in calls)
    short = [c for c in calls if c[2] > have or (c[2] > 0 and not c[3])]
    assert not short, short
    assert getattr(plan, "ctx_lattice", False), "lattice-major context network expected at batch %d" % B


@pytest.mark.gpu
def test_forward_winograd4_error_budget(gpu_device, monkeypatch):
    """Whole-forward gate of the F(4x4) route on the benchmark workload, 16 x 6 x 448 x 1024 -- where the rule sends
    twelve level-2 / level-3 / pyramid / context layers to F(4x4): EPE of items 0 and 15 against the CPU oracle below 1e-4, the whole
    batch within 5e-5 of the same forward with the route switched off (PWC_CONV_WINO4=0: F(2x2) everywhere), fewer executed
    multiplications with it, bit-repeatable."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    xc = torch.rand(16, 6, 448, 1024, generator=torch.Generator().manual_seed(1234))
    x = xc.to(gpu_device)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PWC_CONV_WINO4", flag)
        net = PWCDCNet()
        sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
        net.load_state_dict(sd)
        net = net.to(gpu_device).eval()
        f = net(x).clone()
        assert torch.equal(net(x), f)
        plan = net._plan_for(x)
        res[flag] = (f.cpu(), plan.conv_macs["executed"], sorted(plan.wino4_packed))
        print("forward 16x448x1024, F(4x4) route %s: %.2f GMAC executed per pair, %d layers on F(4x4) %s"
              % ("on" if flag == "1" else "off", plan.conv_macs["executed"] / 16e9, len(plan.wino4_packed), sorted(plan.wino4_packed)))
    assert len(res["1"][2]) >= 10 and not res["0"][2] and res["1"][1] < 0.85 * res["0"][1]
    # (round 4: launches smaller than the chip run as input-channel slices, so conv3_2 / conv3_4 and levels 4-5 are on the route too)
    assert {"dc_conv1.0", "conv2_1.0", "conv2_2.0", "conv2_4.0", "conv3_1.0", "conv3_2.0", "conv4_0.0"} <= set(res["1"][2])
    d = O.epe(res["1"][0], res["0"][0])
    print("F(4x4) vs F(2x2) forward: EPE %.3e" % d)
    assert d < 5e-5
    torch.set_num_threads(max(8, torch.get_num_threads()))
    for i in (0, 15):
        with torch.no_grad():
            ref = O.pwc_forward(sd, xc[i:i + 1])
        e4, e2 = O.epe(res["1"][0][i:i + 1], ref), O.epe(res["0"][0][i:i + 1], ref)
        print("item %d vs CPU oracle: F(4x4) route EPE %.3e, F(2x2) route %.3e" % (i, e4, e2))
        assert e4 < 1e-4 and e2 < 1e-4


@pytest.mark.gpu
def test_split2_store_and_lattice_unsplit(gpu_device):
    """PWC_CONV_SPLIT2 + pwc_lattice_unsplit_f32 (round 3): a layer stores its result as its four pixel lattices so that the next,
    twice-as-dilated layer of the context network (PWCNet.py:126-131) is a dilation-1 convolution on 4x as many half-size images.
    (1) the split store holds exactly the plain result re-ordered; (2) nested three times and unsplit it is the identity; (3) a
    dilation-2 convolution on the normal layout == a dilation-1 convolution on the split layout (against torch fp64)."""
    from opticalflow_amd import ops
    g = torch.Generator().manual_seed(9)
    B, cin, cout, H, W = 2, 40, 64, 16, 64
    x = torch.randn(B, cin, H, W, generator=g).to(gpu_device)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(gpu_device)
    b = (torch.randn(cout, generator=g) * 0.1).to(gpu_device)
    up = ops.pack_conv3x3_wino4(w)
    plain = ops.conv3x3_wino4(x, up, b, cout)
    split = ops.conv3x3_wino4(x, up, b, cout, split2=True)
    assert split.shape == (4 * B, cout, H // 2, W // 2)
    for bb in range(B):
        for py in range(2):
            for px in range(2):
                assert torch.equal(split[4 * bb + 2 * py + px], plain[bb, :, py::2, px::2])
    assert torch.equal(ops.lattice_unsplit(split, B, 1), plain)
    # three nested splits by hand (the order the context network produces) and the unsplit kernel
    t = torch.randn(1, 3, 16, 32, generator=g).to(gpu_device)
    lat = t
    for _ in range(3):
        n, c, h, ww = lat.shape
        lat = torch.stack([lat[:, :, py::2, px::2] for py in range(2) for px in range(2)], 1).reshape(n * 4, c, h // 2, ww // 2).contiguous()
    assert torch.equal(ops.lattice_unsplit(lat, 1, 3), t)
    # dilation 2 on the normal layout == dilation 1 on the lattices
    w2 = (torch.randn(32, cout, 3, 3, generator=g) * (2.0 / (cout * 9)) ** 0.5).to(gpu_device)
    b2 = torch.zeros(32, device=gpu_device)
    ref = F.leaky_relu(F.conv2d(plain.cpu().double(), w2.cpu().double(), b2.cpu().double(), padding=2, dilation=2), 0.1)
    on_lat = ops.conv3x3_wino4(split, ops.pack_conv3x3_wino4(w2), b2, 32)
    back = ops.lattice_unsplit(on_lat, B, 1).cpu().double()
    assert (back - ref).abs().max().item() <= 1e-6 * (cout * 9) ** 0.5 * max(1.0, plain.abs().max().item())
    with pytest.raises(ValueError):
        ops.conv3x3_wino4(x[:, :, :15], up, b, cout, split2=True)


@pytest.mark.gpu
def test_forward_context_lattice_layout_matches_plain_layout(gpu_device, monkeypatch):
    """The context network in lattice-major layout (engine.PwcPlan._context_lattice) against the same forward with PWC_CTX_LATTICE=0
    at the benchmark geometry: every layer computes the reference's layer on the same values, only the storage order between them
    differs -- the flows agree to fp32 rounding (different kernels / tile shapes), and item 3 is checked against the CPU oracle."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    xc = torch.rand(16, 6, 448, 1024, generator=torch.Generator().manual_seed(4321))
    x = xc.to(gpu_device)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PWC_CTX_LATTICE", flag)
        net = PWCDCNet()
        sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
        net.load_state_dict(sd)
        net = net.to(gpu_device).eval()
        out[flag] = net(x).clone()
        assert net._plan_for(x).ctx_lattice == (flag == "1")
    d = O.epe(out["1"].cpu(), out["0"].cpu())
    print("context network lattice-major vs plain layout: EPE %.3e" % d)
    assert d < 2e-5
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        ref = O.pwc_forward(sd, xc[3:4])
    assert O.epe(out["1"][3:4].cpu(), ref) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("dc", "fp32", 2, 128, 192), ("dc", "fp32", 1, 448, 1024), ("old", "fp32", 3, 64, 128),
                                  ("dc", "fp16-strict", 2, 128, 256)])
def test_forward_c1_in_arena_bit_identical_to_copy(gpu_device, case):
    """Option c1_in_arena (the level features of both images live at the arena's batch stride, the pyramid's last convolution of
    levels 2-5 writes the first image's straight into their arena slot) against the form with dense pyramid buffers and one copy per
    level (PWCNet.py:215 concatenates c1 into the dense block's input): the same kernels on the same values at other addresses --
    every flow of the training tuple bit-identical, eager and as a replayed graph, and the plan really is in the layout asked for."""
    from opticalflow_amd import PWCDCNet, PWCDCNet_old, _lib
    from opticalflow_amd.weights import synthetic_state_dict
    variant, precision, B, H, W = case
    x = torch.rand(B, 6, H, W, generator=torch.Generator().manual_seed(77 + B)).to(gpu_device)
    saved = _lib.get_option("c1_in_arena")
    out = {}
    try:
        for flag in (1, 0):
            _lib.set_option("c1_in_arena", flag)
            net = (PWCDCNet_old if variant == "old" else PWCDCNet)(precision=precision, use_graph=True)
            net.load_state_dict(synthetic_state_dict(net.manifest(), seed=3, gain=0.85, bias_std=0.02))
            net = net.to(gpu_device).eval()
            with torch.no_grad():
                first = net(x).clone()
                again = net(x).clone()                             # a replay of the captured graph
            plan = net._plan_for(x)
            upper = getattr(plan, "upper", plan)
            assert upper.c1_in_arena == bool(flag)
            for l in range(2, 6):
                assert (upper.c1[l].data_ptr() == upper.arena[l][:, upper.arena_base[l] + 81:].data_ptr()) == bool(flag)
            assert torch.equal(first, again)
            out[flag] = (first,) + tuple(t.clone() for t in upper.flow.values())
    finally:
        _lib.set_option("c1_in_arena", saved)
    assert len(out[0]) == len(out[1])
    for a, b in zip(out[1], out[0]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(8, 128, 64, 14, 32), (2, 40, 32, 30, 32), (3, 24, 96, 17, 28)])
def test_conv3x3_winograd4_narrow_maps(gpu_device, case):
    """Maps of <= 32 columns (the 14x32 lattice images dc_conv4 runs on in the lattice-major context network) take the F(4x4) kernel's
    second geometry: tile groups of 2 x 8 tiles (8 rows x 32 columns) instead of 1 x 16.  Same budget as the wide form."""
    from opticalflow_amd import ops, _lib
    B, cin, cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    got = ops.conv3x3_wino4(x.to(gpu_device), ops.pack_conv3x3_wino4(w.to(gpu_device)), b.to(gpu_device), cout).cpu()
    kern = _lib.load().pwc_last_conv_kernel().decode()
    assert "wino4p" in kern and ", 32," in kern, kern              # <CB, TG, GW = 32, ...>
    assert (got.double() - ref).abs().max().item() <= 1e-6 * (cin * 9) ** 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 196, 7, 16), (16, 196, 7, 16), (2, 128, 14, 32), (1, 96, 28, 64), (1, 64, 56, 128), (3, 37, 9, 33), (2, 5, 3, 2)])
def test_corr_small_map_kernel(gpu_device, shape):
    """Launches of at most "corr_small_tiles" 8x32 tiles (levels 6-4 at batch 16, levels 6-3 of a single pair) run
    corr81_small_kernel: a workgroup per (64 pixels, displacement row), channels split over its four waves and added up in a fixed
    order.  Against the oracle (correlation.py:12-40) and the tiled kernel (same operator, other summation order: fp32 rounding
    only), LeakyReLU and an arena slot as the output, odd widths, maps smaller than the displacement range; bit-repeatable and
    independent of the batch slot; pwc_warp_corr81_preferred hands these geometries to warp + correlation."""
    from opticalflow_amd import ops, _lib
    B, C, H, W = shape
    dev = gpu_device
    a = seeded_rand(shape, 310, -1, 1)
    b = seeded_rand(shape, 311, -1, 1)
    if B > 1:
        a[-1], b[-1] = a[0], b[0]                                       # same pair in the first and the last slot
    ad, bd = a.to(dev), b.to(dev)
    arena = torch.full((B, 81 + 3, H, W), 7.0, device=dev)
    ops.correlation(ad, bd, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1, out=arena[:, 1:82])
    got = arena[:, 1:82]
    assert (arena[:, :1] == 7).all() and (arena[:, 82:] == 7).all()
    ref = O.leaky_relu(O.correlation(a, b, 4, 1, 4, 1, 1, 1))
    tol = 2e-6 * C ** 0.5 * max(1.0, ref.abs().max().item())
    assert (got.cpu() - ref).abs().max().item() <= tol
    assert torch.equal(ops.correlation(ad, bd, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1), got)
    if B > 1:
        assert torch.equal(got[0], got[-1])
    assert not ops.warp_correlation_preferred(B, C, H, W)
    _lib.set_option("corr_small_tiles", 0)
    try:
        tiled = ops.correlation(ad, bd, 4, 1, 4, 1, 1, 1.0, leaky_slope=0.1)
        assert ops.warp_correlation_preferred(B, C, H, W)
    finally:
        _lib.set_option("corr_small_tiles", 48)
    d = (tiled - got).abs().max().item()
    assert d <= tol, d
    norm = ops.correlation(ad, bd, 4, 1, 4, 1, 1, 1.0, normalize=True)
    assert (norm.cpu() - O.correlation(a, b, 4, 1, 4, 1, 1, 1) / C).abs().max().item() <= tol / C


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 529, 7, 16), (16, 597, 14, 32), (2, 213, 9, 11), (1, 565, 56, 128)])
def test_head10_conv_and_upsample_entry_vs_fp64(gpu_device, case):
    """Small levels: predict_flowL and upfeatL as ONE 3x3 convolution with 10 output channels (ConvTranspose2d(k4,s2,p1) = a 3x3
    convolution with four output phases per channel, ops.deconv_as_conv3x3) on the matrix cores, finished by pwc_upsample_entry_f32
    (deconvL of the flow + pixel shuffle of the phases into the next level's four arena channels).  Against torch's fp64 conv2d /
    conv_transpose2d (PWCNet.py:207-209), arena-strided output, odd sizes; the whole-forward goldens cover it end to end."""
    from opticalflow_amd import ops
    B, cin, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, h, w, generator=g)
    wf = torch.randn(2, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bf = torch.randn(2, generator=g) * 0.1
    wu = torch.randn(cin, 2, 4, 4, generator=g) * (2.0 / (cin * 4)) ** 0.5
    bu = torch.randn(2, generator=g) * 0.1
    wd = torch.randn(2, 2, 4, 4, generator=g) * 0.3
    bd = torch.randn(2, generator=g) * 0.1
    dev = gpu_device
    w10 = torch.cat((wf, ops.deconv_as_conv3x3(wu)), 0).contiguous()
    b10 = torch.cat((bf, bu.repeat_interleave(4))).contiguous()
    need = ops.conv3x3_workspace_bytes(B, cin, h, w, 10)
    ws = torch.empty(max(need, 16) // 4, device=dev)
    head = ops.conv3x3(x.to(dev), ops.pack_conv3x3(w10.to(dev)), b10.to(dev), 10, leaky_slope=None, workspace=ws)
    arena = torch.full((B, 9, 2 * h, 2 * w), 7.0, device=dev)
    ops.upsample_entry(head, wd.to(dev), bd.to(dev), arena[:, 3:7])
    assert (arena[:, :3] == 7).all() and (arena[:, 7:] == 7).all()
    flow = F.conv2d(x.double(), wf.double(), bf.double(), padding=1)
    up_flow = F.conv_transpose2d(flow, wd.double(), bd.double(), stride=2, padding=1)
    up_feat = F.conv_transpose2d(x.double(), wu.double(), bu.double(), stride=2, padding=1)
    tol = 3e-6 * (cin * 9) ** 0.5
    assert (head[:, 0:2].cpu().double() - flow).abs().max().item() <= tol
    assert (arena[:, 3:5].cpu().double() - up_flow).abs().max().item() <= 4 * tol
    assert (arena[:, 5:7].cpu().double() - up_feat).abs().max().item() <= tol
    with pytest.raises(ValueError):
        ops.upsample_entry(head[:, :9], wd.to(dev), bd.to(dev), arena[:, 3:7])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 196, 196, 7, 16, False), (2, 128, 196, 7, 16, True), (3, 37, 33, 9, 11, False), (1, 529, 128, 12, 8, True),
                                  (2, 64, 96, 5, 16, False), (4, 213, 64, 17, 13, True)])
def test_conv3x3_folded_tile_small_maps(gpu_device, case):
    """Maps of at most 16 columns (pyramid / decoder level 6 at 448x1024: 7x16) run the direct MFMA kernel on a FOLDED tile: the 32
    MFMA columns are 16 pixel columns x two groups of four rows (8 x 16 pixels), so a 7x16 map is one tile at 7/8 use instead of two
    4x32 tiles at 7/16 (conv6a / conv6b 62 -> ~35 us at batch 16).  Unsplit and split-K forms against torch's fp64 conv2d: odd
    widths and heights, ragged channel chunks and cout groups, bias, LeakyReLU, residual, arena-strided output."""
    from opticalflow_amd import ops, _lib
    B, cin, cout, H, W, split = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    dev = gpu_device
    ws = None
    if split:
        need = ops.conv3x3_workspace_bytes(B, cin, H, W, cout)
        assert need > 0
        ws = torch.empty(need // 4, device=dev)
    arena = torch.full((B, cout + 4, H, W), 7.0, device=dev)
    ops.conv3x3(x.to(dev), ops.pack_conv3x3(w.to(dev)), b.to(dev), cout, leaky_slope=0.1, residual=res.to(dev), out=arena[:, 2:2 + cout], workspace=ws)
    kern = _lib.load().pwc_last_conv_kernel().decode()
    if not split:
        assert kern.rstrip(">").split(",")[-1].strip() == "16", kern                   # the folded 8 x 16 tile ran
    assert (arena[:, :2] == 7).all() and (arena[:, 2 + cout:] == 7).all()
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1) + res.double()
    err = (arena[:, 2:2 + cout].cpu().double() - ref).abs().max().item()
    assert err <= 3e-6 * (cin * 9) ** 0.5, (case, err)
    again = torch.empty(B, cout, H, W, device=dev)
    ops.conv3x3(x.to(dev), ops.pack_conv3x3(w.to(dev)), b.to(dev), cout, leaky_slope=0.1, residual=res.to(dev), out=again, workspace=ws)
    assert torch.equal(again, arena[:, 2:2 + cout])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(32, 128, 196, 14, 32), (2, 37, 40, 9, 21), (1, 16, 32, 11, 30)])
def test_conv3x3_stride2_folded_tile(gpu_device, case):
    """Stride-2 layers whose OUTPUT has at most 16 columns (conv6aa: 14x32 -> 7x16) take the folded 8 x 16 tile too; vs fp64 conv2d."""
    from opticalflow_amd import ops, _lib
    B, cin, cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    got = ops.conv3x3(x.to(gpu_device), ops.pack_conv3x3(w.to(gpu_device)), b.to(gpu_device), cout, stride=2, leaky_slope=0.1)
    assert _lib.load().pwc_last_conv_kernel().decode().rstrip(">").split(",")[-1].strip() == "16"
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1), 0.1)
    assert got.shape == ref.shape
    assert (got.cpu().double() - ref).abs().max().item() <= 3e-6 * (cin * 9) ** 0.5


@pytest.mark.gpu
def test_borrow_output_returns_the_plans_buffer(gpu_device):
    """PWCDCNet(borrow_output=True): eval-mode forwards hand out the plan's own flow buffer (no copy) -- same values as the default, and
    the NEXT forward of the same geometry overwrites it, which is the documented contract (bench.py's loop and the sharded gather
    consume each flow before the next forward)."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    x1 = torch.rand(1, 6, 128, 192, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    x2 = torch.rand(1, 6, 128, 192, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    nets = {}
    for borrow in (False, True):
        net = PWCDCNet(borrow_output=borrow).to(gpu_device).eval()
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
        nets[borrow] = net
    with torch.no_grad():
        a1, b1 = nets[False](x1), nets[True](x1)
        assert torch.equal(a1, b1)
        keep = b1.clone()
        b2 = nets[True](x2)
        a2 = nets[False](x2)
    assert torch.equal(a2, b2) and b2.data_ptr() == b1.data_ptr()          # the same buffer, now holding the second flow
    assert torch.equal(a1, keep) and not torch.equal(a1, a2)                # the default mode's first result is untouched
