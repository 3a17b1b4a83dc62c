"""fp16 building blocks (c8 layout + MFMA 3x3 convolution) against torch fp64 on the SAME fp16-rounded operands.

Tolerance: the kernel accumulates in fp32 and rounds the result to half once, so
|err| <= 2^-11 * |ref| (output rounding) + 3e-6 * sqrt(K) (fp32 accumulation); the bound below is 1e-3 * max(1, |ref|).
"""
import pytest
import torch
import numpy as np
import torch.nn.functional as F

from conftest import seeded_rand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(gpu_device):
    from opticalflow_amd import _lib
    _lib.load()
    return gpu_device


def test_c8_layout_round_trip_and_padding(dev):
    from opticalflow_amd import ops_f16 as F16
    x = seeded_rand((2, 13, 5, 7), 500, -2, 2).half().float()             # exactly representable in half
    c8 = F16.to_c8(x.to(dev))
    assert c8.shape == (2, 2, 5, 7, 8) and c8.dtype == torch.float16
    ref = torch.zeros(2, 16, 5, 7)
    ref[:, :13] = x
    assert torch.equal(c8.cpu().float(), ref.view(2, 2, 8, 5, 7).permute(0, 1, 3, 4, 2))
    assert torch.equal(F16.from_c8(c8, 13).cpu(), x)
    with pytest.raises(ValueError):
        F16.from_c8(c8, 17)
    from opticalflow_amd import PwcHipError
    with pytest.raises(PwcHipError):
        F16.to_c8(x)                                                       # CPU tensor


F16_CASES = [
    # (B, Cin, Cout, H, W, stride, dilation, act)
    (2, 16, 16, 20, 45, 1, 1, True),
    (1, 117, 128, 16, 40, 1, 1, True),
    (1, 565, 128, 16, 32, 1, 1, True),          # 71 channel groups: ragged last pair
    (1, 373, 96, 9, 33, 1, 1, True),            # MT = 3
    (1, 533, 32, 12, 64, 1, 1, True),
    (1, 128, 128, 24, 40, 1, 2, True),
    (1, 128, 128, 24, 40, 1, 4, True),
    (1, 128, 96, 24, 40, 1, 8, True),           # 2-slot ring (wide halo)
    (2, 96, 64, 40, 72, 1, 16, True),           # three ky row-sets staged separately
    (2, 3, 16, 33, 70, 2, 1, True),             # image layer: 3 channels padded to one group
    (1, 96, 196, 14, 32, 2, 1, True),           # Cout not a multiple of 8
    (1, 64, 2, 9, 33, 1, 1, False),             # flow head, no activation
    (16, 16, 64, 64, 256, 1, 1, True),          # >= 512 tiles of 16 rows and 64 couts: the 16-row tile (MT2, NT4, 2-slot ring)
    (16, 8, 96, 60, 250, 1, 2, True),           # ... 96 couts (MT3, NT4), dilation 2, ragged edges
]


@pytest.mark.parametrize("case", F16_CASES)
def test_conv3x3_f16_vs_torch(dev, case):
    from opticalflow_amd import ops_f16 as F16
    B, cin, cout, H, W, stride, dil, act = case
    x = seeded_rand((B, cin, H, W), 510, -1, 1).half().float()
    w = (seeded_rand((cout, cin, 3, 3), 511, -1, 1) * (2.0 / (cin * 9)) ** 0.5).half().float()
    bias = seeded_rand((cout,), 512, -0.5, 0.5)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=dil, dilation=dil)
    if act:
        ref = F.leaky_relu(ref, 0.1)
    xc = F16.to_c8(x.to(dev))
    wp = F16.pack_conv3x3_f16(w.to(dev))
    yc = F16.conv3x3_f16(xc, wp, bias.to(dev), cin, cout, stride=stride, dilation=dil, leaky_slope=0.1 if act else None)
    got = F16.from_c8(yc, cout).cpu().double()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (case, err)
    # channels past Cout inside the last group are written as zero (they are inputs of the next layer)
    if cout % 8:
        assert (yc[:, -1, :, :, cout % 8:] == 0).all()
    again = F16.conv3x3_f16(xc, wp, bias.to(dev), cin, cout, stride=stride, dilation=dil, leaky_slope=0.1 if act else None)
    assert torch.equal(again, yc)                                          # deterministic


W8_CASES = [
    # (B, Cin, Cout, H, W, dilation, act) -- grids of >= 256 16-row tiles so that the 8-wave kernel is taken
    (16, 117, 128, 64, 256, 1, True),
    (16, 24, 64, 60, 250, 1, True),             # ragged right / bottom edges, 64 couts
    (8, 40, 96, 112, 250, 2, True),             # 96 couts, dilation 2
    (16, 64, 32, 64, 256, 4, False),            # dilation 4, no activation
    (16, 16, 50, 64, 256, 8, True),             # dilation 8, Cout not a multiple of 8 (zero pad channels in the last group)
    (16, 565, 128, 64, 128, 1, True),           # 71 channel groups: ragged last pair
]


@pytest.mark.parametrize("w8", ["1", "2", "4"])
@pytest.mark.parametrize("case", W8_CASES)
def test_conv3x3_f16_eight_wave_kernel(dev, case, w8, monkeypatch):
    """conv3x3_f16w8_kernel (8 MFMA waves, 16-row tiles, double buffer filled by all waves) against torch fp64, for every
    cout-tile width; same bound as the 5-wave kernel, deterministic, and bit-identical to it (same fp32 summation order)."""
    from opticalflow_amd import _lib, ops_f16 as F16
    B, cin, cout, H, W, dil, act = case
    x = seeded_rand((B, cin, H, W), 610, -1, 1).half().float()
    w = (seeded_rand((cout, cin, 3, 3), 611, -1, 1) * (2.0 / (cin * 9)) ** 0.5).half().float()
    bias = seeded_rand((cout,), 612, -0.5, 0.5)
    xc = F16.to_c8(x.to(dev))
    wp = F16.pack_conv3x3_f16(w.to(dev))
    monkeypatch.setenv("PWC_CONV16F_W8", "0")
    base = F16.conv3x3_f16(xc, wp, bias.to(dev), cin, cout, dilation=dil, leaky_slope=0.1 if act else None)
    assert "conv3x3_f16_kernel" in _lib.load().pwc_last_conv_kernel().decode()
    monkeypatch.setenv("PWC_CONV16F_W8", w8)
    yc = F16.conv3x3_f16(xc, wp, bias.to(dev), cin, cout, dilation=dil, leaky_slope=0.1 if act else None)
    assert "conv3x3_f16w8_kernel" in _lib.load().pwc_last_conv_kernel().decode()
    assert torch.equal(yc, base)
    ref = F.conv2d(x[:2].double(), w.double(), bias.double(), padding=dil, dilation=dil)
    if act:
        ref = F.leaky_relu(ref, 0.1)
    got = F16.from_c8(yc[:2], cout).cpu().double()
    assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    if cout % 8:
        assert (yc[:, -1, :, :, cout % 8:] == 0).all()
    assert torch.equal(F16.conv3x3_f16(xc, wp, bias.to(dev), cin, cout, dilation=dil, leaky_slope=0.1 if act else None), yc)


def test_conv3x3_f16_arena_slices_and_errors(dev):
    """input = channel-group suffix of an arena, output = a group slice of the same arena (the DenseNet concat)."""
    from opticalflow_amd import PwcHipError, ops_f16 as F16
    B, H, W = 2, 16, 32
    cin, cout = 200, 64
    x = seeded_rand((B, cin, H, W), 520, -1, 1).half().float()
    w = (seeded_rand((cout, cin, 3, 3), 521, -1, 1) * 0.03).half().float()
    bias = seeded_rand((cout,), 522, -0.5, 0.5)
    arena = torch.full((B, 40, H, W, 8), 7.0, dtype=torch.float16, device=dev)      # 40 groups = 320 channels
    F16.to_c8(x.to(dev), out=arena[:, 15:])                                           # 25 groups = 200 channels
    wp = F16.pack_conv3x3_f16(w.to(dev))
    F16.conv3x3_f16(arena[:, 15:], wp, bias.to(dev), cin, cout, out=arena[:, 7:15])
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), bias.double(), padding=1), 0.1)
    got = F16.from_c8(arena[:, 7:15].contiguous(), cout).cpu().double()
    assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    assert (arena[:, :7] == 7).all()
    assert torch.equal(F16.from_c8(arena[:, 15:], cin).cpu(), x)                     # batch-strided operand
    with pytest.raises(PwcHipError):
        F16.conv3x3_f16(arena[:, 15:], wp, bias.to(dev), cin, cout, dilation=3)       # PWC-Net has no dilation 3
    with pytest.raises(ValueError):
        F16.conv3x3_f16(arena[:, 15:], wp, bias.to(dev), cin + 8, cout)


def _to_c8_cpu(x: torch.Tensor) -> torch.Tensor:
    B, C, H, W = x.shape
    cg = (C + 7) // 8
    pad = torch.zeros(B, cg * 8, H, W)
    pad[:, :C] = x
    return pad.view(B, cg, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().half()


# small maps take the one-thread-per-output kernel, >= 256 tiles the LDS-tiled one (last two cases, one with ragged edges)
@pytest.mark.parametrize("shape", [(2, 32, 24, 64), (1, 196, 7, 16), (1, 96, 13, 37), (2, 13, 9, 5), (4, 32, 112, 256),
                                   (5, 40, 100, 250)])
def test_correlation_c8_vs_oracle(dev, shape):
    """fp16 cost volume vs the CPU oracle on the same fp16-rounded inputs (fp32 accumulation in the kernel, one
    rounding of the result to half: 1e-3 relative)."""
    from opticalflow_amd import ops_f16 as F16
    from oracle import pwc_oracle as O
    B, C, H, W = shape
    a = seeded_rand(shape, 530, -1, 1).half().float()
    b = seeded_rand(shape, 531, -1, 1).half().float()
    ref = O.correlation(a, b, 4, 1, 4, 1, 1, 1)
    got_c8 = F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), C)
    assert got_c8.shape == (B, 11, H, W, 8)
    got = F16.from_c8(got_c8, 81).cpu()
    assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())
    assert (got_c8[:, 10, :, :, 1:] == 0).all()                                      # channels 81..87
    gotn = F16.from_c8(F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), C, normalize=True, leaky_slope=0.1), 81).cpu()
    assert (gotn - O.leaky_relu(ref / C)).abs().max().item() <= 1e-3
    # arena slot output
    arena = torch.full((B, 20, H, W, 8), 3.0, dtype=torch.float16, device=dev)
    F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), C, out=arena[:, 4:15])
    assert torch.equal(arena[:, 4:15], got_c8) and (arena[:, :4] == 3).all() and (arena[:, 15:] == 3).all()


@pytest.mark.parametrize("align,thr", [(False, 0.9999), (True, 0.999)])
def test_warp_c8_vs_oracle(dev, align, thr):
    from opticalflow_amd import ops_f16 as F16
    from oracle import pwc_oracle as O
    B, C, H, W = 2, 21, 14, 33
    x = seeded_rand((B, C, H, W), 540, -1, 1).half().float()
    flo = seeded_rand((B, 2, H, W), 541, -3, 3).half().float()
    ref = O.warp(x, flo * 1.25, align_corners=align, mask_threshold=thr)
    flo_group = torch.zeros(B, 8, H, W)
    flo_group[:, 2:4] = flo                                                          # (u, v) at channels 2, 3 of the group
    got = F16.from_c8(F16.warp_c8(F16.to_c8(x.to(dev)), F16.to_c8(flo_group.to(dev)), C, flo_channel=2, flow_scale=1.25,
                                  align_corners=align, mask_threshold=thr), C).cpu()
    assert ((got == 0) == (ref == 0)).float().mean().item() > 0.995                 # mask decisions
    assert (got - ref).abs().max().item() < 2e-3
    assert torch.equal(_to_c8_cpu(x), F16.to_c8(x.to(dev)).cpu())


@pytest.mark.parametrize("B,C,H,W", [(2, 21, 14, 34), (1, 32, 8, 6), (3, 128, 28, 64)])
def test_level_entry_fp32_flow_chain(dev, B, C, H, W):
    """pwc_level_entry_c8_f16 (one launch): deconvL applied in fp32 to the fp32 flow of the level above, pixel shuffle of
    the fp32 upfeat phases, c1 copy, warp with the fp32 up_flow.  Checked against torch's conv_transpose2d (fp64) and the
    CPU oracle's warp; operands are batch-strided slices of larger buffers as in the plan's arena."""
    from opticalflow_amd import ops_f16 as F16
    from oracle import pwc_oracle as O
    g = (C + 7) // 8
    c1f = seeded_rand((B, C, H, W), 560, -1, 1).half().float()
    c2f = seeded_rand((B, C, H, W), 561, -1, 1).half().float()
    c1, c2 = F16.to_c8(c1f.to(dev)), F16.to_c8(c2f.to(dev))
    flow = seeded_rand((B, 2, H // 2, W // 2), 562, -3, 3)                           # fp32 flow of the level above
    featp = seeded_rand((B, H // 2, W // 2, 8), 563, -3, 3)                          # upfeat phases, channel co*4 + py*2 + px
    dw = seeded_rand((2, 2, 4, 4), 564, -0.5, 0.5)
    db = seeded_rand((2,), 565, -0.1, 0.1)
    head = torch.zeros(B, 2, H // 2, W // 2, 8)
    head[:, 0, :, :, 0:2] = flow.permute(0, 2, 3, 1)
    head[:, 0, :, :, 2:] = 99.0                                                       # must be ignored
    head[:, 1] = featp
    head = head.to(dev)
    arena = torch.full((B, 3 + g + 1, H, W, 8), 0.25, dtype=torch.float16, device=dev)
    arena[:, 3 + g] = 0
    warped = torch.zeros_like(c2)
    F16.level_entry(c1, c2, head[:, 0:1], head[:, 1:2], dw.to(dev), db.to(dev), C, c1_dst=arena[:, 3:3 + g],
                    flow_group=arena[:, 3 + g:4 + g], out=warped, flow_scale=1.25)
    torch.cuda.synchronize()
    up = F.conv_transpose2d(flow.double(), dw.double(), db.double(), stride=2, padding=1)          # [B,2,H,W]
    fg = arena[:, 3 + g].cpu().float()                                                               # [B,H,W,8]
    assert (fg[..., 0:2] - up.permute(0, 2, 3, 1).float()).abs().max().item() <= 2e-3 * max(1.0, up.abs().max().item())
    feat = featp.view(B, H // 2, W // 2, 2, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, H, W, 2)      # [b, 2y+py, 2x+px, co]
    assert torch.equal(fg[..., 2:4], feat.half().float())
    assert (fg[..., 4:] == 0).all() and (arena[:, :3] == 0.25).all()
    assert torch.equal(arena[:, 3:3 + g], c1)
    ref = O.warp(c2f, up.float() * 1.25)
    got = F16.from_c8(warped, C).cpu()
    assert ((got == 0) == (ref == 0)).float().mean().item() > 0.995                                 # mask decisions
    assert (got - ref).abs().max().item() < 2e-3
    with pytest.raises(Exception):
        F16.level_entry(c1[:, :, :H - 1], c2[:, :, :H - 1], head[:, 0:1], head[:, 1:2], dw.to(dev), db.to(dev), C,
                        c1_dst=arena[:, 3:3 + g, :H - 1], flow_group=arena[:, 3 + g:4 + g, :H - 1], out=warped[:, :, :H - 1])


@pytest.mark.parametrize("cin,cout,H,W", [(565, 2, 16, 40), (533, 16, 9, 33), (32, 2, 24, 64)])
def test_conv3x3_f16_split_filters_fp32_out(dev, cin, cout, H, W):
    """Flow heads: split filters (hi + residual/2^11 in the idle half of the 32-row cout tile) and float32 output.
    Against torch fp64 on the UNROUNDED float32 filters the error must be far below the half-filter kernel's."""
    from opticalflow_amd import ops_f16 as F16
    x = seeded_rand((2, cin, H, W), 570, -1, 1).half().float()
    w = seeded_rand((cout, cin, 3, 3), 571, -1, 1) * (2.0 / (cin * 9)) ** 0.5
    bias = seeded_rand((cout,), 572, -0.5, 0.5)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    xc = F16.to_c8(x.to(dev))
    y32 = F16.conv3x3_f16(xc, F16.pack_conv3x3_f16(w.to(dev), split=True), bias.to(dev), cin, cout, leaky_slope=None,
                          out_f32=True, split_w=True)
    assert y32.dtype == torch.float32 and y32.shape == (2, (cout + 7) // 8, H, W, 8)
    got = y32.permute(0, 1, 4, 2, 3).reshape(2, -1, H, W)[:, :cout].cpu().double()
    err_split = (got - ref).abs().max().item()
    y16 = F16.conv3x3_f16(xc, F16.pack_conv3x3_f16(w.to(dev)), bias.to(dev), cin, cout, leaky_slope=None)
    err_half = (F16.from_c8(y16, cout).cpu().double() - ref).abs().max().item()
    print("split+fp32 err %.2e, half filters + half out err %.2e" % (err_split, err_half))
    assert err_split <= 2e-5 * max(1.0, ref.abs().max().item()) and err_split < 0.1 * err_half
    if cout % 8:
        assert (y32[:, -1, :, :, cout % 8:] == 0).all()
    # wider layers pack too since ABI v8 (16 couts per 32-row tile: twice the rows; test_conv3x3_f16_split_filters_any_width)
    assert F16.pack_conv3x3_f16(seeded_rand((32, 8, 3, 3), 1).to(dev), split=True).numel() == 2 * F16.pack_conv3x3_f16(seeded_rand((32, 8, 3, 3), 1).to(dev)).numel()
    with pytest.raises(ValueError):                                   # a plain bank passed as a split one: size mismatch
        F16.conv3x3_f16(xc, F16.pack_conv3x3_f16(seeded_rand((32, cin, 3, 3), 1).to(dev)), torch.zeros(32, device=dev), cin, 32, split_w=True)


def test_half_stores_saturate_no_inf_nan(dev):
    """SURVEY section 7 risk 2 / ADVICE r1: an un-normalised cost volume over C=196 channels of O(10..30) features exceeds the
    half range.  Every store to half saturates at +-65504 (never inf, so no inf-inf = NaN downstream); values inside
    the range keep the usual 1e-3 relative accuracy; the normalised mode (/C) stays in range."""
    from opticalflow_amd import PWCDCNet, ops_f16 as F16
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    for shape in ((1, 196, 7, 16), (2, 196, 24, 64)):                     # direct kernel / LDS-tiled kernel
        a = seeded_rand(shape, 580, -30, 30).half().float()
        b = a.clone()                                                      # in1 == in2: the zero-displacement channel is sum a^2 ~ 58800..
        b[:, :, ::2] *= 1.5
        b = b.half().float()
        ref = O.correlation(a, b, 4, 1, 4, 1, 1, 1)
        assert ref.abs().max().item() > 65504
        got = F16.from_c8(F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), 196), 81).cpu()
        assert torch.isfinite(got).all() and got.abs().max().item() == 65504.0
        big = ref.abs() >= 65504
        assert (got[big].abs() == 65504).all() and (torch.sign(got[big]) == torch.sign(ref[big])).all()
        assert ((got - ref)[~big].abs() <= 1e-3 * ref[~big].abs() + 0.1).all()      # half rounding + fp32 accumulation of ~1e5-sized terms
        gotn = F16.from_c8(F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), 196, normalize=True), 81).cpu()
        assert (gotn - ref / 196).abs().max().item() <= 1e-3 * (ref / 196).abs().max().item()
    # conv: a layer whose outputs exceed the range
    x = seeded_rand((1, 64, 12, 40), 581, 100, 200).half().float()
    w = torch.full((32, 64, 3, 3), 1.0)
    w[16:] = -1.0
    y = F16.from_c8(F16.conv3x3_f16(F16.to_c8(x.to(dev)), F16.pack_conv3x3_f16(w.to(dev)), torch.zeros(32, device=dev), 64, 32,
                                    leaky_slope=None), 32).cpu()
    assert torch.isfinite(y).all() and (y[:, :16, 2:-2, 2:-2] == 65504).all() and (y[:, 16:, 2:-2, 2:-2] == -65504).all()
    # whole network with exploding activations (gain 2.5 per layer instead of 0.85): finite output, no NaN
    net = PWCDCNet(precision="fp16").to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=3, gain=2.5, bias_std=0.02))
    out = net(seeded_rand((1, 6, 64, 128), 582).to(dev))
    assert torch.isfinite(out).all()


def test_half_stores_keep_nan(dev):
    """the saturating half store must not launder NaN into -65504 (v_med3_f32 with a NaN operand returns the minimum of
    the other two): a NaN in an activation or a filter stays NaN through conv3x3_f16, corr81_c8 and the level-entry kernel, so the
    callers' isfinite checks (bench.py `outputs_finite`, the tests) can see it."""
    from opticalflow_amd import ops_f16 as F16
    x = seeded_rand((1, 16, 12, 40), 590, -1, 1)
    x[0, 3, 5, 17] = float("nan")
    w = seeded_rand((32, 16, 3, 3), 591, -0.2, 0.2)
    y = F16.from_c8(F16.conv3x3_f16(F16.to_c8(x.to(dev)), F16.pack_conv3x3_f16(w.to(dev)), torch.zeros(32, device=dev), 16, 32), 32).cpu()
    assert torch.isnan(y[0, :, 4:7, 16:19]).all() and torch.isfinite(y[0, :, :3]).all()
    yf = F16.conv3x3_f16(F16.to_c8(x.to(dev)), F16.pack_conv3x3_f16(w.to(dev)), torch.zeros(32, device=dev), 16, 32, out_f32=True).cpu()
    assert torch.isnan(yf[0, :, 5, 17, :]).all()
    a = seeded_rand((1, 32, 16, 32), 592, -1, 1)
    b = seeded_rand((1, 32, 16, 32), 593, -1, 1)
    a[0, 7, 8, 9] = float("nan")
    c = F16.from_c8(F16.correlation_c8(F16.to_c8(a.to(dev)), F16.to_c8(b.to(dev)), 32, leaky_slope=0.1), 81).cpu()
    assert torch.isnan(c[0, :, 8, 9]).all() and torch.isfinite(c[0, :, 0, :4]).all()
    wrp = F16.from_c8(F16.warp_c8(F16.to_c8(a.to(dev)), torch.zeros((1, 1, 16, 32, 8), device=dev, dtype=torch.float16), 32), 32).cpu()
    assert torch.isnan(wrp[0, 7]).any() and not torch.isnan(wrp[0, :7]).any() and not torch.isnan(wrp[0, 8:]).any()


# Bars of the half-precision plan (fp32 flow chain, split head filters; see DESIGN.md section 7 and
# profiles/r02_f16_error_budget.txt): the network stores ~25 tensors per level in half (11-bit significands), which
# alone puts the flow ~1e-3 RELATIVE from the fp32 result -- a CPU emulation with nothing but those roundings gives
# 0.6e-3 / 1.0e-3 / 1.4e-3 on the three inputs below (mean |flow| 0.61 / 0.92 / 1.34).
# Measured on MI355X with the fp32 flow chain: 0.63e-3 / 1.11e-3 / 1.61e-3 (round 1, flow chain in half: 0.8 / 1.5 / 2.1).
F16_EPE_BAR = {"s": 1.0e-3, "m": 1.3e-3, "full": 1.9e-3}          # absolute mean EPE, flow2 units
F16_REL_BAR = 1.4e-3                                             # and relative to mean |flow|


def test_forward_fp16_vs_reference_golden(dev):
    """Whole network with half-precision activations/filters (fp32 accumulation, fp32 flow chain) vs the reference's fp32
    output on the golden inputs, and vs the CPU oracle at 1 x 448 x 1024 (BASELINE geometry)."""
    from conftest import load_golden
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    g = load_golden("g3_forward.npz")
    net = PWCDCNet(precision="fp16").to(dev).eval()
    sd = synthetic_state_dict(net.manifest(), seed=int(g["wseed"]), gain=float(g["gain"]), bias_std=float(g["bias_std"]))
    net.load_state_dict(sd)
    for tag in ("s", "m"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f2 = net(x).cpu()
        ref = torch.from_numpy(g["flow2_" + tag])
        assert f2.shape == ref.shape and f2.dtype == torch.float32
        epe, scale = O.epe(f2, ref), ref.abs().mean().item()
        print("fp16 forward [%s]: EPE %.3e, mean|flow| %.3f" % (tag, epe, scale))
        assert epe < F16_EPE_BAR[tag] and epe < F16_REL_BAR * scale, (tag, epe, scale)
    eager = net(x)
    net.use_graph = True
    assert torch.equal(net(x), eager) and torch.equal(net(x), eager)
    net.train()                                     # training mode returns the 5-tuple (PWCNet.py:270-271), as float32
    outs = net(x)
    assert len(outs) == 5 and all(o.dtype == torch.float32 for o in outs)
    for lvl, o in zip((2, 3, 4, 5, 6), outs):
        ref_l = torch.from_numpy(g["train_flow%d_m" % lvl])
        e = O.epe(o.cpu(), ref_l)
        print("fp16 forward [m] flow%d: EPE %.3e, mean|flow| %.3f" % (lvl, e, ref_l.abs().mean().item()))
        assert o.shape == ref_l.shape and e <= 2e-3 * max(ref_l.abs().mean().item(), 1e-1), lvl
    net.eval()
    # BASELINE geometry, one pair, against the CPU oracle (fp32) run here
    xf = torch.rand(1, 6, 448, 1024, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        ref = O.pwc_forward(sd, xf)
    net.use_graph = False
    f2 = net(xf.to(dev)).cpu()
    epe, scale = O.epe(f2, ref), ref.abs().mean().item()
    print("fp16 forward [1x448x1024]: EPE %.3e, mean|flow| %.3f" % (epe, scale))
    assert epe < F16_EPE_BAR["full"] and epe < F16_REL_BAR * scale
    with pytest.raises(ValueError):
        PWCDCNet(precision="bf16")


def test_image_conv_s2_vs_torch(dev):
    """conv1a straight from the float32 image (pair tensor halves are batch-strided views) to c8 halves."""
    from opticalflow_amd import ops_f16 as F16
    x = seeded_rand((2, 6, 37, 70), 550, 0, 1)
    w = seeded_rand((16, 3, 3, 3), 551, -1, 1) * 0.3
    b = seeded_rand((16,), 552, -0.5, 0.5)
    xd = x.to(dev)
    for half in (slice(0, 3), slice(3, 6)):
        ref = F.leaky_relu(F.conv2d(x[:, half].double(), w.double(), b.double(), stride=2, padding=1), 0.1)
        got = F16.from_c8(F16.image_conv_s2(xd[:, half], w.to(dev), b.to(dev)), 16).cpu().double()
        assert got.shape == ref.shape == (2, 16, 19, 35)
        assert (got - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (1, 100, 150), (3, 37, 70), (2, 448, 1024)])
def test_pyramid1_fused_vs_layer_by_layer(dev, B, H, W):
    """conv1a -> conv1aa -> conv1b -> conv2a as ONE kernel (level-1 maps in LDS, halos recomputed per tile) against the same four
    layers run one by one (image_conv_s2 + three MFMA convs) and against torch fp64 with every map rounded to half where
    the kernels round.  The fused conv1a reads the image as halves (the layer-by-layer one in fp32): tolerance 4e-3 of the range;
    odd sizes exercise the zero-padding rule (map values outside the image are ZERO, not convolution results)."""
    from opticalflow_amd import ops_f16 as F16
    x = seeded_rand((B, 6, H, W), 700)                                   # pair tensor: the two images are batch-strided views
    ws = {n: (seeded_rand((co, ci, 3, 3), 701 + k, -1, 1) * (2.0 / (ci * 9)) ** 0.5, seeded_rand((co,), 711 + k, -0.3, 0.3))
          for k, (n, ci, co) in enumerate((("1a", 3, 16), ("1aa", 16, 16), ("1b", 16, 16), ("2a", 16, 32)))}
    xd = x.to(dev)
    dw = {n: (w.to(dev), b.to(dev)) for n, (w, b) in ws.items()}
    packed, bias = F16.pack_pyramid1(*dw["1a"], *dw["1aa"], *dw["1b"], *dw["2a"])
    for half in (slice(0, 3), slice(3, 6)):
        got_c8 = F16.pyramid1_fused(xd[:, half], packed, bias)
        t = F16.image_conv_s2(xd[:, half], *dw["1a"])
        for n, st in (("1aa", 1), ("1b", 1), ("2a", 2)):
            t = F16.conv3x3_f16(t, F16.pack_conv3x3_f16(dw[n][0]), dw[n][1], dw[n][0].shape[1], dw[n][0].shape[0], stride=st)
        assert got_c8.shape == t.shape
        got, ref_hip = F16.from_c8(got_c8, 32).cpu().double(), F16.from_c8(t, 32).cpu().double()
        r = x[:, half].half().double()                                    # torch fp64 statement with the same roundings
        for n, st in (("1a", 2), ("1aa", 1), ("1b", 1), ("2a", 2)):
            wq = ws[n][0].half().double()
            r = F.leaky_relu(F.conv2d(r, wq, ws[n][1].double(), stride=st, padding=1), 0.1).half().double()
        scale = max(1.0, r.abs().max().item())
        assert got.shape == r.shape
        assert (got - r).abs().max().item() <= 2e-3 * scale, (got - r).abs().max().item()
        assert (got - ref_hip).abs().max().item() <= 4e-3 * scale
        assert torch.equal(F16.pyramid1_fused(xd[:, half], packed, bias), got_c8)     # deterministic
    with pytest.raises(ValueError):
        F16.pyramid1_fused(xd[:, :4], packed, bias)


def test_forward_fp16_old_variant(dev):
    """PWCDCNet_old through the half-precision plan.  Levels 6..3 must agree with the fp32 plan of the same model to
    1e-2 relative.  At level 2 of the 64x64 golden input (a 16x16 map) the warp's hard validity threshold (0.999 for this
    variant) lets ONE pixel flip when the flow is rounded to half (a sample point ~1e-3 px from the border), and the
    dilated context network spreads that over the whole tiny map: the final flow is therefore held to 5e-2 x mean |flow|
    against the reference's output (golden g6) -- observed 4e-2 on 's'; the dc variant (threshold 0.9999) meets 1e-2."""
    from conftest import load_golden
    from opticalflow_amd import pwcnet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    g = load_golden("g6_old.npz")
    sd = synthetic_state_dict(pwcnet.PWCDCNet_old().manifest(), seed=int(g["wseed"]), gain=float(g["gain"]), bias_std=float(g["bias_std"]))
    net16 = pwcnet.PWCDCNet_old(precision="fp16").to(dev).eval()
    net32 = pwcnet.PWCDCNet_old().to(dev).eval()
    net16.load_state_dict(sd)
    net32.load_state_dict(sd)
    for tag in ("s", "m"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f16, f32 = net16(x).cpu(), net32(x).cpu()
        p16, p32 = net16._plan_for(x), net32._plan_for(x)
        for l in (6, 5, 4, 3):
            a = p32.flow[l]
            b = p16.head[l][:, 0, :, :, 0:2].permute(0, 3, 1, 2)
            assert (a - b).abs().max().item() <= 1e-2 * max(a.abs().max().item(), 1e-3), (tag, l)
        ref = torch.from_numpy(g["flow2_" + tag])
        assert O.epe(f32, ref) < 1e-3
        epe, scale = O.epe(f16, ref), ref.abs().mean().item()
        print("fp16 old [%s]: EPE %.3e, mean|flow| %.3f" % (tag, epe, scale))
        assert epe <= 5e-2 * scale, (tag, epe, scale)


def test_forward_fp16_full_size_batch16_repeatable(dev):
    """BASELINE geometry (16 x 6 x 448 x 1024): the half-precision plan is bit-reproducible run to run (fixed reduction
    orders, no atomics; exercises the 16-row tiles, 2-slot rings and two-per-CU variants the small cases do not reach) and
    stays within F16_REL_BAR x mean|flow| (+35 % for the batch's worst items) of the fp32 plan of the same model."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    net16 = PWCDCNet(precision="fp16", use_graph=True).to(dev).eval()
    net32 = PWCDCNet(use_graph=True).to(dev).eval()
    sd = synthetic_state_dict(net16.manifest(), seed=0, gain=0.85, bias_std=0.02)
    net16.load_state_dict(sd)
    net32.load_state_dict(sd)
    x = torch.rand(16, 6, 448, 1024, generator=torch.Generator().manual_seed(1234)).to(dev)
    a = net16(x)
    for _ in range(5):
        assert torch.equal(net16(x), a)
    ref = net32(x)
    epe, scale = O.epe(a.cpu(), ref.cpu()), ref.abs().mean().item()
    print("fp16 vs fp32 plan at 16x448x1024: EPE %.3e, mean|flow| %.3f" % (epe, scale))
    assert torch.isfinite(a).all() and epe <= 1.35 * F16_REL_BAR * scale
    # ... and item 7 of the batch against the CPU oracle itself (not only the repo's own fp32 plan)
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        ref7 = O.pwc_forward(sd, x[7:8].cpu())
    e7, s7 = O.epe(a[7:8].cpu(), ref7), ref7.abs().mean().item()
    print("fp16 item 7 of 16 vs CPU oracle: EPE %.3e, mean|flow| %.3f" % (e7, s7))
    assert e7 <= 1.35 * F16_REL_BAR * s7


# ---- strict half-precision mode: north_star's 1e-3 (engine_strict.PwcPlanStrict) ----------------------------------------------------
SPLIT_CASES = [  # B, Cin, Cout, H, W, dilation
    (1, 40, 128, 16, 32, 1),       # 5-wave kernel, 128 couts = 8 split tiles
    (2, 117, 96, 24, 40, 1),       # ragged Cin, 96 couts
    (1, 64, 32, 20, 36, 2),        # dilated
    (1, 128, 64, 40, 72, 8),
    (1, 96, 64, 36, 64, 16),       # row-separated staging of dilation 16
    (16, 72, 128, 64, 128, 1),     # >= 256 16-row tiles: the 8-wave kernel
    (16, 245, 96, 32, 128, 1),     # 8-wave, 96 couts (6 split tiles), ragged chunk
    (16, 64, 32, 64, 128, 1),      # 8-wave, 32 couts, long-K rule
    (4, 20, 7, 9, 13, 1),          # Cout < 16: the free form (heads)
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv3x3_f16_split_filters_any_width(dev, case):
    """PWC_CONV_SPLIT_W for any Cout (round 3; was Cout <= 16): hi + lo halves of every filter in one 32-row MFMA tile, summed in
    the epilogue.  With half INPUTS the result must match an fp64 convolution of those inputs with the UNROUNDED filters to fp32
    accumulation accuracy -- i.e. the filter rounding is gone -- while the plain fp16 kernel is ~2^-11 relative away."""
    from opticalflow_amd import ops_f16 as F16
    B, cin, cout, H, W, D = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, H, W, generator=g).half().float()
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xs = slice(0, 1) if B > 4 else slice(0, B)
    ref = F.conv2d(x[xs].double(), w.double(), b.double(), padding=D, dilation=D)
    xd, wd, bd = F16.to_c8(x.to(dev)), w.to(dev), b.to(dev)
    got = F16.conv3x3_f16(xd, F16.pack_conv3x3_f16(wd, split=True), bd, cin, cout, dilation=D, leaky_slope=None, out_f32=True, split_w=True)
    got = got.permute(0, 1, 4, 2, 3).reshape(B, -1, H, W)[xs, :cout].cpu().double()
    plain = F16.conv3x3_f16(xd, F16.pack_conv3x3_f16(wd), bd, cin, cout, dilation=D, leaky_slope=None, out_f32=True)
    plain = plain.permute(0, 1, 4, 2, 3).reshape(B, -1, H, W)[xs, :cout].cpu().double()
    e_split, e_plain = (got - ref).abs().max().item(), (plain - ref).abs().max().item()
    print("split filters %s: max err %.2e (plain fp16 filters %.2e)" % (case, e_split, e_plain))
    assert e_split <= 3e-6 * (cin * 9) ** 0.5                      # the fp32 kernels' bound: nothing but accumulation order left
    assert e_plain > 8 * e_split                                   # the unsplit kernel carries the filter rounding
    # half output + LeakyReLU + pad channels zero, as the plan stores it
    y = F16.conv3x3_f16(xd, F16.pack_conv3x3_f16(wd, split=True), bd, cin, cout, dilation=D, split_w=True)
    yf = F16.from_c8(y, cout)[xs].cpu().double()
    assert (yf - F.leaky_relu(ref, 0.1)).abs().max().item() <= 1.1 * 2.0 ** -11 * max(1.0, ref.abs().max().item())
    if cout % 8:
        assert bool((y[:, -1, :, :, cout % 8:] == 0).all())


def test_forward_fp16_strict_meets_north_star(dev):
    """`PWCDCNet(precision="fp16-strict")`: mean EPE < 1e-3 ABSOLUTE against the reference's own fp32 outputs (golden g3 `s`, `m`;
    g7 `full` = 1x6x448x1024 and `w` = 4x6x256x512) and against the oracle on a KITTI-sized 384x1280 pair -- north_star's bar, which
    the fast half plan misses (1.15e-3 / 1.65e-3 on `m` / `full`).  CPU emulation of this policy: 0.35 / 0.60 / 0.85 / 0.85e-3."""
    from conftest import load_golden
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    g3, g7 = load_golden("g3_forward.npz"), load_golden("g7_forward_wino.npz")
    net = PWCDCNet(precision="fp16-strict").to(dev).eval()
    fast = PWCDCNet(precision="fp16").to(dev).eval()
    sd = synthetic_state_dict(net.manifest(), seed=int(g3["wseed"]), gain=float(g3["gain"]), bias_std=float(g3["bias_std"]))
    net.load_state_dict(sd)
    fast.load_state_dict(sd)
    for g, tag in ((g3, "s"), (g3, "m"), (g7, "w"), (g7, "full")):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f2 = net(x).cpu()
        ref = torch.from_numpy(g["flow2_" + tag])
        assert f2.shape == ref.shape and f2.dtype == torch.float32
        epe, scale = O.epe(f2, ref), ref.abs().mean().item()
        e_fast = O.epe(fast(x).cpu(), ref)
        print("fp16-strict forward [%s]: EPE %.3e vs the reference (fast fp16 plan %.3e), mean|flow| %.3f" % (tag, epe, e_fast, scale))
        assert epe < 1e-3, (tag, epe)
        assert O.epe(f2, torch.from_numpy(g["flow2_f64_" + tag]).float()) < 1e-3
    # training-mode tuple: flow3..flow6 come from the fp32 part
    xm = seeded_rand(g3["xshape_m"], g3["xseed_m"]).to(dev)
    eager = net(xm)
    net.train()
    with torch.no_grad():
        outs = net(xm)
    net.eval()
    assert len(outs) == 5 and torch.equal(outs[0], eager)
    for lvl, o in zip((3, 4, 5, 6), outs[1:]):
        assert O.epe(o.cpu(), torch.from_numpy(g3["train_flow%d_m" % lvl])) < 1e-4, lvl
    net.use_graph = True
    assert torch.equal(net(xm), eager) and torch.equal(net(xm), eager)      # captured replay = eager, bit for bit
    net.use_graph = False
    # KITTI geometry (375x1242 replicate-padded to 384x1280), smooth synthetic frames like tests/test_kitti.py
    gk = torch.Generator().manual_seed(21)
    base = torch.rand(1, 3, 24, 80, generator=gk)
    big = F.interpolate(base, size=(400, 1296), mode="bicubic", align_corners=False).clamp(0, 1)
    xk = torch.cat([big[:, :, 8:392, 8:1288], big[:, :, 5:389, 12:1292]], 1) + torch.rand(1, 6, 384, 1280, generator=gk) * 0.08
    xk = xk.clamp(0, 1).contiguous()
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        refk = O.pwc_forward(sd, xk)
    ek, sk = O.epe(net(xk.to(dev)).cpu(), refk), refk.abs().mean().item()
    print("fp16-strict forward [KITTI 384x1280]: EPE %.3e vs the CPU oracle (fast %.3e), mean|flow| %.3f"
          % (ek, O.epe(fast(xk.to(dev)).cpu(), refk), sk))
    assert ek < 1e-3


STRICT_REL_BAR = 0.75e-3       # mean EPE / mean |flow2| of the strict mode: measured 0.50-0.60e-3 (printed by the tests)


@pytest.mark.parametrize("gain", [0.88, 0.90])
def test_forward_fp16_strict_larger_motion(dev, gain):
    """Where the strict mode's 1e-3 stops holding.  Its error is RELATIVE to the flow magnitude -- the rounding of
    the stored level-2 / context activations -- so north_star's absolute 1e-3 is met while mean |flow2| stays below ~1.8.  The same
    weights at a higher gain give mean |flow2| 2.5 (gain 0.88) and 4.0 (0.90) on a 256x512 pair: the test reports EPE absolute AND
    relative against the CPU oracle, asserts the relative bound (and that the mode still beats the fast one by the usual factor),
    and documents that the absolute figure exceeds 1e-3 there.  Statement of the range: INTEGRATION.md section 4."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    net = PWCDCNet(precision="fp16-strict").to(dev).eval()
    fast = PWCDCNet(precision="fp16").to(dev).eval()
    sd = synthetic_state_dict(net.manifest(), seed=0, gain=gain, bias_std=0.02)
    net.load_state_dict(sd)
    fast.load_state_dict(sd)
    x = torch.rand(1, 6, 256, 512, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        ref = O.pwc_forward(sd, x)
    mag = ref.abs().mean().item()
    e = O.epe(net(x.to(dev)).cpu(), ref)
    ef = O.epe(fast(x.to(dev)).cpu(), ref)
    print("fp16-strict, gain %.2f: mean|flow2| %.2f, EPE %.3e absolute = %.3e x mean|flow2| (fast mode %.3e = %.3e x)"
          % (gain, mag, e, e / mag, ef, ef / mag))
    assert 2.0 < mag < 6.0                      # the input does have the larger motion the test is about
    assert e < STRICT_REL_BAR * mag
    assert e < 0.75 * ef


def test_forward_fp16_strict_full_size_batch16(dev):
    """16 x 6 x 448 x 1024 (BASELINE configs[3]'s per-GPU shard) in the strict mode: items 0 and 15 against the CPU oracle under
    the 1e-3 bar, bit-repeatable run to run, 16 copies of one pair give 16 identical flows."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    net = PWCDCNet(precision="fp16-strict", use_graph=True).to(dev).eval()
    sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
    net.load_state_dict(sd)
    x = torch.rand(16, 6, 448, 1024, generator=torch.Generator().manual_seed(1234))
    a = net(x.to(dev)).clone()
    assert torch.equal(net(x.to(dev)), a)
    torch.set_num_threads(max(8, torch.get_num_threads()))
    for i in (0, 15):
        with torch.no_grad():
            ref = O.pwc_forward(sd, x[i:i + 1])
        e = O.epe(a[i:i + 1].cpu(), ref)
        print("fp16-strict item %d of 16: EPE %.3e vs CPU oracle, mean|flow| %.3f" % (i, e, ref.abs().mean().item()))
        assert e < 1e-3
    same = x[:1].expand(16, -1, -1, -1).contiguous().to(dev)
    fs = net(same)
    assert all(torch.equal(fs[0], fs[i]) for i in range(1, 16))


@pytest.mark.parametrize("B,C,H,W", [(2, 21, 14, 34), (1, 32, 8, 6), (3, 128, 28, 64), (2, 32, 40, 96), (1, 196, 6, 10)])
def test_level_entry_correlation_fused_equals_two_calls(dev, B, C, H, W):
    """pwc_level_corr81_c8_f16 (VERDICT r3 next #1c): level entry + warp + 81-channel cost volume + LeakyReLU as ONE kernel, the
    warped features in LDS only.  BIT-IDENTICAL to pwc_level_entry_c8_f16 followed by pwc_corr81_c8_f16 (cost volume, flow group,
    c1 slot; neighbours of the arena slots untouched; flows that leave the image, ragged tiles, ragged channel groups, more than
    one step of four groups, both scale modes), and the cost volume agrees with the CPU oracle
    (PWCNet.py:141-177,208-214; correlation.py:12-40) on the half-rounded operands."""
    from opticalflow_amd import ops_f16 as F16
    from oracle import pwc_oracle as O
    g = (C + 7) // 8
    c1f = seeded_rand((B, C, H, W), 660, -1, 1).half().float()
    c2f = seeded_rand((B, C, H, W), 661, -1, 1).half().float()
    c1, c2 = F16.to_c8(c1f.to(dev)), F16.to_c8(c2f.to(dev))
    flow = seeded_rand((B, 2, H // 2, W // 2), 662, -3, 3)
    flow[0, :, : H // 6] *= 5.0                                                       # part of image 0 samples far outside
    featp = seeded_rand((B, H // 2, W // 2, 8), 663, -3, 3)
    dw = seeded_rand((2, 2, 4, 4), 664, -0.5, 0.5).to(dev)
    db = seeded_rand((2,), 665, -0.1, 0.1).to(dev)
    head = torch.zeros(B, 2, H // 2, W // 2, 8)
    head[:, 0, :, :, 0:2] = flow.permute(0, 2, 3, 1)
    head[:, 1] = featp
    head = head.to(dev)

    def arena():
        a = torch.full((B, 1 + 11 + g + 1 + 1, H, W, 8), 0.25, dtype=torch.float16, device=dev)       # [pad | corr 11 | c1 g | flow 1 | pad]
        a[:, 12 + g] = 0
        return a
    two, one = arena(), arena()
    warped = torch.zeros_like(c2)
    kw = dict(flow_scale=1.25)
    F16.level_entry(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=two[:, 12:12 + g], flow_group=two[:, 12 + g:13 + g], out=warped, **kw)
    F16.correlation_c8(c1, warped, C, leaky_slope=0.1, out=two[:, 1:12])
    F16.level_entry_correlation(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=one[:, 12:12 + g], flow_group=one[:, 12 + g:13 + g],
                                out=one[:, 1:12], leaky_slope=0.1, **kw)
    assert torch.equal(one, two)
    assert (one[:, 0] == 0.25).all() and (one[:, 13 + g] == 0.25).all() and torch.equal(one[:, 12:12 + g], c1)
    again = arena()
    F16.level_entry_correlation(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=again[:, 12:12 + g], flow_group=again[:, 12 + g:13 + g],
                                out=again[:, 1:12], leaky_slope=0.1, **kw)
    assert torch.equal(again, one)                                                   # repeatable
    n1 = F16.level_entry_correlation(c1, c2, head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=again[:, 12:12 + g],
                                     flow_group=again[:, 12 + g:13 + g], out=torch.empty_like(one[:, 1:12]).contiguous(), normalize=True, **kw)
    assert torch.equal(n1, F16.correlation_c8(c1, warped, C, normalize=True))
    # against the oracle: up_flow by conv_transpose2d, warp, correlation, LeakyReLU -- the warped features are rounded to half on the way
    up = F.conv_transpose2d(flow, dw.cpu(), db.cpu(), stride=2, padding=1)
    wref = O.warp(c2f, up * 1.25).half().float()
    ref = O.leaky_relu(O.correlation(c1f, wref, 4, 1, 4, 1, 1, 1))
    got = F16.from_c8(one[:, 1:12].contiguous(), 81).cpu()
    tol = 4e-3 * max(1.0, ref.abs().max().item())
    bad = ((got - ref).abs() > tol).float().mean().item()
    assert bad < 2e-3, bad                                                          # a mask decision at the threshold may flip a few pixels
    with pytest.raises(Exception):
        F16.level_entry_correlation(c1[:, :, :H - 1], c2[:, :, :H - 1], head[:, 0:1], head[:, 1:2], dw, db, C, c1_dst=again[:, 12:12 + g, :H - 1],
                                    flow_group=again[:, 12 + g:13 + g, :H - 1], out=again[:, 1:12, :H - 1])


@pytest.mark.parametrize("precision", ["fp16", "fp16-strict"])
def test_forward_fp16_fused_level_entry_same_bits(dev, precision):
    """Option f16_level_corr: the half-precision plans enter every level below the coarsest through pwc_level_corr81_c8_f16 (one
    kernel) instead of level entry + correlation -- the forward's flow must not change by a bit."""
    from opticalflow_amd import PWCDCNet, _lib
    from opticalflow_amd.weights import synthetic_state_dict
    x = torch.rand(2, 6, 192, 320, generator=torch.Generator().manual_seed(21)).to(dev)
    flows = []
    for v in (0, 1):
        _lib.set_option("f16_level_corr", v)
        try:
            net = PWCDCNet(precision=precision).to(dev).eval()
            net.load_state_dict(synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02))
            with torch.no_grad():
                flows.append(net(x).clone())
        finally:
            _lib.set_option("f16_level_corr", 0)
    assert torch.isfinite(flows[0]).all() and torch.equal(flows[0], flows[1])
