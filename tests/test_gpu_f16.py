"""fp16 building blocks (c8 layout + MFMA 3x3 convolution) against torch fp64 on the SAME fp16-rounded operands.

Tolerance: the kernel accumulates in fp32 and rounds the result to half once, so
|err| <= 2^-11 * |ref| (output rounding) + 3e-6 * sqrt(K) (fp32 accumulation); the bound below is 1e-3 * max(1, |ref|).
"""
import pytest
import torch
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
def test_level_entry_equals_shuffle_copy_warp(dev, B, C, H, W):
    """pwc_level_entry_c8_f16 (one launch) == pixel shuffle of both 4-phase tensors + c1 copy + pwc_warp_c8_f16, bit for bit;
    operands are batch-strided slices of larger buffers as in the plan's arena."""
    from opticalflow_amd import ops_f16 as F16
    from opticalflow_amd.engine_f16 import PwcPlanF16
    g = (C + 7) // 8
    c1 = F16.to_c8(seeded_rand((B, C, H, W), 560, -1, 1).to(dev))
    c2 = F16.to_c8(seeded_rand((B, C, H, W), 561, -1, 1).to(dev))
    heads = (seeded_rand((B, 2, H // 2, W // 2, 8), 562, -3, 3).half().to(dev))    # group 1 = upfeat phases (head[l][:, 1:2])
    flowp = (seeded_rand((B, 1, H // 2, W // 2, 8), 563, -3, 3).half().to(dev))    # deconv phases (upflow[l])
    def fresh():
        arena = torch.full((B, 3 + g + 1, H, W, 8), 0.25, dtype=torch.float16, device=dev)
        arena[:, 3 + g] = 0
        return arena, torch.zeros_like(c2)
    a_ref, w_ref = fresh()
    a_ref[:, 3:3 + g].copy_(c1)
    fg = a_ref[:, 3 + g]
    PwcPlanF16._shuffle(flowp[:, 0], fg[..., 0:2])
    PwcPlanF16._shuffle(heads[:, 1], fg[..., 2:4])
    F16.warp_c8(c2, a_ref[:, 3 + g:4 + g], C, flo_channel=0, flow_scale=1.25, out=w_ref)
    a_got, w_got = fresh()
    F16.level_entry(c1, c2, flowp, heads[:, 1:2], C, c1_dst=a_got[:, 3:3 + g], flow_group=a_got[:, 3 + g:4 + g], out=w_got,
                    flow_scale=1.25)
    torch.cuda.synchronize()
    assert torch.equal(a_got, a_ref) and torch.equal(w_got, w_ref)
    assert (a_got[:, :3] == 0.25).all() and (a_got[:, 3 + g, ..., 4:] == 0).all()
    with pytest.raises(Exception):
        F16.level_entry(c1[:, :, :H - 1], c2[:, :, :H - 1], flowp, heads[:, 1:2], C, c1_dst=a_got[:, 3:3 + g, :H - 1],
                        flow_group=a_got[:, 3 + g:4 + g, :H - 1], out=w_got[:, :, :H - 1])


def test_forward_fp16_vs_reference_golden(dev):
    """Whole network with half-precision activations/filters (fp32 accumulation) vs the reference's fp32 output on the
    golden inputs.  Bar: mean EPE <= 1e-2 * mean |flow|; observed ~1e-3 relative."""
    from conftest import load_golden
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.weights import synthetic_state_dict
    from oracle import pwc_oracle as O
    g = load_golden("g3_forward.npz")
    net = PWCDCNet(precision="fp16").to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=int(g["wseed"]), gain=float(g["gain"]),
                                             bias_std=float(g["bias_std"])))
    for tag in ("s", "m"):
        x = seeded_rand(g["xshape_" + tag], g["xseed_" + tag]).to(dev)
        f2 = net(x).cpu()
        ref = torch.from_numpy(g["flow2_" + tag])
        assert f2.shape == ref.shape and f2.dtype == torch.float32
        epe, scale = O.epe(f2, ref), ref.abs().mean().item()
        print("fp16 forward [%s]: EPE %.3e, mean|flow| %.3f" % (tag, epe, scale))
        assert epe <= 1e-2 * scale, (tag, epe, scale)
    eager = net(x)
    net.use_graph = True
    assert torch.equal(net(x), eager) and torch.equal(net(x), eager)
    net.train()                                     # training mode returns the 5-tuple (PWCNet.py:270-271), as float32
    outs = net(x)
    assert len(outs) == 5 and all(o.dtype == torch.float32 for o in outs)
    for lvl, o in zip((2, 3, 4, 5, 6), outs):
        ref_l = torch.from_numpy(g["train_flow%d_m" % lvl])
        assert o.shape == ref_l.shape and O.epe(o.cpu(), ref_l) <= 1e-2 * max(ref_l.abs().mean().item(), 1e-2), lvl
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
            b = p16.head[l][:, 0, :, :, 0:2].permute(0, 3, 1, 2).float()
            assert (a - b).abs().max().item() <= 1e-2 * max(a.abs().max().item(), 1e-3), (tag, l)
        ref = torch.from_numpy(g["flow2_" + tag])
        assert O.epe(f32, ref) < 1e-3
        epe, scale = O.epe(f16, ref), ref.abs().mean().item()
        print("fp16 old [%s]: EPE %.3e, mean|flow| %.3f" % (tag, epe, scale))
        assert epe <= 5e-2 * scale, (tag, epe, scale)


def test_forward_fp16_full_size_batch16_repeatable(dev):
    """BASELINE geometry (16 x 6 x 448 x 1024): the half-precision plan is bit-reproducible run to run (fixed reduction
    orders, no atomics; exercises the 16-row tiles, 2-slot rings and two-per-CU variants the small cases do not reach) and
    stays within 1e-2 x mean|flow| of the fp32 plan of the same model."""
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
    assert torch.isfinite(a).all() and epe <= 1e-2 * scale
