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
    (2, 3, 16, 33, 70, 2, 1, True),             # image layer: 3 channels padded to one group
    (1, 96, 196, 14, 32, 2, 1, True),           # Cout not a multiple of 8
    (1, 64, 2, 9, 33, 1, 1, False),             # flow head, no activation
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
        F16.conv3x3_f16(arena[:, 15:], wp, bias.to(dev), cin, cout, dilation=8)       # no fp16 kernel for dilation 8 yet
    with pytest.raises(ValueError):
        F16.conv3x3_f16(arena[:, 15:], wp, bias.to(dev), cin + 8, cout)
