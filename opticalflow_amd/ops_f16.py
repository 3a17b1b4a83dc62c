"""Half-precision building blocks (BASELINE configs 3-4): channel-blocked "c8" activations
``[B, ceil(C/8), H, W, 8]`` (torch.float16) and the kernels over them -- MFMA 3x3 convolution, first pyramid layer from
the float32 image, PWC-Net's cost volume, the warp, layout conversions.  engine_f16.PwcPlanF16 assembles the network.

Device tensors only; every function launches kernels of libpwc_hip.so on the current stream (no CPU path).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import FLAG_ACT_LEAKY, PwcHipError, check


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_device(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise PwcHipError("%s is on %s: the HIP path needs device tensors and has no CPU fallback" % (name, t.device))


def c8_shape(B: int, C: int, H: int, W: int):
    return (B, (C + 7) // 8, H, W, 8)


def _c8_bstride(t: torch.Tensor, name: str, dtype: torch.dtype = torch.float16) -> int:
    """c8 tensors must have dense [Cg,H,W,8] planes; the batch stride is free (channel-group slices of an arena)."""
    if t.dim() != 5 or t.shape[4] != 8 or t.dtype != dtype:
        raise ValueError("%s must be a %s [B,Cg,H,W,8] tensor, got %s %s" % (name, dtype, t.dtype, tuple(t.shape)))
    _require_device(t, name)
    _, cg, h, w, _ = t.shape
    if t.stride()[1:] != (h * w * 8, w * 8, 8, 1):
        raise ValueError("%s must have dense [Cg,H,W,8] planes (strides %s)" % (name, t.stride()))
    return t.stride(0) if t.shape[0] > 1 else cg * h * w * 8


def to_c8(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,C,H,W] float32 (dense C,H,W planes) -> [B,ceil(C/8),H,W,8] float16, zero channel padding."""
    _require_device(x, "x")
    if x.dim() != 4 or x.dtype != torch.float32 or not x[0].is_contiguous():
        raise ValueError("x must be float32 [B,C,H,W] with dense planes")
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty(c8_shape(B, C, H, W), dtype=torch.float16, device=x.device)
    elif tuple(out.shape) != c8_shape(B, C, H, W):
        raise ValueError("out must be %s" % (c8_shape(B, C, H, W),))
    bso = _c8_bstride(out, "out")
    with torch.cuda.device(x.device):
        rc = _lib.load().pwc_nchw_to_c8_f16(x.data_ptr(), out.data_ptr(), B, C, H, W,
                                            x.stride(0) if B > 1 else C * H * W, bso, _stream(x))
    check(rc, "pwc_nchw_to_c8_f16")
    return out


def to_c8_hilo(x: torch.Tensor, out_hi: torch.Tensor, out_lo: torch.Tensor) -> None:
    """[B,C,H,W] float32 -> two c8 half tensors: out_hi = half(x), out_lo = half(x - out_hi) (the rounding residual)."""
    _require_device(x, "x")
    if x.dim() != 4 or x.dtype != torch.float32 or not x[0].is_contiguous():
        raise ValueError("x must be float32 [B,C,H,W] with dense planes")
    B, C, H, W = x.shape
    for t, n in ((out_hi, "out_hi"), (out_lo, "out_lo")):
        if tuple(t.shape) != c8_shape(B, C, H, W):
            raise ValueError("%s must be %s" % (n, c8_shape(B, C, H, W)))
    bh, bl = _c8_bstride(out_hi, "out_hi"), _c8_bstride(out_lo, "out_lo")
    with torch.cuda.device(x.device):
        rc = _lib.load().pwc_nchw_to_c8_f16_hilo(x.data_ptr(), out_hi.data_ptr(), out_lo.data_ptr(), B, C, H, W,
                                                 x.stride(0) if B > 1 else C * H * W, bh, bl, _stream(x))
    check(rc, "pwc_nchw_to_c8_f16_hilo")


def from_c8(x: torch.Tensor, channels: int) -> torch.Tensor:
    """[B,Cg,H,W,8] float16 -> [B,channels,H,W] float32."""
    bsx = _c8_bstride(x, "x")
    B, cg, H, W, _ = x.shape
    if not (cg - 1) * 8 < channels <= cg * 8:
        raise ValueError("channels=%d does not match %d channel groups" % (channels, cg))
    out = torch.empty((B, channels, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().pwc_c8_f16_to_nchw(x.data_ptr(), out.data_ptr(), B, channels, H, W, bsx, channels * H * W, _stream(x))
    check(rc, "pwc_c8_f16_to_nchw")
    return out


def pack_conv3x3_f16(weight: torch.Tensor, split: bool = False) -> torch.Tensor:
    """[Cout,Cin,3,3] float32 device tensor -> packed float16 filter bank for conv3x3_f16.
    split: every 32-row cout tile carries 16 filters rounded to half plus their rounding residuals, i.e. ~22-bit filters --
    pass split_w=True to conv3x3_f16.  For Cout <= 16 (the 2-channel flow heads) the residuals ride in the idle half of the
    tile at no MFMA cost; wider layers (the strict mode's level-2 / context blocks) pay twice the MFMA passes."""
    _require_device(weight, "weight")
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.dtype != torch.float32:
        raise ValueError("expected float32 [Cout,Cin,3,3], got %s %s" % (weight.dtype, tuple(weight.shape)))
    lib = _lib.load()
    cout, cin = weight.shape[:2]
    nbytes = lib.pwc_conv3x3_f16_packed_bytes_split(cin, cout) if split else lib.pwc_conv3x3_f16_packed_bytes(cin, cout)
    wp = torch.empty((nbytes // 2,), dtype=torch.float16, device=weight.device)
    fn = lib.pwc_conv3x3_f16_pack_split if split else lib.pwc_conv3x3_f16_pack
    with torch.cuda.device(weight.device):
        rc = fn(weight.contiguous().data_ptr(), wp.data_ptr(), cin, cout, _stream(weight))
    check(rc, "pwc_conv3x3_f16_pack_split" if split else "pwc_conv3x3_f16_pack")
    return wp


def conv3x3_f16(x: torch.Tensor, wpacked: torch.Tensor, bias: torch.Tensor, cin: int, cout: int, stride: int = 1,
                dilation: int = 1, leaky_slope: Optional[float] = 0.1, out: Optional[torch.Tensor] = None,
                out_f32: bool = False, split_w: bool = False) -> torch.Tensor:
    """3x3 convolution (padding = dilation) + bias (+ LeakyReLU) on c8 float16 activations, fp32 accumulation.
    out_f32: the result stays float32 (c8 layout) instead of being rounded to half; split_w: `wpacked` comes from
    pack_conv3x3_f16(split=True)."""
    lib = _lib.load()
    bsx = _c8_bstride(x, "x")
    B, cg, H, W, _ = x.shape
    if cg != (cin + 7) // 8:
        raise ValueError("x has %d channel groups, Cin=%d needs %d" % (cg, cin, (cin + 7) // 8))
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    odt = torch.float32 if out_f32 else torch.float16
    if out is None:
        out = torch.empty(c8_shape(B, cout, ho, wo), dtype=odt, device=x.device)
    elif tuple(out.shape) != c8_shape(B, cout, ho, wo):
        raise ValueError("out must be %s" % (c8_shape(B, cout, ho, wo),))
    bsy = _c8_bstride(out, "out", odt)
    need = lib.pwc_conv3x3_f16_packed_bytes_split(cin, cout) if split_w else lib.pwc_conv3x3_f16_packed_bytes(cin, cout)
    if wpacked.dtype != torch.float16 or wpacked.numel() * 2 != need or wpacked.device != x.device:
        raise ValueError("packed filters do not match Cin=%d Cout=%d" % (cin, cout))
    if bias.dtype != torch.float32 or bias.numel() != cout or bias.device != x.device or not bias.is_contiguous():
        raise ValueError("bias must be float32[%d] on %s" % (cout, x.device))
    with torch.cuda.device(x.device):
        rc = lib.pwc_conv2d_f16_fwd(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr(), out.data_ptr(), B, cin, H, W, cout,
                                    stride, dilation, (FLAG_ACT_LEAKY if leaky_slope is not None else 0) |
                                    (_lib.FLAG_CONV_OUT_F32 if out_f32 else 0) | (_lib.FLAG_CONV_SPLIT_W if split_w else 0),
                                    float(leaky_slope or 0.0), bsx, bsy, _stream(x))
    check(rc, "pwc_conv2d_f16_fwd")
    return out


def correlation_c8(in1: torch.Tensor, in2: torch.Tensor, channels: int, corr_multiply: float = 1.0, normalize: bool = False,
                   leaky_slope: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PWC-Net cost volume (pad 4, k 1, d 4, strides 1) on c8 tensors -> [B,11,H,W,8] (81 channels + 7 zeros)."""
    lib = _lib.load()
    bs1, bs2 = _c8_bstride(in1, "in1"), _c8_bstride(in2, "in2")
    if in1.shape != in2.shape or in1.shape[1] != (channels + 7) // 8:
        raise ValueError("in1/in2 must both be %d channel groups, got %s / %s" % ((channels + 7) // 8, tuple(in1.shape), tuple(in2.shape)))
    B, _, H, W, _ = in1.shape
    if out is None:
        out = torch.empty((B, 11, H, W, 8), dtype=torch.float16, device=in1.device)
    elif tuple(out.shape) != (B, 11, H, W, 8):
        raise ValueError("out must be %s" % ((B, 11, H, W, 8),))
    bso = _c8_bstride(out, "out")
    flags = (_lib.FLAG_CORR_NORMALIZE if normalize else 0) | (FLAG_ACT_LEAKY if leaky_slope is not None else 0)
    with torch.cuda.device(in1.device):
        rc = lib.pwc_corr81_c8_f16(in1.data_ptr(), in2.data_ptr(), out.data_ptr(), B, channels, H, W, float(corr_multiply),
                                   flags, float(leaky_slope or 0.0), bs1, bs2, bso, _stream(in1))
    check(rc, "pwc_corr81_c8_f16")
    return out


def warp_c8(x: torch.Tensor, flo: torch.Tensor, channels: int, flo_channel: int = 0, flow_scale: float = 1.0,
            align_corners: bool = False, mask_threshold: float = 0.9999, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PWCDCNet.warp on c8 tensors; `flo` is a ONE-group c8 tensor [B,1,H,W,8] with (u,v) at channels flo_channel, +1."""
    lib = _lib.load()
    bsx, bsf = _c8_bstride(x, "x"), _c8_bstride(flo, "flo")
    B, cg, H, W, _ = x.shape
    if cg != (channels + 7) // 8 or tuple(flo.shape) != (B, 1, H, W, 8):
        raise ValueError("x must have %d groups and flo must be %s" % ((channels + 7) // 8, (B, 1, H, W, 8)))
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    elif out.shape != x.shape:
        raise ValueError("out must match x")
    bso = _c8_bstride(out, "out")
    with torch.cuda.device(x.device):
        rc = lib.pwc_warp_c8_f16(x.data_ptr(), flo.data_ptr(), out.data_ptr(), B, channels, H, W, int(flo_channel),
                                 float(flow_scale), 1 if align_corners else 0, float(mask_threshold), bsx, bsf, bso, _stream(x))
    check(rc, "pwc_warp_c8_f16")
    return out


def level_entry(c1: torch.Tensor, c2: torch.Tensor, flow32: torch.Tensor, feat_phases: torch.Tensor,
                deconv_w: torch.Tensor, deconv_b: torch.Tensor, channels: int,
                c1_dst: torch.Tensor, flow_group: torch.Tensor, out: torch.Tensor, flow_scale: float = 1.0,
                align_corners: bool = False, mask_threshold: float = 0.9999) -> torch.Tensor:
    """Entry of a decoder level in one launch (PWCNet.py:208-212): up_flow = deconvL(flow) computed in fp32 from
    `flow32` (float32 c8 [B,1,H/2,W/2,8], flow in channels 0,1) with the ConvTranspose2d parameters `deconv_w`
    [2,2,4,4] / `deconv_b` [2]; up_feat from the 4-phase float32 tensor `feat_phases` [B,1,H/2,W/2,8]; both rounded to
    half into channels 0..3 of `flow_group` [B,1,H,W,8]; c1 copied into `c1_dst`; warp(c2, up_flow * flow_scale)
    (fp32 up_flow) written to `out`."""
    lib = _lib.load()
    B, cg, H, W, _ = c2.shape
    if cg != (channels + 7) // 8 or c1.shape != c2.shape or c1_dst.shape != c2.shape or out.shape != c2.shape:
        raise ValueError("c1, c2, c1_dst, out must all be [B,%d,H,W,8]" % ((channels + 7) // 8))
    if H % 2 or W % 2 or tuple(flow32.shape) != (B, 1, H // 2, W // 2, 8) or feat_phases.shape != flow32.shape \
            or tuple(flow_group.shape) != (B, 1, H, W, 8):
        raise ValueError("flow32 / feat_phases must be [B,1,H/2,W/2,8] and flow_group [B,1,H,W,8]")
    for t, n, shp in ((deconv_w, "deconv_w", (2, 2, 4, 4)), (deconv_b, "deconv_b", (2,))):
        if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != c2.device:
            raise ValueError("%s must be contiguous float32 %s on %s" % (n, shp, c2.device))
    strides = [_c8_bstride(t, n, dt) for t, n, dt in (
        (c1, "c1", torch.float16), (c2, "c2", torch.float16), (flow32, "flow32", torch.float32),
        (feat_phases, "feat_phases", torch.float32), (c1_dst, "c1_dst", torch.float16),
        (flow_group, "flow_group", torch.float16), (out, "out", torch.float16))]
    with torch.cuda.device(c2.device):
        rc = lib.pwc_level_entry_c8_f16(c1.data_ptr(), c2.data_ptr(), flow32.data_ptr(), feat_phases.data_ptr(),
                                        deconv_w.data_ptr(), deconv_b.data_ptr(),
                                        c1_dst.data_ptr(), flow_group.data_ptr(), out.data_ptr(), B, channels, H, W,
                                        float(flow_scale), 1 if align_corners else 0, float(mask_threshold),
                                        *strides, _stream(c2))
    check(rc, "pwc_level_entry_c8_f16")
    return out


def level_entry_correlation(c1: torch.Tensor, c2: torch.Tensor, flow32: torch.Tensor, feat_phases: torch.Tensor,
                            deconv_w: torch.Tensor, deconv_b: torch.Tensor, channels: int,
                            c1_dst: torch.Tensor, flow_group: torch.Tensor, out: torch.Tensor, flow_scale: float = 1.0,
                            align_corners: bool = False, mask_threshold: float = 0.9999, corr_multiply: float = 1.0,
                            normalize: bool = False, leaky_slope: Optional[float] = None) -> torch.Tensor:
    """level_entry() followed by correlation_c8(c1, warped) as ONE kernel (PWCNet.py:208-214): the warped features stay in LDS,
    `out` is the cost volume [B,11,H,W,8]; flow_group and c1_dst are written as by level_entry().  Bit-identical to the two calls."""
    lib = _lib.load()
    B, cg, H, W, _ = c2.shape
    if cg != (channels + 7) // 8 or c1.shape != c2.shape or c1_dst.shape != c2.shape or tuple(out.shape) != (B, 11, H, W, 8):
        raise ValueError("c1, c2, c1_dst must be [B,%d,H,W,8] and out [B,11,H,W,8]" % ((channels + 7) // 8))
    if H % 2 or W % 2 or tuple(flow32.shape) != (B, 1, H // 2, W // 2, 8) or feat_phases.shape != flow32.shape \
            or tuple(flow_group.shape) != (B, 1, H, W, 8):
        raise ValueError("flow32 / feat_phases must be [B,1,H/2,W/2,8] and flow_group [B,1,H,W,8]")
    for t, n, shp in ((deconv_w, "deconv_w", (2, 2, 4, 4)), (deconv_b, "deconv_b", (2,))):
        if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != c2.device:
            raise ValueError("%s must be contiguous float32 %s on %s" % (n, shp, c2.device))
    strides = [_c8_bstride(t, n, dt) for t, n, dt in (
        (c1, "c1", torch.float16), (c2, "c2", torch.float16), (flow32, "flow32", torch.float32),
        (feat_phases, "feat_phases", torch.float32), (c1_dst, "c1_dst", torch.float16),
        (flow_group, "flow_group", torch.float16), (out, "out", torch.float16))]
    flags = (_lib.FLAG_CORR_NORMALIZE if normalize else 0) | (FLAG_ACT_LEAKY if leaky_slope is not None else 0)
    with torch.cuda.device(c2.device):
        rc = lib.pwc_level_corr81_c8_f16(c1.data_ptr(), c2.data_ptr(), flow32.data_ptr(), feat_phases.data_ptr(),
                                         deconv_w.data_ptr(), deconv_b.data_ptr(),
                                         c1_dst.data_ptr(), flow_group.data_ptr(), out.data_ptr(), B, channels, H, W,
                                         float(flow_scale), 1 if align_corners else 0, float(mask_threshold),
                                         float(corr_multiply), flags, float(leaky_slope or 0.0), *strides, _stream(c2))
    check(rc, "pwc_level_corr81_c8_f16")
    return out


def image_conv_s2(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, leaky_slope: float = 0.1,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv1a (Conv2d(3,16,3,stride 2,pad 1) + LeakyReLU) from a float32 [B,3,H,W] image (dense planes, free batch
    stride) to c8 halves [B,2,H/2,W/2,8]; weight [16,3,3,3] / bias [16] float32."""
    _require_device(x, "x")
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x[0].is_contiguous():
        raise ValueError("x must be float32 [B,3,H,W] with dense planes")
    if tuple(weight.shape) != (16, 3, 3, 3) or weight.dtype != torch.float32 or not weight.is_contiguous() or bias.numel() != 16:
        raise ValueError("weight must be contiguous float32 [16,3,3,3] and bias [16]")
    B, _, H, W = x.shape
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, 2, ho, wo, 8), dtype=torch.float16, device=x.device)
    elif tuple(out.shape) != (B, 2, ho, wo, 8):
        raise ValueError("out must be %s" % ((B, 2, ho, wo, 8),))
    bso = _c8_bstride(out, "out")
    with torch.cuda.device(x.device):
        rc = _lib.load().pwc_image_conv_s2_c8_f16(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                                                  float(leaky_slope), x.stride(0) if B > 1 else 3 * H * W, bso, _stream(x))
    check(rc, "pwc_image_conv_s2_c8_f16")
    return out


def pack_pyramid1(w1a: torch.Tensor, b1a: torch.Tensor, w1aa: torch.Tensor, b1aa: torch.Tensor, w1b: torch.Tensor,
                  b1b: torch.Tensor, w2a: torch.Tensor, b2a: torch.Tensor):
    """Filters / biases of conv1a, conv1aa, conv1b, conv2a (nn.Conv2d layouts, float32, device) -> (packed halves, biases
    float32[80]) for pyramid1_fused: per layer the [tap*2 + kh][cout][8 channels] image the kernel's 16x16x32 MFMA steps read
    (20 rows, the last two zero); conv1a as [k/8][cout][k%8] with k = ci*9 + ky*3 + kx padded from 27 to 32."""
    for t, shp in ((w1a, (16, 3, 3, 3)), (w1aa, (16, 16, 3, 3)), (w1b, (16, 16, 3, 3)), (w2a, (32, 16, 3, 3))):
        _require_device(t, "weight")
        if tuple(t.shape) != shp or t.dtype != torch.float32:
            raise ValueError("expected float32 %s, got %s %s" % (shp, t.dtype, tuple(t.shape)))
    dev = w1a.device
    k1 = torch.zeros((32, 16), device=dev)
    k1[:27] = w1a.reshape(16, 27).t()
    parts = [k1.view(4, 8, 16).permute(0, 2, 1).reshape(-1)]
    for w in (w1aa, w1b, w2a):
        co = w.shape[0]
        img = torch.zeros((20, co, 8), device=dev)
        img[:18] = w.reshape(co, 2, 8, 9).permute(3, 1, 0, 2).reshape(18, co, 8)       # (tap, kh, cout, j)
        parts.append(img.reshape(-1))
    packed = torch.cat(parts).to(torch.float16).contiguous()
    need = _lib.load().pwc_pyramid1_f16_packed_bytes()
    assert packed.numel() * 2 == need, (packed.numel() * 2, need)
    bias = torch.cat([b.reshape(-1).float() for b in (b1a, b1aa, b1b, b2a)]).contiguous()
    return packed, bias


def pyramid1_fused(x: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, leaky_slope: float = 0.1,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv2a(conv1b(conv1aa(conv1a(x)))) (PWCNet.py:52-55,184-187) in one launch: x float32 [B,3,H,W] (dense planes, free
    batch stride) -> c8 halves [B,4,H/4,W/4,8]; `packed`, `bias` from pack_pyramid1.  The level-1 maps stay in LDS."""
    _require_device(x, "x")
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not x[0].is_contiguous():
        raise ValueError("x must be float32 [B,3,H,W] with dense planes")
    lib = _lib.load()
    if packed.dtype != torch.float16 or packed.numel() * 2 != lib.pwc_pyramid1_f16_packed_bytes() or bias.numel() != 80 \
            or bias.dtype != torch.float32 or packed.device != x.device or bias.device != x.device:
        raise ValueError("packed / bias do not come from pack_pyramid1 on %s" % x.device)
    B, _, H, W = x.shape
    h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    h2, w2 = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
    if out is None:
        out = torch.empty((B, 4, h2, w2, 8), dtype=torch.float16, device=x.device)
    elif tuple(out.shape) != (B, 4, h2, w2, 8):
        raise ValueError("out must be %s" % ((B, 4, h2, w2, 8),))
    bso = _c8_bstride(out, "out")
    with torch.cuda.device(x.device):
        rc = lib.pwc_pyramid1_fused_f16(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                                        float(leaky_slope), x.stride(0) if B > 1 else 3 * H * W, bso, _stream(x))
    check(rc, "pwc_pyramid1_fused_f16")
    return out
