"""Tensor-level wrappers over the C ABI (device pointers + current HIP stream).

PyTorch is used only for device memory and streams; all arithmetic happens in
libpwc_hip.so.  Every wrapper validates what the kernels assume (device,
dtype, dense C/H/W planes, batch stride) *before* launching -- the reference
does no validation and silently assumes contiguous NCHW
(correlation_cuda_kernel.cu:341-360).
"""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (FLAG_ACT_LEAKY, FLAG_CONV_RESIDUAL, FLAG_CORR_NORMALIZE, PWC_F16, PWC_F32, PwcHipError, check)

_DTYPES = {torch.float32: PWC_F32, torch.float16: PWC_F16}


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError("unsupported dtype %s (float32 / float16 only)" % t.dtype) from None


def _plane_dense(t: torch.Tensor, name: str) -> int:
    """Require [B,C,H,W] with dense C,H,W planes; return the batch stride in elements."""
    if t.dim() != 4:
        raise ValueError("%s must be 4-D [B,C,H,W], got %s" % (name, tuple(t.shape)))
    if not t.is_cuda:
        raise PwcHipError("%s is on %s: the HIP path needs device tensors and has no CPU fallback" % (name, t.device))
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    ok = (W == 1 or sw == 1) and (H == 1 or sh == W) and (C == 1 or sc == H * W)
    if not ok:
        raise ValueError("%s must have dense C,H,W planes (strides %s for shape %s)" % (name, t.stride(), tuple(t.shape)))
    if B == 1:
        return C * H * W
    if sb < C * H * W:
        raise ValueError("%s batch stride %d smaller than C*H*W" % (name, sb))
    return sb


def densify(t: torch.Tensor) -> torch.Tensor:
    """Return t if its C,H,W planes are dense (batch stride free), else a contiguous copy."""
    if t.dim() == 4:
        B, C, H, W = t.shape
        sb, sc, sh, sw = t.stride()
        if (W == 1 or sw == 1) and (H == 1 or sh == W) and (C == 1 or sc == H * W) and (B == 1 or sb >= C * H * W):
            return t
    return t.contiguous()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def corr_output_shape(C: int, H: int, W: int, pad_size: int, kernel_size: int, max_displacement: int,
                      stride1: int, stride2: int) -> Tuple[int, int, int]:
    """Shape contract of the reference binding (correlation_cuda.cc:25-38)."""
    krad = (kernel_size - 1) // 2
    border = krad + max_displacement
    drad = max_displacement // stride2
    return ((2 * drad + 1) ** 2,
            int(math.ceil((H + 2 * pad_size - 2 * border) / float(stride1))),
            int(math.ceil((W + 2 * pad_size - 2 * border) / float(stride1))))


def correlation(in1: torch.Tensor, in2: torch.Tensor, pad_size: int = 4, kernel_size: int = 1,
                max_displacement: int = 4, stride1: int = 1, stride2: int = 1, corr_multiply: float = 1.0,
                normalize: bool = False, leaky_slope: Optional[float] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    if in1.shape != in2.shape or in1.dtype != in2.dtype or in1.device != in2.device:
        raise ValueError("correlation inputs differ: %s/%s vs %s/%s" % (tuple(in1.shape), in1.dtype, tuple(in2.shape), in2.dtype))
    bs1 = _plane_dense(in1, "input1")
    bs2 = _plane_dense(in2, "input2")
    B, C, H, W = in1.shape
    nch, oh, ow = corr_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    if oh <= 0 or ow <= 0:
        raise ValueError("correlation output would be empty (%d x %d)" % (oh, ow))
    if out is None:
        out = torch.empty((B, nch, oh, ow), dtype=in1.dtype, device=in1.device)
    elif tuple(out.shape) != (B, nch, oh, ow) or out.dtype != in1.dtype or out.device != in1.device:
        raise ValueError("out must be %s %s on %s" % ((B, nch, oh, ow), in1.dtype, in1.device))
    bso = _plane_dense(out, "out")
    flags = (FLAG_CORR_NORMALIZE if normalize else 0) | (FLAG_ACT_LEAKY if leaky_slope is not None else 0)
    with torch.cuda.device(in1.device):
        rc = lib.pwc_corr_fwd(in1.data_ptr(), in2.data_ptr(), out.data_ptr(), B, C, H, W,
                              pad_size, kernel_size, max_displacement, stride1, stride2,
                              float(corr_multiply), _dtype_code(in1), flags, float(leaky_slope or 0.0),
                              bs1, bs2, bso, _stream(in1))
    check(rc, "pwc_corr_fwd")
    return out


def correlation_backward(in1: torch.Tensor, in2: torch.Tensor, grad_out: torch.Tensor, pad_size: int = 4,
                         kernel_size: int = 1, max_displacement: int = 4, stride1: int = 1, stride2: int = 1,
                         corr_multiply: float = 1.0, normalize: bool = False):
    lib = _lib.load()
    for name, t in (("input1", in1), ("input2", in2), ("grad_output", grad_out)):
        _plane_dense(t, name)
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous for the backward kernel" % name)
    B, C, H, W = in1.shape
    g1 = torch.empty_like(in1)
    g2 = torch.empty_like(in2)
    with torch.cuda.device(in1.device):
        rc = lib.pwc_corr_bwd(in1.data_ptr(), in2.data_ptr(), grad_out.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                              B, C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                              float(corr_multiply), _dtype_code(in1), FLAG_CORR_NORMALIZE if normalize else 0,
                              _stream(in1))
    check(rc, "pwc_corr_bwd")
    return g1, g2


def warp_correlation_preferred(B: int, C: int, H: int, W: int) -> bool:
    """Whether the fused warp + correlation kernel beats warp() followed by correlation() for this geometry (rule in the library:
    not for maps of a few tiles, where the small-map correlation kernel wins)."""
    return bool(_lib.load().pwc_warp_corr81_preferred(B, C, H, W))


def warp_correlation(in1: torch.Tensor, x2: torch.Tensor, flo: torch.Tensor, flow_scale: float = 1.0,
                     align_corners: bool = False, mask_threshold: float = 0.9999, corr_multiply: float = 1.0,
                     normalize: bool = False, leaky_slope: Optional[float] = None,
                     out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """correlation(in1, warp(x2, flow_scale * flo)) for PWC-Net's configuration (pad 4, k 1, d 4, strides 1) in ONE kernel:
    the warped tensor is produced tile by tile in LDS and never written to HBM (PWCNet.py:212-213 etc.).  Bit-identical to
    warp() followed by correlation().  Returns None (nothing launched) when the geometry is outside the fused kernel
    (W % 4 != 0 or unaligned operands): call the two operators then."""
    lib = _lib.load()
    bs1, bs2, bsf = _plane_dense(in1, "in1"), _plane_dense(x2, "x2"), _plane_dense(flo, "flo")
    B, C, H, W = in1.shape
    if in1.dtype != torch.float32 or x2.shape != in1.shape or x2.dtype != in1.dtype or tuple(flo.shape) != (B, 2, H, W) \
            or flo.dtype != in1.dtype or x2.device != in1.device or flo.device != in1.device:
        raise ValueError("in1, x2 must be float32 [B,C,H,W] and flo [B,2,H,W] on one device")
    if out is None:
        out = torch.empty((B, 81, H, W), dtype=in1.dtype, device=in1.device)
    elif tuple(out.shape) != (B, 81, H, W) or out.dtype != in1.dtype or out.device != in1.device:
        raise ValueError("out must be %s" % ((B, 81, H, W),))
    bso = _plane_dense(out, "out")
    flags = (FLAG_CORR_NORMALIZE if normalize else 0) | (FLAG_ACT_LEAKY if leaky_slope is not None else 0)
    with torch.cuda.device(in1.device):
        rc = lib.pwc_warp_corr81_fwd(in1.data_ptr(), x2.data_ptr(), flo.data_ptr(), out.data_ptr(), B, C, H, W,
                                     float(flow_scale), 1 if align_corners else 0, float(mask_threshold),
                                     float(corr_multiply), flags, float(leaky_slope or 0.0), bs1, bs2, bsf, bso, _stream(in1))
    if rc == -2:                               # PWC_EUNSUPPORTED: geometry outside the fused kernel
        return None
    check(rc, "pwc_warp_corr81_fwd")
    return out


def warp(x: torch.Tensor, flo: torch.Tensor, flow_scale: float = 1.0, align_corners: bool = False,
         mask_threshold: float = 0.9999, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    bsf = _plane_dense(flo, "flo")
    B, C, H, W = x.shape
    if tuple(flo.shape) != (B, 2, H, W) or flo.dtype != x.dtype or flo.device != x.device:
        raise ValueError("flo must be %s %s, got %s %s" % ((B, 2, H, W), x.dtype, tuple(flo.shape), flo.dtype))
    if out is None:
        out = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (B, C, H, W) or out.dtype != x.dtype or out.device != x.device:
        raise ValueError("out must match x")
    bso = _plane_dense(out, "out")
    with torch.cuda.device(x.device):
        rc = lib.pwc_warp_fwd(x.data_ptr(), flo.data_ptr(), out.data_ptr(), B, C, H, W,
                              float(flow_scale), 1 if align_corners else 0, float(mask_threshold), _dtype_code(x),
                              bsx, bsf, bso, _stream(x))
    check(rc, "pwc_warp_fwd")
    return out


def warp_backward(x: torch.Tensor, flo: torch.Tensor, grad_out: torch.Tensor, flow_scale: float = 1.0,
                  align_corners: bool = False, mask_threshold: float = 0.9999, deterministic: bool = True):
    """(grad_x, grad_flo) of `warp` for contiguous float32 tensors.  deterministic (default): the scatter into grad_x
    accumulates 64-bit fixed-point integers in a scratch buffer (bit-reproducible); False: float atomics."""
    lib = _lib.load()
    for name, t in (("x", x), ("flo", flo), ("grad_out", grad_out)):
        _plane_dense(t, name)
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("%s must be contiguous float32 for the backward kernel" % name)
    B, C, H, W = x.shape
    if tuple(flo.shape) != (B, 2, H, W) or tuple(grad_out.shape) != (B, C, H, W):
        raise ValueError("flo must be %s and grad_out %s" % ((B, 2, H, W), (B, C, H, W)))
    gx = torch.empty_like(x)
    gf = torch.empty_like(flo)
    ws, ws_bytes = None, 0
    if deterministic:
        ws_bytes = lib.pwc_warp_bwd_workspace_bytes(B, C, H, W)
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.pwc_warp_bwd(x.data_ptr(), flo.data_ptr(), grad_out.data_ptr(), gx.data_ptr(), gf.data_ptr(),
                              B, C, H, W, float(flow_scale), 1 if align_corners else 0, float(mask_threshold),
                              _dtype_code(x), ws.data_ptr() if ws is not None else None, ws_bytes, _stream(x))
    check(rc, "pwc_warp_bwd")
    return gx, gf


class WarpFunction(torch.autograd.Function):
    """autograd wrapper of the fused warp: forward and backward both run HIP kernels (what autograd builds from
    PWCNet.py:141-177 in the reference's training scripts)."""

    @staticmethod
    def forward(ctx, x, flo, flow_scale=1.0, align_corners=False, mask_threshold=0.9999):
        x, flo = x.contiguous(), flo.contiguous()
        ctx.save_for_backward(x, flo)
        ctx.cfg = (flow_scale, align_corners, mask_threshold)
        return warp(x, flo, flow_scale, align_corners, mask_threshold)

    @staticmethod
    def backward(ctx, grad_out):
        x, flo = ctx.saved_tensors
        gx, gf = warp_backward(x, flo, grad_out.contiguous(), *ctx.cfg)
        return gx, gf, None, None, None


def pack_conv3x3(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] nn.Conv2d filter bank -> kernel-native packed buffer (device, float32)."""
    lib = _lib.load()
    if weight.dim() != 4 or weight.shape[2:] != (3, 3):
        raise ValueError("expected [Cout,Cin,3,3], got %s" % (tuple(weight.shape),))
    if not weight.is_cuda:
        raise PwcHipError("weights must be on the device")
    w = weight.detach().to(torch.float32).contiguous()
    cout, cin = w.shape[:2]
    nbytes = lib.pwc_conv3x3_packed_bytes(cin, cout, PWC_F32)
    if nbytes <= 0:
        raise PwcHipError("pwc_conv3x3_packed_bytes(%d,%d) = %d" % (cin, cout, nbytes))
    wp = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = lib.pwc_conv3x3_pack(w.data_ptr(), wp.data_ptr(), cin, cout, PWC_F32, _stream(w))
    check(rc, "pwc_conv3x3_pack")
    return wp


def conv3x3(x: torch.Tensor, wpacked: torch.Tensor, bias: torch.Tensor, cout: int, stride: int = 1,
            dilation: int = 1, leaky_slope: Optional[float] = 0.1, residual: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`workspace`: optional device scratch (any dtype, see conv3x3_workspace_bytes) enabling the split-K
    route for layers with few output tiles; without it the layer runs unsplit."""
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    B, cin, H, W = x.shape
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((B, cout, ho, wo), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (B, cout, ho, wo) or out.dtype != x.dtype or out.device != x.device:
        raise ValueError("out must be %s, got %s" % ((B, cout, ho, wo), tuple(out.shape)))
    bsy = _plane_dense(out, "out")
    need = lib.pwc_conv3x3_packed_bytes(cin, cout, PWC_F32)
    if wpacked.dtype != torch.float32 or wpacked.numel() * 4 != need or wpacked.device != x.device:
        raise ValueError("packed weights do not match Cin=%d Cout=%d (have %d B, need %d B)" % (cin, cout, wpacked.numel() * 4, need))
    if bias.dtype != torch.float32 or bias.numel() != cout or bias.device != x.device or not bias.is_contiguous():
        raise ValueError("bias must be float32[%d] on %s" % (cout, x.device))
    flags = FLAG_ACT_LEAKY if leaky_slope is not None else 0
    res_ptr, bsr = 0, 0
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape) or residual.dtype != x.dtype:
            raise ValueError("residual must match the output")
        bsr = _plane_dense(residual, "residual")
        res_ptr = residual.data_ptr()
        flags |= FLAG_CONV_RESIDUAL
    ws_ptr, ws_bytes = 0, 0
    if workspace is not None:
        if workspace.device != x.device or not workspace.is_contiguous():
            raise ValueError("workspace must be a contiguous tensor on %s" % x.device)
        ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    with torch.cuda.device(x.device):
        rc = lib.pwc_conv2d_fwd(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr(), res_ptr, out.data_ptr(),
                                B, cin, H, W, cout, stride, dilation, _dtype_code(x), flags,
                                float(leaky_slope or 0.0), bsx, bsy, bsr, ws_ptr, ws_bytes, _stream(x))
    check(rc, "pwc_conv2d_fwd")
    return out


def conv3x3_wino_preferred(B: int, cin: int, H: int, W: int, cout: int, dilation: int = 1) -> bool:
    """Whether the Winograd route is expected to beat conv3x3 for this stride-1 layer (rule lives in the library)."""
    return bool(_lib.load().pwc_conv3x3_wino_preferred(B, cin, H, W, cout, dilation))


def pack_conv3x3_wino(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] filter bank -> Winograd F(2x2,3x3) filters G g Gt in the kernel's LDS order (device, float32)."""
    lib = _lib.load()
    if weight.dim() != 4 or weight.shape[2:] != (3, 3):
        raise ValueError("expected [Cout,Cin,3,3], got %s" % (tuple(weight.shape),))
    if not weight.is_cuda:
        raise PwcHipError("weights must be on the device")
    w = weight.detach().to(torch.float32).contiguous()
    cout, cin = w.shape[:2]
    up = torch.empty(lib.pwc_conv3x3_wino_packed_bytes(cin, cout) // 4, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = lib.pwc_conv3x3_wino_pack(w.data_ptr(), up.data_ptr(), cin, cout, _stream(w))
    check(rc, "pwc_conv3x3_wino_pack")
    return up


def conv3x3_wino_workspace_bytes(B: int, cin: int, H: int, W: int, cout: int, dilation: int = 1) -> int:
    """Scratch bytes the split-K form of this layer wants (0 = it does not split)."""
    n = _lib.load().pwc_conv3x3_wino_workspace_bytes(B, cin, H, W, cout, dilation)
    if n < 0:
        raise ValueError("bad conv geometry")
    return int(n)


def conv3x3_wino(x: torch.Tensor, upacked: torch.Tensor, bias: torch.Tensor, cout: int, leaky_slope: Optional[float] = 0.1,
                 out: Optional[torch.Tensor] = None, dilation: int = 1, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 / stride 1 convolution (padding = dilation) + bias (+ LeakyReLU) by Winograd F(2x2,3x3) on the matrix cores (fp32)."""
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    B, cin, H, W = x.shape
    if x.dtype != torch.float32:
        raise ValueError("conv3x3_wino is fp32 only")
    if out is None:
        out = torch.empty((B, cout, H, W), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (B, cout, H, W) or out.dtype != x.dtype or out.device != x.device:
        raise ValueError("out must be %s, got %s" % ((B, cout, H, W), tuple(out.shape)))
    bsy = _plane_dense(out, "out")
    need = lib.pwc_conv3x3_wino_packed_bytes(cin, cout)
    if upacked.dtype != torch.float32 or upacked.numel() * 4 != need or upacked.device != x.device:
        raise ValueError("packed Winograd filters do not match Cin=%d Cout=%d (have %d B, need %d B)" % (cin, cout, upacked.numel() * 4, need))
    if bias.dtype != torch.float32 or bias.numel() != cout or bias.device != x.device or not bias.is_contiguous():
        raise ValueError("bias must be float32[%d] on %s" % (cout, x.device))
    ws_ptr, ws_bytes = 0, 0
    if workspace is not None:
        if workspace.device != x.device or not workspace.is_contiguous():
            raise ValueError("workspace must be a contiguous tensor on %s" % x.device)
        ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    with torch.cuda.device(x.device):
        rc = lib.pwc_conv3x3_wino_fwd(x.data_ptr(), upacked.data_ptr(), bias.data_ptr(), out.data_ptr(), B, cin, H, W, cout, dilation,
                                      FLAG_ACT_LEAKY if leaky_slope is not None else 0, float(leaky_slope or 0.0), bsx, bsy,
                                      ws_ptr, ws_bytes, _stream(x))
    check(rc, "pwc_conv3x3_wino_fwd")
    return out


def conv3x3_wino4_preferred(B: int, cin: int, H: int, W: int, cout: int, dilation: int = 1) -> bool:
    """Measured rule: does Winograd F(4x4,3x3) beat F(2x2,3x3) for this layer (large, well-filled maps, dilation 1)?"""
    return bool(_lib.load().pwc_conv3x3_wino4_preferred(B, cin, H, W, cout, dilation))


def pack_conv3x3_wino4(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] float32 device tensor -> G g Gt of F(4x4,3x3) in the kernel's LDS order (4x the filter bytes)."""
    if not weight.is_cuda:
        raise PwcHipError("weight is on %s: the HIP path needs device tensors" % weight.device)
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.dtype != torch.float32:
        raise ValueError("expected float32 [Cout,Cin,3,3], got %s %s" % (weight.dtype, tuple(weight.shape)))
    lib = _lib.load()
    cout, cin = weight.shape[:2]
    up = torch.empty((lib.pwc_conv3x3_wino4_packed_bytes(cin, cout) // 4,), dtype=torch.float32, device=weight.device)
    w = weight.contiguous()
    with torch.cuda.device(w.device):
        rc = lib.pwc_conv3x3_wino4_pack(w.data_ptr(), up.data_ptr(), cin, cout, _stream(w))
    check(rc, "pwc_conv3x3_wino4_pack")
    return up


def _workspace_args(workspace: Optional[torch.Tensor], x: torch.Tensor) -> Tuple[int, int]:
    if workspace is None:
        return 0, 0
    if workspace.device != x.device or not workspace.is_contiguous():
        raise ValueError("workspace must be a contiguous tensor on %s" % x.device)
    return workspace.data_ptr(), workspace.numel() * workspace.element_size()


def conv3x3_wino4_workspace_bytes(B: int, cin: int, H: int, W: int, cout: int) -> int:
    """Scratch bytes the tail split of this layer's F(4x4) launches wants (0 = no partial last round worth splitting)."""
    n = _lib.load().pwc_conv3x3_wino4_workspace_bytes(B, cin, H, W, cout)
    if n < 0:
        raise ValueError("bad conv geometry")
    return int(n)


def conv3x3_wino4(x: torch.Tensor, upacked: torch.Tensor, bias: torch.Tensor, cout: int, leaky_slope: Optional[float] = 0.1,
                  out: Optional[torch.Tensor] = None, split2: bool = False, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 / stride 1 / padding 1 convolution + bias (+ LeakyReLU) by Winograd F(4x4,3x3) on the matrix cores (fp32; W % 4 == 0).
    split2: the result is stored as its four pixel lattices, [4B, cout, H/2, W/2] with image 4b + 2(y & 1) + (x & 1) -- the input
    layout in which the next, twice-as-dilated layer is a dilation-1 convolution (lattice_unsplit is the inverse).
    workspace (conv3x3_wino4_workspace_bytes): lets a launch whose last round of workgroups would leave most CUs idle run that
    round's tiles as input-channel slices (same result up to the fp32 summation order of those tiles; deterministic)."""
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    B, cin, H, W = x.shape
    if x.dtype != torch.float32:
        raise ValueError("conv3x3_wino4 is fp32 only")
    oshape = (4 * B, cout, H // 2, W // 2) if split2 else (B, cout, H, W)
    if split2 and (H % 2 or W % 8):
        raise ValueError("split2 needs even H and W % 8 == 0")
    if out is None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != oshape or out.dtype != x.dtype or out.device != x.device:
        raise ValueError("out must be %s, got %s" % (oshape, tuple(out.shape)))
    bsy = _plane_dense(out, "out")
    need = lib.pwc_conv3x3_wino4_packed_bytes(cin, cout)
    if upacked.dtype != torch.float32 or upacked.numel() * 4 != need or upacked.device != x.device:
        raise ValueError("packed F(4x4,3x3) filters do not match Cin=%d Cout=%d (have %d B, need %d B)" % (cin, cout, upacked.numel() * 4, need))
    if bias.dtype != torch.float32 or bias.numel() != cout or bias.device != x.device or not bias.is_contiguous():
        raise ValueError("bias must be float32[%d] on %s" % (cout, x.device))
    ws_ptr, ws_bytes = _workspace_args(workspace, x)
    with torch.cuda.device(x.device):
        rc = lib.pwc_conv3x3_wino4_fwd(x.data_ptr(), upacked.data_ptr(), bias.data_ptr(), out.data_ptr(), B, cin, H, W, cout, 1,
                                       (FLAG_ACT_LEAKY if leaky_slope is not None else 0) | (_lib.FLAG_CONV_SPLIT2 if split2 else 0),
                                       float(leaky_slope or 0.0), bsx, bsy, ws_ptr, ws_bytes, _stream(x))
    check(rc, "pwc_conv3x3_wino4_fwd")
    return out


def kitti_ingest(pairs_u8: torch.Tensor, mean, std, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 RGB pairs [n,2,H,W,3] on the device -> float32 [n,6,Hp,Wp] (Hp, Wp = H, W rounded up to multiples of 64): ToTensor +
    (v - mean) / std per channel, the two images concatenated along the channels, replicate padding (inference_kitti.py:53-63,175-178,
    208-210) as one kernel (C-ABI pwc_kitti_ingest_u8)."""
    import ctypes
    if not pairs_u8.is_cuda or pairs_u8.dtype != torch.uint8 or pairs_u8.dim() != 5 or pairs_u8.shape[1] != 2 or pairs_u8.shape[4] != 3 \
            or not pairs_u8.is_contiguous():
        raise ValueError("pairs_u8 must be a contiguous uint8 device tensor [n,2,H,W,3]")
    n, _, H, W, _ = pairs_u8.shape
    Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
    if out is None:
        out = torch.empty((n, 6, Hp, Wp), dtype=torch.float32, device=pairs_u8.device)
    elif tuple(out.shape) != (n, 6, Hp, Wp) or out.dtype != torch.float32 or out.device != pairs_u8.device:
        raise ValueError("out must be float32 %s" % ((n, 6, Hp, Wp),))
    bso = _plane_dense(out, "out")
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(pairs_u8.device):
        rc = _lib.load().pwc_kitti_ingest_u8(pairs_u8.data_ptr(), out.data_ptr(), n, H, W, m3, s3, bso, _stream(pairs_u8))
    check(rc, "pwc_kitti_ingest_u8")
    return out


def flow_upsample(flow_q: torch.Tensor, crop_h: int, crop_w: int, out_h: int, out_w: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[n,2,Hq,Wq] -> [n,2,out_h,out_w]: crop to the top-left crop_h x crop_w, bilinear resize (align_corners=True), u * out_w / crop_w,
    v * out_h / crop_h -- `unpad` + `flow_resize` of inference_kitti.py:66-91 as one kernel (C-ABI pwc_flow_upsample_f32)."""
    if not flow_q.is_cuda or flow_q.dtype != torch.float32 or flow_q.dim() != 4 or flow_q.shape[1] != 2:
        raise ValueError("flow_q must be a float32 device tensor [n,2,Hq,Wq]")
    n, _, Hq, Wq = flow_q.shape
    bsq = _plane_dense(flow_q, "flow_q")
    if out is None:
        out = torch.empty((n, 2, out_h, out_w), dtype=torch.float32, device=flow_q.device)
    elif tuple(out.shape) != (n, 2, out_h, out_w) or out.dtype != torch.float32 or out.device != flow_q.device or not out.is_contiguous():
        raise ValueError("out must be contiguous float32 %s" % ((n, 2, out_h, out_w),))
    with torch.cuda.device(flow_q.device):
        rc = _lib.load().pwc_flow_upsample_f32(flow_q.data_ptr(), out.data_ptr(), n, Hq, Wq, crop_h, crop_w, out_h, out_w, bsq, _stream(flow_q))
    check(rc, "pwc_flow_upsample_f32")
    return out


def lattice_unsplit(x: torch.Tensor, batch: int, levels: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inverse of `levels` nested split2 stores: [batch * 4**levels, C, h, w] (contiguous) -> [batch, C, h << levels, w << levels]."""
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4 or x.shape[0] != batch * 4 ** levels:
        raise ValueError("x must be a contiguous float32 device tensor [batch * 4**levels, C, h, w]")
    _, C, h, w = x.shape
    oshape = (batch, C, h << levels, w << levels)
    if out is None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != oshape or out.dtype != x.dtype or out.device != x.device:
        raise ValueError("out must be %s, got %s" % (oshape, tuple(out.shape)))
    bsy = _plane_dense(out, "out")
    with torch.cuda.device(x.device):
        rc = _lib.load().pwc_lattice_unsplit_f32(x.data_ptr(), out.data_ptr(), batch, C, h, w, levels, bsy, _stream(x))
    check(rc, "pwc_lattice_unsplit_f32")
    return out


def conv3x3_workspace_bytes(B: int, cin: int, H: int, W: int, cout: int, stride: int = 1, dilation: int = 1) -> int:
    """Scratch bytes the split-K route of this layer wants (0 = it never splits)."""
    n = _lib.load().pwc_conv2d_workspace_bytes(B, cin, H, W, cout, stride, dilation)
    if n < 0:
        raise ValueError("bad conv geometry")
    return int(n)


def head_upfeat_supported(B: int, H: int, W: int, min_tiles: int = 64) -> bool:
    """Geometry gate of pwc_head_upfeat_fwd (mirrors stream3x3_ok in csrc/pwc_stream3x3.hip).  min_tiles < 64: the gate of
    pwc_head_upfeat_ws_fwd with a workspace (Cin slices: stream3x3_head_upfeat_sliced_ok, at least 4 tiles)."""
    return (W % 4 == 0 and W >= int(os.environ.get("PWC_STREAM_MINW", "64"))
            and B * ((W + 127) // 128) * ((H + 7) // 8) >= max(4, min(64, min_tiles)))


def head_upfeat_workspace_bytes(B: int, cin: int, H: int, W: int) -> int:
    """Scratch bytes pwc_head_upfeat_ws_fwd wants for this geometry (Cin slices of launches smaller than the chip; 0 = never)."""
    n = _lib.load().pwc_head_upfeat_workspace_bytes(B, cin, H, W)
    if n < 0:
        raise ValueError("bad head geometry")
    return int(n)


def head_upfeat(x: torch.Tensor, head_wpacked: torch.Tensor, head_bias: torch.Tensor, up_weight: torch.Tensor,
                up_bias: torch.Tensor, flow_out: torch.Tensor, up_out: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> None:
    """predict_flowL + upfeatL in one pass over the arena x (fused C-ABI entry pwc_head_upfeat_ws_fwd; `workspace`: scratch for the
    Cin slices of launches smaller than the chip, see head_upfeat_workspace_bytes)."""
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    B, cin, H, W = x.shape
    if tuple(flow_out.shape) != (B, 2, H, W) or tuple(up_out.shape) != (B, 2, 2 * H, 2 * W):
        raise ValueError("flow_out must be %s and up_out %s" % ((B, 2, H, W), (B, 2, 2 * H, 2 * W)))
    if tuple(up_weight.shape) != (cin, 2, 4, 4) or not up_weight.is_contiguous():
        raise ValueError("up_weight must be contiguous [Cin=%d,2,4,4]" % cin)
    need = lib.pwc_conv3x3_packed_bytes(cin, 2, PWC_F32)
    if head_wpacked.numel() * 4 != need:
        raise ValueError("packed head weights do not match Cin=%d" % cin)
    bsf = _plane_dense(flow_out, "flow_out")
    bsu = _plane_dense(up_out, "up_out")
    with torch.cuda.device(x.device):
        ws_ptr, ws_bytes = _workspace_args(workspace, x)
        rc = lib.pwc_head_upfeat_ws_fwd(x.data_ptr(), head_wpacked.data_ptr(), head_bias.data_ptr(), flow_out.data_ptr(),
                                        up_weight.data_ptr(), up_bias.data_ptr(), up_out.data_ptr(),
                                        B, cin, H, W, _dtype_code(x), bsx, bsf, bsu, ws_ptr, ws_bytes, _stream(x))
    check(rc, "pwc_head_upfeat_ws_fwd")


def deconv_as_conv3x3(wt: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(k4, s2, p1) filters [Cin, Cout, 4, 4] -> 3x3 conv filters [Cout*4, Cin, 3, 3], output channel
    co*4 + py*2 + px = phase (py, px) of output pixel (2*iy+py, 2*ix+px):
    py = 0 takes window rows a = 0, 1 with ky = 3, 1;  py = 1 takes a = 1, 2 with ky = 2, 0  (same along x)."""
    cin, cout = wt.shape[:2]
    k = wt.new_zeros((cout, 2, 2, cin, 3, 3))
    taps = {0: ((0, 3), (1, 1)), 1: ((1, 2), (2, 0))}          # phase -> ((window index, kernel index), ...)
    for py in (0, 1):
        for a, ky in taps[py]:
            for px in (0, 1):
                for e, kx in taps[px]:
                    k[:, py, px, :, a, e] = wt[:, :, ky, kx].t()
    return k.reshape(cout * 4, cin, 3, 3)


def upsample_entry(head: torch.Tensor, deconv_w: torch.Tensor, deconv_b: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Exit of a decoder level whose flow head and upfeat ran as one 10-channel 3x3 convolution: head [B,10,h,w] = [flow | upfeat
    phases] -> out [B,4,2h,2w] = [deconvL(flow) | up_feat] (C-ABI pwc_upsample_entry_f32; PWCNet.py:208-209)."""
    lib = _lib.load()
    B, c, h, w = head.shape
    if c != 10 or head.dtype != torch.float32 or tuple(out.shape) != (B, 4, 2 * h, 2 * w) or out.dtype != torch.float32:
        raise ValueError("head must be float32 [B,10,h,w] and out [B,4,2h,2w], got %s / %s" % (tuple(head.shape), tuple(out.shape)))
    for t, n, shp in ((deconv_w, "deconv_w", (2, 2, 4, 4)), (deconv_b, "deconv_b", (2,))):
        if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != head.device:
            raise ValueError("%s must be contiguous float32 %s on %s" % (n, shp, head.device))
    bsh, bso = _plane_dense(head, "head"), _plane_dense(out, "out")
    with torch.cuda.device(head.device):
        rc = lib.pwc_upsample_entry_f32(head.data_ptr(), deconv_w.data_ptr(), deconv_b.data_ptr(), out.data_ptr(), B, h, w, bsh, bso, _stream(head))
    check(rc, "pwc_upsample_entry_f32")
    return out


def deconv4x4s2(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.ConvTranspose2d(k=4, s=2, p=1); weight [Cin,Cout,4,4] float32 contiguous."""
    lib = _lib.load()
    bsx = _plane_dense(x, "x")
    B, cin, H, W = x.shape
    if weight.dim() != 4 or weight.shape[0] != cin or weight.shape[2:] != (4, 4) or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [Cin=%d,Cout,4,4], got %s" % (cin, tuple(weight.shape)))
    cout = weight.shape[1]
    if out is None:
        out = torch.empty((B, cout, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
    elif tuple(out.shape) != (B, cout, 2 * H, 2 * W):
        raise ValueError("out must be %s" % ((B, cout, 2 * H, 2 * W),))
    bsy = _plane_dense(out, "out")
    with torch.cuda.device(x.device):
        rc = lib.pwc_deconv4x4s2_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                     B, cin, H, W, cout, _dtype_code(x), bsx, bsy, _stream(x))
    check(rc, "pwc_deconv4x4s2_fwd")
    return out
