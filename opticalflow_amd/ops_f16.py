"""fp16 building blocks (first piece of the half-precision path, BASELINE configs 3-4): channel-blocked "c8"
activations ``[B, ceil(C/8), H, W, 8]`` (torch.float16) and the MFMA 3x3 convolution over them.

Device tensors only; every function launches kernels of libpwc_hip.so on the current stream (no CPU path).
The full fp16 network is not assembled yet: correlation / warp / heads in c8 layout are the next round's work.
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


def _c8_bstride(t: torch.Tensor, name: str) -> int:
    """c8 tensors must have dense [Cg,H,W,8] planes; the batch stride is free (channel-group slices of an arena)."""
    if t.dim() != 5 or t.shape[4] != 8 or t.dtype != torch.float16:
        raise ValueError("%s must be a float16 [B,Cg,H,W,8] tensor, got %s %s" % (name, t.dtype, tuple(t.shape)))
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


def pack_conv3x3_f16(weight: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] float32 device tensor -> packed float16 filter bank for conv3x3_f16."""
    _require_device(weight, "weight")
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.dtype != torch.float32:
        raise ValueError("expected float32 [Cout,Cin,3,3], got %s %s" % (weight.dtype, tuple(weight.shape)))
    lib = _lib.load()
    cout, cin = weight.shape[:2]
    nbytes = lib.pwc_conv3x3_f16_packed_bytes(cin, cout)
    wp = torch.empty((nbytes // 2,), dtype=torch.float16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.pwc_conv3x3_f16_pack(weight.contiguous().data_ptr(), wp.data_ptr(), cin, cout, _stream(weight))
    check(rc, "pwc_conv3x3_f16_pack")
    return wp


def conv3x3_f16(x: torch.Tensor, wpacked: torch.Tensor, bias: torch.Tensor, cin: int, cout: int, stride: int = 1,
                dilation: int = 1, leaky_slope: Optional[float] = 0.1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 convolution (padding = dilation) + bias (+ LeakyReLU) on c8 float16 activations, fp32 accumulation."""
    lib = _lib.load()
    bsx = _c8_bstride(x, "x")
    B, cg, H, W, _ = x.shape
    if cg != (cin + 7) // 8:
        raise ValueError("x has %d channel groups, Cin=%d needs %d" % (cg, cin, (cin + 7) // 8))
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty(c8_shape(B, cout, ho, wo), dtype=torch.float16, device=x.device)
    elif tuple(out.shape) != c8_shape(B, cout, ho, wo):
        raise ValueError("out must be %s" % (c8_shape(B, cout, ho, wo),))
    bsy = _c8_bstride(out, "out")
    need = lib.pwc_conv3x3_f16_packed_bytes(cin, cout)
    if wpacked.dtype != torch.float16 or wpacked.numel() * 2 != need or wpacked.device != x.device:
        raise ValueError("packed filters do not match Cin=%d Cout=%d" % (cin, cout))
    if bias.dtype != torch.float32 or bias.numel() != cout or bias.device != x.device or not bias.is_contiguous():
        raise ValueError("bias must be float32[%d] on %s" % (cout, x.device))
    with torch.cuda.device(x.device):
        rc = lib.pwc_conv2d_f16_fwd(x.data_ptr(), wpacked.data_ptr(), bias.data_ptr(), out.data_ptr(), B, cin, H, W, cout,
                                    stride, dilation, FLAG_ACT_LEAKY if leaky_slope is not None else 0,
                                    float(leaky_slope or 0.0), bsx, bsy, _stream(x))
    check(rc, "pwc_conv2d_f16_fwd")
    return out
