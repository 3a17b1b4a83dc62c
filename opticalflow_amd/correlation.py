"""Correlation operator with the reference's signature, backed by libpwc_hip.so.

Mirrors ``models/correlation_package/correlation.py:92-117`` of the reference:
``Correlation(pad_size, kernel_size, max_displacement, stride1, stride2,
corr_multiply=1)`` is an ``nn.Module`` whose ``forward(input1, input2)``
returns the ``[B, (2*(d//s2)+1)**2, H', W']`` cost volume.

Additive keyword (not in the reference): ``normalize``.
  * ``normalize=False`` (default): raw channel sum times ``corr_multiply`` -- the
    semantics of the reference's pure-PyTorch path (correlation.py:35-36), which
    is the parity definition of this project.
  * ``normalize=True``: divide by ``kernel_size**2 * C`` like the reference's
    CUDA kernel (correlation_cuda_kernel.cu:104,143) -- what published PWC-Net
    weights were trained with.

Device tensors only: there is no CPU implementation in the product.  The
reference's global ``USE_ONNX_CORRELATION`` (correlation.py:9) is kept as a
name for source compatibility; setting it selects a traceable torch-op
expression for exporters and is never chosen implicitly.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops

import sys

# Same global switch name as the reference (correlation.py:9).  Off by default.
USE_ONNX_CORRELATION = False

_SHIM = "models.correlation_package.correlation"


def onnx_correlation_enabled() -> bool:
    """True when the reference's export switch is on, set EITHER here or -- the way the reference's own callers do it
    (pth2onnx.py:44-46, onnx_pth_compare.py:91-93: ``corr_mod.USE_ONNX_CORRELATION = True`` on the module they import as
    ``models.correlation_package.correlation``) -- as a plain module attribute of the import-path shim."""
    if USE_ONNX_CORRELATION:
        return True
    shim = sys.modules.get(_SHIM)
    return bool(shim is not None and shim.__dict__.get("USE_ONNX_CORRELATION", False))


def correlation_traceable(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                          corr_multiply, normalize=False):
    """Exporter-only expression of the k=1, stride1=1 cost volume out of traceable torch ops.

    Exists because the reference flips USE_ONNX_CORRELATION in pth2onnx.py:46 to make the graph
    exportable; it is opt-in, runs on whatever device the tensors live on and is not used by any
    parity or performance path of this project.
    """
    if kernel_size != 1 or stride1 != 1:
        raise NotImplementedError("traceable correlation covers kernel_size=1, stride1=1 only")
    _, C, H, W = input1.shape
    reach = pad_size + max_displacement
    padded = F.pad(input2, (reach, reach, reach, reach))
    planes = []
    for oy in range(-max_displacement, max_displacement + 1, stride2):
        for ox in range(-max_displacement, max_displacement + 1, stride2):
            window = padded[:, :, reach + oy:reach + oy + H, reach + ox:reach + ox + W]
            planes.append((input1 * window).sum(dim=1, keepdim=True))
    out = torch.cat(planes, dim=1)
    return out / float(C) if normalize else out * corr_multiply


class CorrelationFunction(Function):
    """autograd wrapper: forward and backward both run HIP kernels (correlation.py:43-89)."""

    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1, normalize=False):
        ctx.save_for_backward(input1, input2)
        ctx.cfg = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply, normalize)
        return ops.correlation(input1.contiguous(), input2.contiguous(), pad_size, kernel_size, max_displacement,
                               stride1, stride2, corr_multiply, normalize=normalize)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply, normalize = ctx.cfg
        g1, g2 = ops.correlation_backward(input1.contiguous(), input2.contiguous(), grad_output.contiguous(),
                                          pad_size, kernel_size, max_displacement, stride1, stride2,
                                          corr_multiply, normalize=normalize)
        return g1, g2, None, None, None, None, None, None, None


class Correlation(nn.Module):
    def __init__(self, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply=1,
                 normalize=False):
        super().__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply
        self.normalize = normalize

    def extra_repr(self):
        return "pad_size=%d, kernel_size=%d, max_displacement=%d, stride1=%d, stride2=%d, corr_multiply=%s, normalize=%s" % (
            self.pad_size, self.kernel_size, self.max_displacement, self.stride1, self.stride2,
            self.corr_multiply, self.normalize)

    def forward(self, input1, input2):
        if onnx_correlation_enabled():
            return correlation_traceable(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                         self.stride1, self.stride2, self.corr_multiply, self.normalize)
        if torch.is_grad_enabled() and (input1.requires_grad or input2.requires_grad):
            return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                             self.stride1, self.stride2, self.corr_multiply, self.normalize)
        return ops.correlation(ops.densify(input1), ops.densify(input2), self.pad_size, self.kernel_size,
                               self.max_displacement, self.stride1, self.stride2, self.corr_multiply,
                               normalize=self.normalize)
