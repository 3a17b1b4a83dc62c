"""Import-path compatibility: ``from models.correlation_package.correlation import Correlation``
(reference models/PWCNet.py:14).

Semantics follow the reference's own module (correlation.py:92-117): with ``USE_ONNX_CORRELATION`` off (the default)
``Correlation.forward`` is the NATIVE operator, which divides by ``kernel_size**2 * C``
(correlation_cuda_kernel.cu:104,143) -- here the HIP kernel with ``normalize=True``; with the flag on it is the
un-normalised torch-op expression (correlation.py:12-40).  ``opticalflow_amd.Correlation`` (additive ``normalize``
keyword, un-normalised by default) is the same operator with this project's parity default.

``USE_ONNX_CORRELATION`` is an ordinary module global exactly as in the reference (correlation.py:9), so the reference's
callers' ``corr_mod.USE_ONNX_CORRELATION = True`` (pth2onnx.py:44-46, onnx_pth_compare.py:91-93) takes effect:
``Correlation.forward`` below reads it, and ``PWCDCNet.forward`` reads it through
``opticalflow_amd.correlation.onnx_correlation_enabled`` (its cost volumes are then un-normalised, like the
reference's net with the flag on)."""
import opticalflow_amd.correlation as _impl
from opticalflow_amd.correlation import CorrelationFunction  # noqa: F401

USE_ONNX_CORRELATION = False


class Correlation(_impl.Correlation):
    def __init__(self, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply=1):
        super().__init__(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply, normalize=True)

    def forward(self, input1, input2):
        if USE_ONNX_CORRELATION or _impl.USE_ONNX_CORRELATION:      # reference fallback: raw channel sum * corr_multiply
            return _impl.correlation_traceable(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                               self.stride1, self.stride2, self.corr_multiply, normalize=False)
        return _impl.Correlation.forward(self, input1, input2)
