"""Import-path compatibility: ``from models.correlation_package.correlation import Correlation``
(reference models/PWCNet.py:14).  The flag is shared with the implementation module."""
import opticalflow_amd.correlation as _impl
from opticalflow_amd.correlation import Correlation, CorrelationFunction  # noqa: F401


def __getattr__(name):
    if name == "USE_ONNX_CORRELATION":
        return _impl.USE_ONNX_CORRELATION
    raise AttributeError(name)
