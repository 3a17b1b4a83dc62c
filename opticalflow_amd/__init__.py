"""opticalflow_amd -- MI355X-native PWC-Net inference path (hand-written gfx950 HIP kernels behind
the reference's Correlation / warp / PWCDCNet interface).  See DESIGN.md."""
from ._lib import LIB_PATH, PwcHipError  # noqa: F401
from .correlation import Correlation, CorrelationFunction  # noqa: F401
from .pwcnet import PWCDCNet, PWCDCNet_old, pwc_dc_net, pwc_dc_net_old  # noqa: F401
from .flowio import read_flo, write_flo  # noqa: F401

__version__ = "0.1.0"
