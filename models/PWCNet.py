"""Name-compatible entry points of the reference's models/PWCNet.py (``__all__`` at :22-24).

The implementation lives in ``opticalflow_amd.pwcnet``: ``pwc_dc_net`` (PWCNet.py:497-506) builds the
128-key PWCDCNet, ``pwc_dc_net_old`` (PWCNet.py:511-520) the 116-key legacy variant PWCDCNet_old.
"""
from opticalflow_amd.pwcnet import PWCDCNet, PWCDCNet_old, pwc_dc_net, pwc_dc_net_old  # noqa: F401

__all__ = ["pwc_dc_net", "pwc_dc_net_old"]
