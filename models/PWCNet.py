"""Name-compatible entry points of the reference's models/PWCNet.py (``__all__`` at :22-24).

The implementation lives in ``opticalflow_amd.pwcnet``.  ``pwc_dc_net_old`` (the 116-key legacy
variant, reference PWCNet.py:277-491) is used by none of the reference's scripts and is not built;
asking for it fails loudly rather than silently substituting the new network.
"""
from opticalflow_amd.pwcnet import PWCDCNet, pwc_dc_net  # noqa: F401

__all__ = ["pwc_dc_net", "pwc_dc_net_old"]


def pwc_dc_net_old(path=None):
    raise NotImplementedError("PWCDCNet_old (reference models/PWCNet.py:277-491) is out of scope of the "
                              "MI355X inference path; use pwc_dc_net")
