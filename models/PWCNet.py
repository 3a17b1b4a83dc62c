"""Name-compatible entry points of the reference's models/PWCNet.py (``__all__`` at :22-24).

The implementation lives in ``opticalflow_amd.pwcnet``: ``pwc_dc_net`` (PWCNet.py:497-506) builds the
128-key PWCDCNet, ``pwc_dc_net_old`` (PWCNet.py:511-520) the 116-key legacy variant PWCDCNet_old.

Everything imported from HERE mirrors the reference's module, whose ``self.corr`` is the native operator that divides
by ``kernel_size**2 * C`` (correlation_cuda_kernel.cu:104,143): the classes below default to ``normalize_corr=True``,
so an unchanged caller that does ``from models.PWCNet import PWCDCNet; net = PWCDCNet(); net.load_state_dict(ckpt)``
(inference_kitti.py:301, inference.py:328, pwc_extract_flow.py:129) runs a trained checkpoint with the cost volumes
it was trained on -- the same semantics as ``models.correlation_package.correlation.Correlation``.
``opticalflow_amd.PWCDCNet`` keeps the un-normalised parity default (the reference's CPU fallback)."""
from opticalflow_amd import pwcnet as _impl
from opticalflow_amd.weights import load_checkpoint as _load_checkpoint

__all__ = ["pwc_dc_net", "pwc_dc_net_old"]


class PWCDCNet(_impl.PWCDCNet):
    def __init__(self, md=4, **kwargs):
        kwargs.setdefault("normalize_corr", True)
        super().__init__(md, **kwargs)


class PWCDCNet_old(_impl.PWCDCNet_old):
    def __init__(self, md=4, **kwargs):
        kwargs.setdefault("normalize_corr", True)
        super().__init__(md, **kwargs)


def pwc_dc_net(path=None, **kwargs):
    """PWCNet.py:497-506: ``model = PWCDCNet(); if path is not None: load_state_dict`` (bare or ``{'state_dict': ..}``)."""
    model = PWCDCNet(**_impl._checkpoint_kwargs(path, kwargs))
    if path is not None:
        model.load_state_dict(_load_checkpoint(path))
    return model


def pwc_dc_net_old(path=None, **kwargs):
    """PWCNet.py:511-520."""
    model = PWCDCNet_old(**_impl._checkpoint_kwargs(path, kwargs))
    if path is not None:
        model.load_state_dict(_load_checkpoint(path))
    return model
