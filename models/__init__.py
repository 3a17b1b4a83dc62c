"""Drop-in for the reference's ``models`` package (reference models/__init__.py:1):
``import models; net = models.pwc_dc_net('pwc_net.pth.tar')`` resolves to the MI355X path."""
from .PWCNet import *  # noqa: F401,F403
