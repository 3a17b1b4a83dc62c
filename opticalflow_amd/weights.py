"""Checkpoint loading and synthetic weights for PWCDCNet.

Checkpoint layouts accepted ("Checkpoint / resume"):
  * bare state-dict                                   (models/PWCNet.py:504-505)
  * ``{'state_dict': ...}``                            (models/PWCNet.py:501-503)
  * ``{'model': ...}``                                 (train.py:136, pwc_extract_flow.py:133)
  * any of the above with a ``module.`` key prefix     (pwc_extract_flow.py:137)
Files are read with ``torch.load(..., weights_only=True)``: nothing in a
checkpoint is executed.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import torch


def unwrap_checkpoint(data) -> Dict[str, torch.Tensor]:
    if not isinstance(data, dict):
        raise TypeError("checkpoint must be a dict, got %s" % type(data).__name__)
    for key in ("state_dict", "model"):
        if key in data and isinstance(data[key], dict):
            data = data[key]
            break
    out = {}
    for k, v in data.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, torch.Tensor]:
    return unwrap_checkpoint(torch.load(path, map_location=map_location, weights_only=True))


def _fan_in(key: str, shape: Tuple[int, ...]) -> int:
    # nn.init.kaiming_normal_(mode='fan_in') uses size(1) * receptive field for both Conv2d
    # ([Cout,Cin,k,k]) and ConvTranspose2d ([Cin,Cout,k,k]) weights (PWCNet.py:134-138).
    return shape[1] * shape[2] * shape[3]


def synthetic_state_dict(manifest: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, gain: float = 1.0,
                         bias_std: float = 0.0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic Kaiming-fan-in weights, one generator draw per tensor in manifest order.

    No checkpoint or dataset can be fetched in this project's environments, so benches and goldens
    use these: std = gain * sqrt(2 / fan_in) like the reference's init (PWCNet.py:134-138), optional
    non-zero biases so the bias path is exercised.  ``gain`` < 1 keeps |flow| = O(1) at 1024x448
    (SURVEY.md section 0, fact 10).
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in manifest:
        if key.endswith(".weight"):
            std = gain * math.sqrt(2.0 / _fan_in(key, shape))
            sd[key] = (torch.randn(shape, generator=g, dtype=torch.float32) * std).to(dtype)
        else:
            sd[key] = (torch.randn(shape, generator=g, dtype=torch.float32) * bias_std).to(dtype)
    return sd
