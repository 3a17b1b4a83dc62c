"""Middlebury ``.flo`` container, the on-disk format of the reference's single-pair CLI.

Layout (script_pwc.py:12-27 writer; data_processing.py:17-29 and pwc_extract_flow.py:46-56 agree):
    float32  202021.25   (tag; little-endian bytes 'PIEH')
    int32    W
    int32    H
    float32  H*W*2       row-major, (u, v) interleaved per pixel
"""
from __future__ import annotations

import numpy as np

TAG_FLOAT = 202021.25


def write_flo(filename: str, uv) -> None:
    """Write an [H,W,2] flow field (numpy array or CPU tensor)."""
    uv = np.asarray(uv.detach().cpu().numpy() if hasattr(uv, "detach") else uv)
    if uv.ndim != 3 or uv.shape[2] != 2:
        raise ValueError("write_flo: flow must be [H,W,2], got %s" % (uv.shape,))
    h, w = uv.shape[:2]
    with open(filename, "wb") as f:
        np.array(TAG_FLOAT, dtype="<f4").tofile(f)
        np.array([w, h], dtype="<i4").tofile(f)
        np.ascontiguousarray(uv, dtype="<f4").tofile(f)


def read_flo(filename: str) -> np.ndarray:
    """Read a ``.flo`` file into an [H,W,2] float32 array; raises on a bad tag or a short file."""
    with open(filename, "rb") as f:
        tag = np.fromfile(f, dtype="<f4", count=1)
        if tag.size != 1 or float(tag[0]) != TAG_FLOAT:
            raise ValueError("read_flo: %s is not a .flo file (tag %r)" % (filename, tag))
        wh = np.fromfile(f, dtype="<i4", count=2)
        if wh.size != 2 or wh[0] <= 0 or wh[1] <= 0:
            raise ValueError("read_flo: bad header in %s" % filename)
        w, h = int(wh[0]), int(wh[1])
        data = np.fromfile(f, dtype="<f4", count=2 * w * h)
        if data.size != 2 * w * h:
            raise ValueError("read_flo: %s truncated (%d of %d values)" % (filename, data.size, 2 * w * h))
    return data.reshape(h, w, 2)
