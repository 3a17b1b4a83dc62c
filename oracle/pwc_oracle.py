"""CPU oracle for the PWC-Net inference path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``opticalflow_amd/`` imports from ``oracle/``.

It restates, on the CPU, the algorithm of the reference hot path:

* ``correlation``      <- reference ``models/correlation_package/correlation.py:12-40``
  (un-normalised fallback, the parity definition) and, with ``normalize=True``,
  ``correlation_cuda_kernel.cu:73-147`` (``/ (kernel_size**2 * C)``, k x k
  patch sum, stride1 / stride2).
* ``warp``             <- reference ``models/PWCNet.py:141-177`` as executed by
  torch >= 1.3 (``grid_sample`` default ``align_corners=False``): written as
  explicit bilinear arithmetic, not as a call to ``grid_sample``.
* ``pwc_forward``      <- reference ``models/PWCNet.py:180-273`` (functional,
  driven by a state-dict with the reference's 128 keys).
* ``pwc_forward_old`` / ``state_dict_manifest_old`` <- ``models/PWCNet.py:277-491`` (``PWCDCNet_old``:
  no ``*aa`` pyramid convs, mixed concatenation order, warp mask threshold 0.999).
* ``read_flo`` / ``write_flo`` <- ``script_pwc.py:12-27`` / ``data_processing.py:17-29``.

Pinning: the reference holds no golden vectors of its own,
so the oracle is pinned by fixtures produced by importing the reference's own
Python in the build container (``oracle/gen_golden.py`` -> ``tests/golden``);
``tests/test_oracle_golden.py`` checks every function here against them.
"""
from __future__ import annotations

import math
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

FLO_TAG = 202021.25  # script_pwc.py:18

# --------------------------------------------------------------------------
# correlation
# --------------------------------------------------------------------------

def corr_output_shape(C: int, H: int, W: int, pad_size: int, kernel_size: int,
                      max_displacement: int, stride1: int, stride2: int) -> Tuple[int, int, int]:
    """Shape contract of correlation_cuda.cc:25-38."""
    krad = (kernel_size - 1) // 2
    border = krad + max_displacement
    drad = max_displacement // stride2
    nch = (2 * drad + 1) ** 2
    oh = int(math.ceil(float(H + 2 * pad_size - 2 * border) / float(stride1)))
    ow = int(math.ceil(float(W + 2 * pad_size - 2 * border) / float(stride1)))
    return nch, oh, ow


def correlation(in1: torch.Tensor, in2: torch.Tensor, pad_size: int = 4, kernel_size: int = 1,
                max_displacement: int = 4, stride1: int = 1, stride2: int = 1,
                corr_multiply: float = 1, normalize: bool = False) -> torch.Tensor:
    """Cost volume.

    out[b, (tj+r)*D + (ti+r), y, x] =
        sum_{j,i in k x k} sum_c in1p[b,c,y1+j,x1+i] * in2p[b,c,y1+tj*s2+j,x1+ti*s2+i]
    with in*p zero-padded by pad_size, y1 = y*s1 + max_displacement (same for x),
    r = max_displacement // s2, D = 2r+1  (correlation_cuda_kernel.cu:93-141).

    normalize=False : raw sum * corr_multiply    (correlation.py:35-36)
    normalize=True  : sum / (k*k*C)              (correlation_cuda_kernel.cu:104,143)
    """
    B, C, H, W = in1.shape
    nch, oh, ow = corr_output_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    krad = (kernel_size - 1) // 2
    drad = max_displacement // stride2
    D = 2 * drad + 1
    p1 = F.pad(in1, (pad_size,) * 4)
    p2 = F.pad(in2, (pad_size,) * 4)
    out = in1.new_zeros((B, nch, oh, ow))
    ys = max_displacement + stride1 * torch.arange(oh)
    xs = max_displacement + stride1 * torch.arange(ow)
    for tj in range(-drad, drad + 1):
        for ti in range(-drad, drad + 1):
            acc = in1.new_zeros((B, oh, ow))
            for j in range(-krad, krad + 1):
                for i in range(-krad, krad + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + tj * stride2 + j)[:, None], (xs + ti * stride2 + i)[None, :]]
                    acc = acc + (a * b).sum(dim=1)
            out[:, (tj + drad) * D + (ti + drad)] = acc
    if normalize:
        out = out / float(kernel_size * kernel_size * C)
    else:
        out = out * corr_multiply
    return out


def correlation_loops(in1: np.ndarray, in2: np.ndarray, max_displacement: int = 4) -> np.ndarray:
    """Plain-loop fp64 statement of the PWC configuration (pad=d, k=1, s1=s2=1); small cases only."""
    B, C, H, W = in1.shape
    d = max_displacement
    D = 2 * d + 1
    out = np.zeros((B, D * D, H, W), dtype=np.float64)
    for b in range(B):
        for dy in range(-d, d + 1):
            for dx in range(-d, d + 1):
                ch = (dy + d) * D + (dx + d)
                for y in range(H):
                    yy = y + dy
                    if yy < 0 or yy >= H:
                        continue
                    for x in range(W):
                        xx = x + dx
                        if xx < 0 or xx >= W:
                            continue
                        out[b, ch, y, x] = np.dot(in1[b, :, y, x].astype(np.float64),
                                                  in2[b, :, yy, xx].astype(np.float64))
    return out


def leaky_relu(x: torch.Tensor, slope: float = 0.1) -> torch.Tensor:
    """PWCNet.py:72 (nn.LeakyReLU(0.1))."""
    return torch.where(x > 0, x, x * slope)


# --------------------------------------------------------------------------
# warp
# --------------------------------------------------------------------------

def warp(x: torch.Tensor, flo: torch.Tensor, align_corners: bool = False,
         mask_threshold: float = 0.9999) -> torch.Tensor:
    """Backward-warp x by flo with bilinear taps, zero padding and validity mask.

    PWCNet.py:162-163 maps pixel coordinate (x+u) to g = 2(x+u)/max(W-1,1) - 1.
    grid_sample then un-normalises with align_corners=False (torch >= 1.3
    default; PWCNet.py:166 passes no argument): ix = ((g+1)*W - 1)/2.
    align_corners=True gives ix = (g+1)/2*(W-1) = x+u (the behaviour the
    published weights were trained with).
    mask = grid_sample(ones) -> [<0.9999]=0, [>0]=1   (PWCNet.py:167-175).
    """
    B, C, H, W = x.shape
    dt = x.dtype
    gx = torch.arange(W, dtype=dt).view(1, 1, W) + flo[:, 0]
    gy = torch.arange(H, dtype=dt).view(1, H, 1) + flo[:, 1]
    nx = 2.0 * gx / max(W - 1, 1) - 1.0
    ny = 2.0 * gy / max(H - 1, 1) - 1.0
    if align_corners:
        ix = (nx + 1) / 2 * (W - 1)
        iy = (ny + 1) / 2 * (H - 1)
    else:
        ix = ((nx + 1) * W - 1) / 2
        iy = ((ny + 1) * H - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    wx1 = ix - x0
    wy1 = iy - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    x0 = x0.long()
    y0 = y0.long()
    out = x.new_zeros((B, C, H, W))
    msum = x.new_zeros((B, H, W))
    flat = x.reshape(B, C, H * W)
    for (yy, wy) in ((y0, wy0), (y0 + 1, wy1)):
        for (xx, wx) in ((x0, wx0), (x0 + 1, wx1)):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            w = (wy * wx) * ok.to(dt)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(B, 1, H * W).expand(B, C, H * W)
            out = out + torch.gather(flat, 2, idx).reshape(B, C, H, W) * w.unsqueeze(1)
            msum = msum + w
    mask = (msum >= mask_threshold).to(dt)
    return out * mask.unsqueeze(1)


# --------------------------------------------------------------------------
# full forward (functional, state-dict driven)
# --------------------------------------------------------------------------

PYRAMID = (("conv1a", 2), ("conv1aa", 1), ("conv1b", 1),
           ("conv2a", 2), ("conv2aa", 1), ("conv2b", 1),
           ("conv3a", 2), ("conv3aa", 1), ("conv3b", 1),
           ("conv4a", 2), ("conv4aa", 1), ("conv4b", 1),
           ("conv5a", 2), ("conv5aa", 1), ("conv5b", 1),
           ("conv6aa", 2), ("conv6a", 1), ("conv6b", 1))          # PWCNet.py:52-69, order of use :184-195
WARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}                 # PWCNet.py:212,226,240,256
DILATIONS = (1, 2, 4, 8, 16, 1)                                   # PWCNet.py:126-131


def _conv(sd: Dict[str, torch.Tensor], name: str, x: torch.Tensor, stride: int = 1, dilation: int = 1,
          act: bool = True) -> torch.Tensor:
    """conv() helper of PWCNet.py:26-30 (Sequential -> key suffix '.0') or predict_flow :32-33."""
    key = name + ".0" if (name + ".0.weight") in sd else name
    y = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=dilation, dilation=dilation)
    return leaky_relu(y) if act else y


def _deconv(sd, name, x):
    """deconv() helper PWCNet.py:35-36: ConvTranspose2d(k=4, s=2, p=1)."""
    return F.conv_transpose2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=2, padding=1)


def pwc_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, normalize_corr: bool = False,
                align_corners: bool = False, all_levels: bool = False, md: int = 4):
    """PWCDCNet.forward (PWCNet.py:180-273).  Returns flow2, or (flow2..flow6) if all_levels."""
    feats = []
    for im in (x[:, :3], x[:, 3:]):
        pyr = []
        t = im
        for i, (name, stride) in enumerate(PYRAMID):
            t = _conv(sd, name, t, stride=stride)
            if i % 3 == 2:
                pyr.append(t)
        feats.append(pyr)            # index 0 -> level 1 ... index 5 -> level 6
    flows = {}
    up_flow = up_feat = None
    xcat = None
    for lvl in (6, 5, 4, 3, 2):
        c1 = feats[0][lvl - 1]
        c2 = feats[1][lvl - 1]
        if lvl == 6:
            corr = leaky_relu(correlation(c1, c2, md, 1, md, 1, 1, 1, normalize=normalize_corr))
            xcat = corr
        else:
            w = warp(c2, up_flow * WARP_SCALE[lvl], align_corners=align_corners)
            corr = leaky_relu(correlation(c1, w, md, 1, md, 1, 1, 1, normalize=normalize_corr))
            xcat = torch.cat((corr, c1, up_flow, up_feat), 1)
        for i in range(5):
            xcat = torch.cat((_conv(sd, "conv%d_%d" % (lvl, i), xcat), xcat), 1)
        flow = _conv(sd, "predict_flow%d" % lvl, xcat, act=False)
        flows[lvl] = flow
        if lvl > 2:
            up_flow = _deconv(sd, "deconv%d" % lvl, flow)
            up_feat = _deconv(sd, "upfeat%d" % lvl, xcat)
    t = xcat
    for i, dil in enumerate(DILATIONS):
        t = _conv(sd, "dc_conv%d" % (i + 1), t, dilation=dil)
    flow2 = flows[2] + _conv(sd, "dc_conv7", t, act=False)
    if all_levels:
        return flow2, flows[3], flows[4], flows[5], flows[6]
    return flow2


PYRAMID_OLD = (("conv1a", 2), ("conv1b", 1), ("conv2a", 2), ("conv2b", 1), ("conv3a", 2), ("conv3b", 1),
               ("conv4a", 2), ("conv4b", 1), ("conv5a", 2), ("conv5b", 1), ("conv6a", 2), ("conv6b", 1))   # PWCNet.py:290-301
OLD_MASK_THRESHOLD = 0.999                                                                                 # PWCNet.py:400


def pwc_forward_old(sd: Dict[str, torch.Tensor], x: torch.Tensor, normalize_corr: bool = False,
                    align_corners: bool = False, all_levels: bool = False, md: int = 4):
    """PWCDCNet_old.forward (PWCNet.py:407-491).  Concatenation order per level (PWCNet.py:425-429 etc.):
    x = cat(x, conv_0(x)); x = cat(conv_1(x), x); x = cat(x, conv_2(x)); x = cat(x, conv_3(x)); x = cat(x, conv_4(x))."""
    feats = []
    for im in (x[:, :3], x[:, 3:]):
        pyr = []
        t = im
        for i, (name, stride) in enumerate(PYRAMID_OLD):
            t = _conv(sd, name, t, stride=stride)
            if i % 2 == 1:
                pyr.append(t)
        feats.append(pyr)
    flows = {}
    up_flow = up_feat = None
    xcat = None
    for lvl in (6, 5, 4, 3, 2):
        c1 = feats[0][lvl - 1]
        c2 = feats[1][lvl - 1]
        if lvl == 6:
            xcat = leaky_relu(correlation(c1, c2, md, 1, md, 1, 1, 1, normalize=normalize_corr))
        else:
            w = warp(c2, up_flow * WARP_SCALE[lvl], align_corners=align_corners, mask_threshold=OLD_MASK_THRESHOLD)
            corr = leaky_relu(correlation(c1, w, md, 1, md, 1, 1, 1, normalize=normalize_corr))
            xcat = torch.cat((corr, c1, up_flow, up_feat), 1)
        xcat = torch.cat((xcat, _conv(sd, "conv%d_0" % lvl, xcat)), 1)
        xcat = torch.cat((_conv(sd, "conv%d_1" % lvl, xcat), xcat), 1)
        for i in (2, 3, 4):
            xcat = torch.cat((xcat, _conv(sd, "conv%d_%d" % (lvl, i), xcat)), 1)
        flow = _conv(sd, "predict_flow%d" % lvl, xcat, act=False)
        flows[lvl] = flow
        if lvl > 2:
            up_flow = _deconv(sd, "deconv%d" % lvl, flow)
            up_feat = _deconv(sd, "upfeat%d" % lvl, xcat)
    t = xcat
    for i, dil in enumerate(DILATIONS):
        t = _conv(sd, "dc_conv%d" % (i + 1), t, dilation=dil)
    flow2 = flows[2] + _conv(sd, "dc_conv7", t, act=False)
    if all_levels:
        return flow2, flows[3], flows[4], flows[5], flows[6]
    return flow2


# --------------------------------------------------------------------------
# state-dict manifest (PWCNet.py:52-132) -- 128 (key, shape) pairs
# --------------------------------------------------------------------------

def state_dict_manifest_old(md: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    """PWCDCNet_old (PWCNet.py:288-366): the 128-key manifest minus the six *aa pyramid convs -> 116 keys."""
    drop = {"conv1aa", "conv2aa", "conv3aa", "conv4aa", "conv5aa"}
    out = []
    for k, shp in state_dict_manifest(md):
        head = k.split(".")[0]
        if head in drop or head == "conv6a":
            continue                                   # conv6aa (128->196, stride 2) takes the name conv6a below
        if head == "conv6aa":
            k = "conv6a" + k[len("conv6aa"):]
        out.append((k, shp))
    return out


def state_dict_manifest(md: int = 4) -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, cin, cout, seq=True, k=3):
        base = name + (".0" if seq else "")
        out.append((base + ".weight", (cout, cin, k, k)))
        out.append((base + ".bias", (cout,)))

    def deconv(name, cin, cout):
        out.append((name + ".weight", (cin, cout, 4, 4)))
        out.append((name + ".bias", (cout,)))

    chans = [3, 16, 32, 64, 96, 128, 196]
    names = [("conv1a", "conv1aa", "conv1b"), ("conv2a", "conv2aa", "conv2b"), ("conv3a", "conv3aa", "conv3b"),
             ("conv4a", "conv4aa", "conv4b"), ("conv5a", "conv5aa", "conv5b"), ("conv6aa", "conv6a", "conv6b")]
    for lvl, trio in enumerate(names, start=1):
        conv(trio[0], chans[lvl - 1], chans[lvl])
        conv(trio[1], chans[lvl], chans[lvl])
        conv(trio[2], chans[lvl], chans[lvl])
    nd = (2 * md + 1) ** 2
    dd = [128, 256, 352, 416, 448]
    outs = [128, 128, 96, 64, 32]
    for lvl in (6, 5, 4, 3, 2):
        od = nd if lvl == 6 else nd + chans[lvl] + 4
        for i in range(5):
            conv("conv%d_%d" % (lvl, i), od + (dd[i - 1] if i else 0), outs[i])
        conv("predict_flow%d" % lvl, od + dd[4], 2, seq=False)
        deconv("deconv%d" % lvl, 2, 2)
        if lvl > 2:
            deconv("upfeat%d" % lvl, od + dd[4], 2)
    od = nd + chans[2] + 4
    dc = [(od + dd[4], 128), (128, 128), (128, 128), (128, 96), (96, 64), (64, 32)]
    for i, (ci, co) in enumerate(dc, start=1):
        conv("dc_conv%d" % i, ci, co)
    conv("dc_conv7", 32, 2, seq=False)
    return out


# --------------------------------------------------------------------------
# .flo container
# --------------------------------------------------------------------------

def flo_bytes(uv: np.ndarray) -> bytes:
    """script_pwc.py:12-27: float32 tag 202021.25 | int32 W | int32 H | H*W*2 float32 row-major."""
    assert uv.ndim == 3 and uv.shape[2] == 2
    h, w = uv.shape[:2]
    return struct.pack("<f", FLO_TAG) + struct.pack("<i", w) + struct.pack("<i", h) + \
        np.ascontiguousarray(uv, dtype="<f4").tobytes()


def parse_flo(buf: bytes) -> np.ndarray:
    """data_processing.py:17-29 reader semantics: tag check, then W, H, then HxWx2 float32."""
    tag, = struct.unpack("<f", buf[:4])
    if tag != FLO_TAG:
        raise ValueError("bad .flo tag %r" % tag)
    w, h = struct.unpack("<ii", buf[4:12])
    return np.frombuffer(buf, dtype="<f4", count=2 * w * h, offset=12).reshape(h, w, 2).copy()


def epe(a: torch.Tensor, b: torch.Tensor) -> float:
    """mean over pixels of ||a-b||_2 on [B,2,H,W] flows (inference_kitti.py:94-105 without masking)."""
    return torch.sqrt(((a.double() - b.double()) ** 2).sum(dim=1)).mean().item()
