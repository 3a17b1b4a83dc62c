"""Half-precision execution plan of PWCDCNet.forward (BASELINE configs 3-4): same launch order as engine.PwcPlan, with

  * activations in the channel-blocked "c8" layout ``[B, ceil(C/8), H, W, 8]`` float16 (ops_f16), fp32 accumulation in
    every kernel, fp32 biases;
  * one arena per level in GROUPS of 8 channels
        [conv_4 (4) | conv_3 (8) | conv_2 (12) | conv_1 (16) | conv_0 (16) | corr (11 = 81 ch + 7 zeros) | c1 (C_L/8) | flow (1)]
    where the flow group carries up_flow in channels 0,1 and up_feat in channels 2,3 (4..7 zero).  Filters are
    re-indexed from the reference's concatenation order (models/PWCNet.py:215,229,245,259 and the new-features-first
    dense concat :202-206) to this physical order, with zero filters on the pad channels;
  * the 2-channel layers as MFMA convolutions: ``predict_flowL`` and ``upfeatL`` read the same 3x3 windows, and a
    ConvTranspose2d(k4, s2, p1) is a 3x3 convolution with 4 output phases per channel followed by a pixel shuffle
    (PWCNet.py:35-36), so head + upfeat are ONE convolution with 16 output channels (flow in group 0, the 8 upfeat
    phases in group 1);
  * the FLOW CHAIN IN FP32: the values that carry the flow from level to level (PWCNet.py:207-212,268) never pass
    through half.  The head convolutions (and dc_conv7) leave their result in float32 and use split filters
    (hi + residual in the idle half of the 32-row cout tile: ~22-bit filters at no extra MFMA cost); ``deconvL``
    (2 -> 2 channels) is applied in fp32 inside the level-entry kernel, which warps with that fp32 up_flow; the final
    ``flow2 + dc_conv7(...)`` is an fp32 add.  Only the COPY of (up_flow, up_feat) that the next level's convolutions
    read as 4 of their ~500 input channels is rounded to half.  Every store to half saturates at +-65504.

No autograd (like the fp32 plan); training mode returns the 5-tuple via ``flows()``.  ``variant="old"`` (PWCDCNet_old, PWCNet.py:277-491) first brings that model's filters into
PWCDCNet's concatenation order (engine.old_variant_perm), skips the ``*aa`` pyramid convs and uses the 0.999 mask
threshold.  Input/outputs stay float32 NCHW like the reference's interface.
"""
from __future__ import annotations

import os
from typing import Dict

import torch

from . import _lib, ops
from . import ops_f16 as F16
from .engine import (CONTEXT, DENSE_OUT, LEAKY, PYRAMID_CH, PYRAMID_NAMES, PYRAMID_NAMES_OLD, WARP_SCALE, level_in_channels,
                     old_variant_perm)

DENSE_G = (40, 24, 12, 4, 0)          # group offset of conv{L}_i's output (conv_0 .. conv_4) inside the arena
BASE_G = 56                           # first group after the dense-block outputs
CORR_G = 11                           # 81 channels + 7 zeros


def _groups(c: int) -> int:
    return (c + 7) // 8


def _pad_cin(w: torch.Tensor, phys: int) -> torch.Tensor:
    out = w.new_zeros((w.shape[0], phys, 3, 3))
    out[:, :w.shape[1]] = w
    return out


_deconv_as_conv3x3 = ops.deconv_as_conv3x3     # ConvTranspose2d(k4, s2, p1) as a 3x3 convolution with four output phases per channel


def _phys_index(level: int, nd: int = 81) -> torch.Tensor:
    """reference channel r of the level's full concatenation -> physical channel of the c8 arena."""
    c = PYRAMID_CH[level]
    idx = list(range(448)) + [448 + i for i in range(nd)]
    if level < 6:
        feat0 = 448 + CORR_G * 8
        idx += [feat0 + i for i in range(c)]
        idx += [feat0 + c + i for i in range(4)]               # up_flow 0,1 then up_feat 0,1 = channels 0..3 of the flow group
    return torch.tensor(idx, dtype=torch.long)


def prepare_params(params: Dict[str, torch.Tensor], variant: str, nd: int = 81) -> Dict[str, torch.Tensor]:
    """float32 copies of the parameters; PWCDCNet_old's filters re-ordered from its dense-block concatenation order to
    PWCDCNet's (engine.old_variant_perm), as engine.PwcPlan does."""
    p = {k: v.detach().float() for k, v in params.items()}
    if variant == "old":
        for l in range(2, 7):
            od = level_in_channels(l, nd)
            keys = [("conv%d_%d.0.weight" % (l, k), k, 1) for k in range(1, 5)] + [("predict_flow%d.weight" % l, 5, 1)]
            keys.append(("upfeat%d.weight" % l, 5, 0) if l > 2 else ("dc_conv1.0.weight", 5, 1))
            for key, k, dim in keys:
                p[key] = p[key].index_select(dim, old_variant_perm(k, od).to(p[key].device)).contiguous()
    return p


def level_filters(p: Dict[str, torch.Tensor], l: int, nphys: int, nd: int = 81):
    """Filters of decoder level l re-indexed from the reference's concatenation order to the physical channel order of the c8
    arena (`nphys` channels): yields (name, weight [Cout, Cin_phys, 3, 3], bias, is_flow_head).  `head%d` is predict_flowL
    (+ the eight pixel-shuffle phases of upfeatL for l > 2, see the module docstring); level 2 also yields dc_conv1, which
    reads the same arena."""
    full = _phys_index(l, nd)
    od = level_in_channels(l, nd)

    def remap(w, ref_start, phys_start):
        out = w.new_zeros((w.shape[0], nphys - phys_start, 3, 3))
        ref = torch.arange(ref_start, ref_start + w.shape[1])
        out[:, full[ref] - phys_start] = w
        return out

    ref_start = 448                                      # conv_0 reads corr.. ; each next conv one more dense output
    for i, co in enumerate(DENSE_OUT):
        yield "conv%d_%d" % (l, i), remap(p["conv%d_%d.0.weight" % (l, i)], ref_start, ref_start), p["conv%d_%d.0.bias" % (l, i)], False
        ref_start -= co
    assert ref_start == 0 and od + 448 == full.numel()
    wh, bh = p["predict_flow%d.weight" % l], p["predict_flow%d.bias" % l]
    if l > 2:
        wu = _deconv_as_conv3x3(p["upfeat%d.weight" % l])            # [8, Cin_ref, 3, 3]
        k = wh.new_zeros((16, wh.shape[1], 3, 3))
        k[0:2], k[8:16] = wh, wu
        bias = bh.new_zeros(16)
        bias[0:2] = bh
        bias[8:16] = p["upfeat%d.bias" % l].repeat_interleave(4)
        yield "head%d" % l, remap(k, 0, 0), bias, True
    else:
        yield "head2", remap(wh, 0, 0), bh, True
        yield "dc_conv1", remap(p["dc_conv1.0.weight"], 0, 0), p["dc_conv1.0.bias"], False


def context_filters(p: Dict[str, torch.Tensor]):
    """dc_conv2..7 (dc_conv1 comes from level_filters(2): it reads the level-2 arena), input channels padded to whole groups."""
    for i in range(2, 7):
        w = p["dc_conv%d.0.weight" % i]
        yield "dc_conv%d" % i, _pad_cin(w, _groups(w.shape[1]) * 8), p["dc_conv%d.0.bias" % i], False
    yield "dc_conv7", _pad_cin(p["dc_conv7.weight"], 32), p["dc_conv7.bias"], True


class PwcPlanF16:
    def __init__(self, params: Dict[str, torch.Tensor], B: int, H: int, W: int, device: torch.device, md: int = 4,
                 normalize_corr: bool = False, align_corners: bool = False, variant: str = "dc", fuse_pyramid1: bool = True):
        if variant not in ("dc", "old"):
            raise ValueError("variant must be 'dc' or 'old'")
        self.variant = variant
        self.pyramid_names = PYRAMID_NAMES if variant == "dc" else PYRAMID_NAMES_OLD
        self.mask_threshold = 0.9999 if variant == "dc" else 0.999
        if H % 64 or W % 64 or H <= 0 or W <= 0:
            raise ValueError("PWCDCNet needs H and W to be positive multiples of 64 (got %dx%d)" % (H, W))
        if md != 4:
            raise NotImplementedError("the fp16 correlation kernel is built for md=4")
        self.B, self.H, self.W, self.device = B, H, W, device
        self.normalize_corr, self.align_corners = normalize_corr, align_corners
        self.nd = 81
        hk = dict(device=device, dtype=torch.float16)
        self.size = {l: (H >> l, W >> l) for l in range(1, 7)}
        self.pyr_a, self.pyr_b = {}, {}
        # PWC_F16_FUSE_PYR1=0 keeps the layer-by-layer first level (A/B runs, tests)
        self.fuse_pyr1 = variant == "dc" and fuse_pyramid1 and os.environ.get("PWC_F16_FUSE_PYR1", "1") != "0"
        for l in range(1, 7):
            if l == 1 and self.fuse_pyr1:
                continue                                  # the level-1 maps live in LDS (ops_f16.pyramid1_fused)
            h, w = self.size[l]
            g = _groups(PYRAMID_CH[l])
            self.pyr_a[l] = torch.zeros((self._slots(B), g, h, w, 8), **hk)
            self.pyr_b[l] = torch.zeros((self._slots(B), g, h, w, 8), **hk)
        self.arena, self.warped, self.head = {}, {}, {}
        self.fused_entry = _lib.get_option("f16_level_corr") > 0      # level entry + warp + cost volume as one kernel (warped: LDS only)
        for l in range(2, 7):
            h, w = self.size[l]
            g = _groups(PYRAMID_CH[l])
            self.arena[l] = torch.zeros((B, BASE_G + CORR_G + (g + 1 if l < 6 else 0), h, w, 8), **hk)
            if l < 6:
                self.warped[l] = torch.zeros((B, g, h, w, 8), **hk)
            self.head[l] = torch.zeros((B, 2 if l > 2 else 1, h, w, 8), device=device, dtype=torch.float32)
        h2, w2 = self.size[2]
        self.ctx = [torch.zeros((B, _groups(c), h2, w2, 8), **hk) for c, _ in CONTEXT]
        self.dc7 = torch.zeros((B, 1, h2, w2, 8), device=device, dtype=torch.float32)
        self.flow_out = torch.empty((B, 2, h2, w2), device=device, dtype=torch.float32)

        # ---- filters: re-index to the physical channel order, pad, pack --------------------------------------------
        self.w: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}
        self.cin: Dict[str, int] = {}
        self.cout: Dict[str, int] = {}
        self.deconv_w: Dict[int, torch.Tensor] = {}
        self.deconv_b: Dict[int, torch.Tensor] = {}

        self.split = set()

        def put(name, w, bias, split=False):
            self.w[name] = F16.pack_conv3x3_f16(w.contiguous().float(), split=split)
            if split:
                self.split.add(name)
            self.b[name] = bias.contiguous().float()
            self.cin[name], self.cout[name] = w.shape[1], w.shape[0]

        p = prepare_params(params, variant, self.nd)
        # conv1a (3 -> 16, stride 2) runs straight from the float32 image (ops_f16.image_conv_s2): keep its raw filters
        self.w1a = p["conv1a.0.weight"].contiguous()
        self.b1a = p["conv1a.0.bias"].contiguous()
        # PWCDCNet: conv1a -> conv1aa -> conv1b -> conv2a as one kernel, the level-1 maps never leave the CU
        # (ops_f16.pyramid1_fused; PWCDCNet_old has no conv1aa and keeps the layer-by-layer path)
        self.pyr1 = None
        if self.fuse_pyr1:
            self.pyr1 = F16.pack_pyramid1(self.w1a, self.b1a, p["conv1aa.0.weight"].contiguous(), p["conv1aa.0.bias"],
                                          p["conv1b.0.weight"].contiguous(), p["conv1b.0.bias"],
                                          p["conv2a.0.weight"].contiguous(), p["conv2a.0.bias"])
        for l, names in enumerate(self.pyramid_names, start=1):
            for i, n in enumerate(names):
                if n is None or (l == 1 and i == 0):
                    continue
                w = p[n + ".0.weight"]
                put(n, _pad_cin(w, _groups(w.shape[1]) * 8), p[n + ".0.bias"])
        for l in range(2, 7):
            for name, w, bias, is_head in level_filters(p, l, int(self.arena[l].shape[1]) * 8, self.nd):
                put(name, w, bias, split=is_head)
            if l > 2:
                # deconvL (2 -> 2 channels) runs in fp32 inside the level-entry kernel: raw ConvTranspose2d parameters
                self.deconv_w[l] = p["deconv%d.weight" % l].contiguous()
                self.deconv_b[l] = p["deconv%d.bias" % l].contiguous()
        for name, w, bias, is_head in context_filters(p):
            put(name, w, bias, split=is_head)

    @staticmethod
    def _slots(B: int) -> int:
        return 2 * B

    # ---- primitives -------------------------------------------------------------------------------------------------
    def _conv(self, name, x, out, stride=1, dilation=1, act=True):
        F16.conv3x3_f16(x, self.w[name], self.b[name], self.cin[name], self.cout[name], stride=stride, dilation=dilation,
                        leaky_slope=LEAKY if act else None, out=out, out_f32=out.dtype == torch.float32,
                        split_w=name in self.split)

    # ---- the forward ----------------------------------------------------------------------------------------------------
    def run(self, x: torch.Tensor) -> torch.Tensor:
        B = self.B
        if tuple(x.shape) != (B, 6, self.H, self.W) or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("plan built for float32 %s on %s, got %s %s on %s" % (
                (B, 6, self.H, self.W), self.device, x.dtype, tuple(x.shape), x.device))
        x = ops.densify(x)
        self._pyramid([(x[:, :3], 0, B), (x[:, 3:], B, 2 * B)], 0, 2 * B)
        return self._decode()

    def flows(self):
        """(flow2, flow3, flow4, flow5, flow6) of the last run as float32 NCHW -- the training-mode return (PWCNet.py:270-271)."""
        return (self.flow_out,) + tuple(self.head[l][:, 0, :, :, 0:2].permute(0, 3, 1, 2).contiguous() for l in (3, 4, 5, 6))

    def _pair_views(self, l: int):
        """first / second image's level features as views of the pyramid buffer"""
        return self.pyr_a[l][:self.B], self.pyr_a[l][self.B:]

    def _pyramid(self, images, lo: int, hi: int) -> None:
        """conv1a..conv6b over the batch slots [lo,hi); `images` lists (rgb float32 [n,3,H,W], slot_lo, slot_hi)."""
        prev = None
        for l in range(1, 7):
            na, naa, nb = self.pyramid_names[l - 1]
            if l == 1 and self.pyr1 is not None:
                continue                                                   # level 1 + conv2a are one launch below
            a, bb = self.pyr_a[l][lo:hi], self.pyr_b[l][lo:hi]
            first = self.pyr_a[l] if naa is not None else self.pyr_b[l]   # three convs a -> bb -> a; PWCDCNet_old: bb -> a
            if l == 1:
                for img, s0, s1 in images:
                    F16.image_conv_s2(img, self.w1a, self.b1a, LEAKY, out=first[s0:s1])
            elif l == 2 and self.pyr1 is not None:
                for img, s0, s1 in images:
                    F16.pyramid1_fused(img, self.pyr1[0], self.pyr1[1], LEAKY, out=first[s0:s1])
            else:
                self._conv(na, prev, first[lo:hi], stride=2)
            if naa is not None:
                self._conv(naa, a, bb)
            self._conv(nb, bb, a)
            prev = a

    def _decode(self) -> torch.Tensor:
        B = self.B
        for l in (6, 5, 4, 3, 2):
            ar = self.arena[l]
            g = _groups(PYRAMID_CH[l])
            c1, c2 = self._pair_views(l)
            corr_slot = ar[:, BASE_G:BASE_G + CORR_G]
            if l == 6:
                F16.correlation_c8(c1, c2, PYRAMID_CH[6], normalize=self.normalize_corr, leaky_slope=LEAKY, out=corr_slot)
            else:
                # pixel-shuffle deconv / upfeat of the level above into the flow group, c1 into the arena, warp; then the cost volume.
                # Option f16_level_corr = 1: all of it as ONE launch with the warped features in LDS only -- same bits, but slower at
                # batch 16 (each tile gathers its whole halo), so it is opt-in
                f0 = BASE_G + CORR_G
                kw = dict(c1_dst=ar[:, f0:f0 + g], flow_group=ar[:, f0 + g:f0 + g + 1], flow_scale=WARP_SCALE[l],
                          align_corners=self.align_corners, mask_threshold=self.mask_threshold)
                args = (c1, c2, self.head[l + 1][:, 0:1], self.head[l + 1][:, 1:2], self.deconv_w[l + 1], self.deconv_b[l + 1], PYRAMID_CH[l])
                if self.fused_entry:
                    F16.level_entry_correlation(*args, out=corr_slot, normalize=self.normalize_corr, leaky_slope=LEAKY, **kw)
                else:
                    F16.level_entry(*args, out=self.warped[l], **kw)
                    F16.correlation_c8(c1, self.warped[l], PYRAMID_CH[l], normalize=self.normalize_corr, leaky_slope=LEAKY,
                                       out=corr_slot)
            lo = BASE_G
            for i, og in enumerate(DENSE_G):
                self._conv("conv%d_%d" % (l, i), ar[:, lo:], ar[:, og:og + DENSE_OUT[i] // 8])
                lo = og
            self._conv("head%d" % l, ar, self.head[l], act=False)     # float32 out: flow (+ the 8 upfeat phases)
        t = self.arena[2]
        for i, (_, dil) in enumerate(CONTEXT):
            self._conv("dc_conv%d" % (i + 1), t, self.ctx[i], dilation=dil)
            t = self.ctx[i]
        self._conv("dc_conv7", t, self.dc7, act=False)
        # flow2 = predict_flow2 + dc_conv7 (PWCNet.py:268), summed in fp32, channels 0,1 of the one-group tensors
        h2, w2 = self.size[2]
        torch.add(self.head[2][:, 0, :, :, 0:2].permute(0, 3, 1, 2), self.dc7[:, 0, :, :, 0:2].permute(0, 3, 1, 2),
                  out=self.flow_out)
        return self.flow_out


class PwcVideoPlanF16(PwcPlanF16):
    """Half-precision plan for consecutive frame pairs of one video (pwc_extract_flow_video.py:262-305): B+1 batch
    slots per pyramid buffer, slot 0 carries the last frame of the previous step, the pair views overlap ([0:B] / [1:B+1]),
    one pyramid pass per frame -- the fp16 twin of engine.PwcVideoPlan."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.primed = False

    @staticmethod
    def _slots(B: int) -> int:
        return B + 1

    def _pair_views(self, l: int):
        return self.pyr_a[l][:self.B], self.pyr_a[l][1:]

    def _check(self, frames: torch.Tensor, n: int) -> torch.Tensor:
        if tuple(frames.shape) != (n, 3, self.H, self.W) or frames.dtype != torch.float32 or frames.device != self.device:
            raise ValueError("expected float32 frames %s on %s, got %s %s on %s" % (
                (n, 3, self.H, self.W), self.device, frames.dtype, tuple(frames.shape), frames.device))
        return ops.densify(frames)

    def _carry(self) -> None:
        for l in range(2, 7):
            self.pyr_a[l][0].copy_(self.pyr_a[l][self.B])

    def prime(self, frame: torch.Tensor) -> None:
        f = self._check(frame, 1)
        self._pyramid([(f, self.B, self.B + 1)], self.B, self.B + 1)
        self._carry()
        self.primed = True

    def push(self, frames: torch.Tensor) -> torch.Tensor:
        if not self.primed:
            raise RuntimeError("PwcVideoPlanF16.push before prime(first_frame)")
        f = self._check(frames, self.B)
        self._pyramid([(f, 1, self.B + 1)], 1, self.B + 1)
        out = self._decode()
        self._carry()
        return out

    def run(self, x):
        raise RuntimeError("PwcVideoPlanF16 is driven by prime()/push(), not run()")
