"""Execution plan of PWCDCNet.forward on one MI355X: buffers, packed weights, launch order.

What the reference does per forward (models/PWCNet.py:180-273) and what the plan does instead:

  * ``torch.cat((conv(x), x), 1)`` five times per level (PWCNet.py:202-264, 747 MB of copies per
    1024x448 pair): the plan allocates ONE arena ``[B, Ctot, H, W]`` per level whose channel order
    is the final concatenation order
        [conv_4 | conv_3 | conv_2 | conv_1 | conv_0 | corr | c1 | up_flow | up_feat]
    and every producer (dense convs, correlation, the two deconvs) writes its channel slice in
    place (the first image's pyramid features are COPIED into their slot, 46 us per forward at
    batch 16: the pyramid runs both images as one 2B batch with one output stride); a consumer reads a channel *suffix* -- only the batch stride differs
    from a dense tensor, which the C ABI takes as an argument.
  * warp: one fused kernel instead of mesh + 2x grid_sample + mask ops (PWCNet.py:141-177), with the
    per-level flow scale (PWCNet.py:212,226,240,256) folded in.
  * correlation + LeakyReLU (PWCNet.py:198-199): one kernel writing straight into the arena.
  * both images go through the feature pyramid as one 2B batch (PWCNet.py:184-195 runs it twice).

All launches go to the current torch stream; nothing allocates after construction, so a plan can be
captured into a HIP graph (``PWCDCNet.forward(..., )`` does that when ``use_graph`` is set).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib, ops

PYRAMID_CH = (3, 16, 32, 64, 96, 128, 196)
PYRAMID_NAMES = (("conv1a", "conv1aa", "conv1b"), ("conv2a", "conv2aa", "conv2b"), ("conv3a", "conv3aa", "conv3b"),
                 ("conv4a", "conv4aa", "conv4b"), ("conv5a", "conv5aa", "conv5b"), ("conv6aa", "conv6a", "conv6b"))
PYRAMID_NAMES_OLD = (("conv1a", None, "conv1b"), ("conv2a", None, "conv2b"), ("conv3a", None, "conv3b"),
                     ("conv4a", None, "conv4b"), ("conv5a", None, "conv5b"), ("conv6a", None, "conv6b"))   # PWCNet.py:290-301
DENSE_OUT = (128, 128, 96, 64, 32)            # conv{L}_0 .. conv{L}_4 (PWCNet.py:78-82)
DENSE_TOTAL = sum(DENSE_OUT)                  # 448
# channel offset of conv{L}_i's output inside the arena; conv{L}_i reads everything after it
DENSE_OFF = (320, 192, 96, 32, 0)
CONTEXT = ((128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1))   # dc_conv1..6 (PWCNet.py:126-131)
WARP_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}                      # PWCNet.py:212,226,240,256
LEAKY = 0.1


def old_variant_perm(k: int, od: int) -> torch.Tensor:
    """Input-channel permutation for PWCDCNet_old (PWCNet.py:425-429): its dense block concatenates in the order
    [conv_1 | base | conv_0 | conv_2 | conv_3 | conv_4] while the arena is [conv_4 | conv_3 | conv_2 | conv_1 | conv_0 | base].
    Returns `perm` with arena_input_channel j == reference_input_channel perm[j] for the consumer that reads the
    outputs of conv_0..conv_{k-1} (k = 5: flow head / upfeat / dc_conv1).  A convolution over a concatenation does
    not care about the order as long as the filter's input channels follow it."""
    sizes = {"base": od}
    for i, c in enumerate(DENSE_OUT):
        sizes["c%d" % i] = c
    logical = [n for n in ("c1", "base", "c0", "c2", "c3", "c4") if n == "base" or int(n[1]) < k]
    physical = ["c%d" % i for i in range(k - 1, -1, -1)] + ["base"]
    start, pos = {}, 0
    for n in logical:
        start[n] = pos
        pos += sizes[n]
    return torch.cat([torch.arange(start[n], start[n] + sizes[n]) for n in physical])


def level_in_channels(level: int, nd: int = 81) -> int:
    """`od` of PWCNet.py:77,87,97,107,117."""
    return nd if level == 6 else nd + PYRAMID_CH[level] + 4


class PwcPlan:
    def __init__(self, params: Dict[str, torch.Tensor], B: int, H: int, W: int, device: torch.device,
                 dtype: torch.dtype = torch.float32, md: int = 4, normalize_corr: bool = False,
                 align_corners: bool = False, conv_backend: str = "hip", variant: str = "dc", trunk2: bool = True):
        """trunk2=False (engine_strict.PwcPlanStrict): this plan stops after the level-2 ENTRY (correlation, c1, up_flow, up_feat);
        the level-2 dense block, flow head and context network belong to the caller, so their buffers and packed filters are not
        allocated and the level-2 arena holds only its 117 base channels."""
        if variant not in ("dc", "old"):
            raise ValueError("variant must be 'dc' (PWCDCNet) or 'old' (PWCDCNet_old)")
        if H % 64 or W % 64 or H <= 0 or W <= 0:
            raise ValueError("PWCDCNet needs H and W to be positive multiples of 64 (got %dx%d); resize or pad the "
                             "pair first like script_pwc.py:47-54 / inference_kitti.py:53-63" % (H, W))
        self.trunk2 = trunk2
        if conv_backend not in ("hip", "torch"):
            raise ValueError("conv_backend must be 'hip' or 'torch'")
        if dtype != torch.float32:
            raise NotImplementedError("plan dtype %s: only float32 is wired up" % dtype)
        self.B, self.H, self.W = B, H, W
        self.device, self.dtype = device, dtype
        self.md = md
        self.nd = (2 * md + 1) ** 2
        self.normalize_corr = normalize_corr
        self.align_corners = align_corners
        self.conv_backend = conv_backend
        self.variant = variant
        self.fuse_warp = os.environ.get("PWC_FUSE_WARP_CORR", "1") != "0"     # 0: warp and correlation as two launches (A/B runs)
        self.pyramid_names = PYRAMID_NAMES if variant == "dc" else PYRAMID_NAMES_OLD
        self.mask_threshold = 0.9999 if variant == "dc" else 0.999        # PWCNet.py:174 / :400
        self.p = dict(params)
        if variant == "old":
            # re-order the filters' input channels from the reference's concatenation order to the arena's
            for l in range(2, 7):
                od = level_in_channels(l, self.nd)
                names = [("conv%d_%d.0.weight" % (l, k), k, 1) for k in range(1, 5)]
                names.append(("predict_flow%d.weight" % l, 5, 1))
                if l > 2:
                    names.append(("upfeat%d.weight" % l, 5, 0))
                else:
                    names.append(("dc_conv1.0.weight", 5, 1))
                for key, k, dim in names:
                    perm = old_variant_perm(k, od).to(device)
                    self.p[key] = self.p[key].index_select(dim, perm).contiguous()
        kw = dict(device=device, dtype=dtype)

        self.size = {l: (H >> l, W >> l) for l in range(1, 7)}
        # pyramid scratch: per level two ping-pong buffers for the 2B batch, plus c2 (second image)
        self.pyr_a, self.pyr_b, self.c1, self.c2 = {}, {}, {}, {}
        for l in range(1, 7):
            h, w = self.size[l]
            c = PYRAMID_CH[l]
            self.pyr_a[l] = torch.empty((self._slots(B), c, h, w), **kw)      # levels 2-5: replaced by a view of the arena below
            self.pyr_b[l] = torch.empty((self._slots(B), c, h, w), **kw)
        self.warped = {l: torch.empty((B, PYRAMID_CH[l], *self.size[l]), **kw) for l in range(2, 6)}
        self.arena = {}
        self.arena_base = {}                    # first channel after the dense-block outputs
        # Levels 2-5: the first image's level features are part of the dense block's input (PWCNet.py:215), i.e. they
        # have a slot in the level's arena.  Instead of copying them there, the level's pyramid buffer IS that slot: the arena is
        # allocated for 2B items, items [0,B) are the decoder's arena, items [B,2B) only ever hold the second image's features in the
        # same channel slot, and pyr_a[l] is the [2B, c, h, w] view of that slot at the arena's batch stride -- every kernel takes batch
        # strides, so the pyramid's last convolution writes c1 where the decoder reads it (four copies, 47 us at batch 16, gone; the
        # cost is address space: 2.1 GB instead of 1.0 GB at level 2 / batch 16).  Not for the video plan (its c1 / c2 overlap).
        self.c1_in_arena = bool(conv_backend == "hip" and self._slots(B) == 2 * B and _lib.get_option("c1_in_arena"))
        for l in range(2, 7):
            od = level_in_channels(l, self.nd)
            self.arena_base[l] = DENSE_TOTAL if (l > 2 or trunk2) else 0
            if self.c1_in_arena and l < 6:
                pair = torch.empty((2 * B, self.arena_base[l] + od, *self.size[l]), **kw)
                self.arena[l] = pair[:B]
                off = self.arena_base[l] + self.nd
                self.pyr_a[l] = pair[:, off:off + PYRAMID_CH[l]]
            else:
                self.arena[l] = torch.empty((B, self.arena_base[l] + od, *self.size[l]), **kw)
        for l in range(2, 7):
            self.c1[l], self.c2[l] = self._pair_views(self.pyr_a[l], B)
        self.flow = {l: torch.empty((B, 2, *self.size[l]), **kw) for l in range(2 if trunk2 else 3, 7)}
        # Levels too small for the streaming head + upfeat kernel (W < 64: levels 6-5; all of 6-3 for a single pair): predict_flowL and
        # upfeatL run as ONE 3x3 convolution with 10 output channels on the matrix cores (ConvTranspose2d(k4,s2,p1) = a 3x3 convolution
        # with four output phases per channel) and ops.upsample_entry finishes the level -- the VALU deconvolution kernel took 21-26 us
        # per launch there.  flow[l] is then channels 0,1 of that convolution's output.
        self.head10: Dict[int, torch.Tensor] = {}
        # predict_flowL + upfeatL as one streaming pass: where its one-pass kernel runs (64 tiles) and, on Cin slices through the plan's
        # workspace, down to "head_sliced_min_tiles" tiles (level 3 of 2..8 pairs: 20-36 us ahead of the 10-channel convolution)
        self.stream_head = {l: conv_backend == "hip" and ops.head_upfeat_supported(
            B, *self.size[l], min_tiles=_lib.get_option("head_sliced_min_tiles") if _lib.get_option("stream_slice_wgs") > 0 else 64)
            for l in range(3, 7)}
        if conv_backend == "hip" and _lib.get_option("head10"):
            for l in range(3, 7):
                if not self.stream_head[l]:
                    self.head10[l] = torch.empty((B, 10, *self.size[l]), **kw)
                    self.flow[l] = self.head10[l][:, 0:2]
        h2, w2 = self.size[2]
        self.flow_out = torch.empty((B, 2, h2, w2), **kw) if trunk2 else None
        self.ctx = [torch.empty((B, c, h2, w2), **kw) for c, _ in CONTEXT] if trunk2 else []
        # Context network in LATTICE-MAJOR layout (round 3): dc_conv1..3 store their result as its four pixel lattices (PWC_CONV_SPLIT2),
        # so the next layer -- dilation 2, 4, 8 in the reference (PWCNet.py:126-131) -- is a dilation-1 convolution on 4x as many
        # images of half the size, with contiguous rows: F(4x4) / F(2x2) at their dilation-1 speed instead of strided lattice
        # addressing (dc_conv2 611 -> ~430 us, dc_conv3 679 -> ~470 us at batch 16).  dc_conv5 (dilation 16) is a dilation-2
        # convolution on the dilation-8 lattices; its 64-channel result is brought back to NCHW by pwc_lattice_unsplit_f32.
        wino_on = (os.environ.get("PWC_CONV_WINO", "1") != "0" and os.environ.get("PWC_CONV_WINO4", "1") != "0"
                   and (conv_backend != "hip" or _lib.get_option("conv_wino4") != 0))
        self.ctx_lattice = bool(
            trunk2 and conv_backend == "hip" and dtype == torch.float32 and wino_on and os.environ.get("PWC_CTX_LATTICE", "1") != "0"
            and h2 % 8 == 0 and w2 % 32 == 0
            and ops.conv3x3_wino4_preferred(B, level_in_channels(2, self.nd) + DENSE_TOTAL, h2, w2, CONTEXT[0][0])
            and ops.conv3x3_wino4_preferred(4 * B, CONTEXT[0][0], h2 // 2, w2 // 2, CONTEXT[1][0])
            and ops.conv3x3_wino4_preferred(16 * B, CONTEXT[1][0], h2 // 4, w2 // 4, CONTEXT[2][0]))
        if self.ctx_lattice:
            self.ctx[0] = torch.empty((4 * B, CONTEXT[0][0], h2 // 2, w2 // 2), **kw)
            self.ctx[1] = torch.empty((16 * B, CONTEXT[1][0], h2 // 4, w2 // 4), **kw)
            self.ctx[2] = torch.empty((64 * B, CONTEXT[2][0], h2 // 8, w2 // 8), **kw)
            self.ctx[3] = torch.empty((64 * B, CONTEXT[3][0], h2 // 8, w2 // 8), **kw)
            self.ctx4_lat = torch.empty((64 * B, CONTEXT[4][0], h2 // 8, w2 // 8), **kw)

        self.packed: Dict[str, torch.Tensor] = {}
        self.wino_packed: Dict[str, torch.Tensor] = {}
        self.wino4_packed: Dict[str, torch.Tensor] = {}        # F(4x4,3x3) banks, packed on first use by the layers the rule picks
        self.split96_bias: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.conv_macs = {"direct": 0, "executed": 0}
        self.wino = os.environ.get("PWC_CONV_WINO", "1") != "0" and dtype == torch.float32
        # 0 (environment, or pwc_set_option("conv_wino4", 0) before the plan is built): large layers stay on F(2x2,3x3) (A/B runs, error budget)
        self.wino4 = os.environ.get("PWC_CONV_WINO4", "1") != "0" and (conv_backend != "hip" or _lib.get_option("conv_wino4") != 0)
        self.workspace: Optional[torch.Tensor] = None
        if conv_backend == "hip":
            for key, t in self.p.items():
                if not trunk2 and key.startswith(("conv2_", "predict_flow2", "dc_conv")):
                    continue                                   # run by the caller (engine_strict) in its own format
                if key.endswith(".weight") and t.dim() == 4 and t.shape[2:] == (3, 3):
                    self.packed[key[:-len(".weight")]] = ops.pack_conv3x3(t)
                    # G g Gt for every layer the Winograd route could take (67 MB for the whole net): nothing is allocated later
                    if self.wino and t.shape[0] >= 32 and t.shape[1] >= 16:
                        self.wino_packed[key[:-len(".weight")]] = ops.pack_conv3x3_wino(t)
            for l in self.head10:
                w10 = torch.cat((self.p["predict_flow%d.weight" % l], ops.deconv_as_conv3x3(self.p["upfeat%d.weight" % l])), 0)
                self.p["head10_%d.weight" % l] = w10.contiguous()
                self.p["head10_%d.bias" % l] = torch.cat((self.p["predict_flow%d.bias" % l],
                                                           self.p["upfeat%d.bias" % l].repeat_interleave(4))).contiguous()
                self.packed["head10_%d" % l] = ops.pack_conv3x3(self.p["head10_%d.weight" % l])
            # F(4x4,3x3) banks for the layers the measured rule picks at THIS geometry (4x the filter bytes each): packed now, so that a
            # run never allocates (the lazy path in _conv only serves callers that drive _conv with other shapes, e.g. bench probes)
            if self.wino and self.wino4:
                geo = []
                for l in range(2, 7):
                    for n in self.pyramid_names[l - 1][1:]:
                        if n is not None:
                            geo.append((n + ".0", self._slots(B), PYRAMID_CH[l], PYRAMID_CH[l], l))
                for l in range(2 if trunk2 else 3, 7):
                    cin = level_in_channels(l, self.nd)
                    for i, co in enumerate(DENSE_OUT):
                        geo.append(("conv%d_%d.0" % (l, i), B, cin, co, l))
                        cin += co
                if trunk2:
                    cin = level_in_channels(2, self.nd) + DENSE_TOTAL
                    for i, (co, dil) in enumerate(CONTEXT):
                        if dil == 1:
                            geo.append(("dc_conv%d.0" % (i + 1), B, cin, co, 2))
                        cin = co
                    if self.ctx_lattice:                      # dc_conv2 / dc_conv3 run as dilation-1 layers on the lattices
                        for key in ("dc_conv2.0", "dc_conv3.0"):
                            self.wino4_packed[key] = ops.pack_conv3x3_wino4(self.p[key + ".weight"])
                        # dc_conv4 (128 -> 96 on 64B images of h/8 x w/8, e.g. 14x32): its first 64 couts fill F(4x4)'s 32-column tile
                        # groups, a 32-cout F(4x4) launch (32-row workgroups) would be half empty -> those couts stay on F(2x2)
                        w4 = self.p["dc_conv4.0.weight"]
                        self.dc4_split = bool(w4.shape[0] == 96 and ops.conv3x3_wino4_preferred(64 * B, w4.shape[1], h2 // 8, w2 // 8, 64))
                        if self.dc4_split:
                            self._pack_split96("dc_conv4.0")
                for key, b_, cin, co, l in geo:
                    h, w = self.size[l]
                    if key in self.wino_packed and ops.conv3x3_wino4_preferred(b_, cin, h, w, co):
                        self.wino4_packed[key] = ops.pack_conv3x3_wino4(self.p[key + ".weight"])
                    elif key in self.wino_packed and self._split96_wanted(b_, cin, h, w, co):
                        self._pack_split96(key)
            # ONE scratch shared by every convolution of the plan (they run back to back on one stream): split-K of the direct and
            # F(2x2) kernels, tail / whole-launch split of F(4x4).  pwc_conv3x3_wino*_preferred count those splits, so the buffer has to
            # cover EVERY launch the plan can make -- decoder, pyramid (2B images), context layers in NCHW and in the lattice-major
            # layout (the lattice geometries and the pyramid were missing; the C side then quietly ran unsplit).
            self.workspace_need = self._workspace_bytes(B, trunk2)
            need = self.workspace_need
            if need:
                self.workspace = torch.empty((need // 4,), **kw)

    def _launch_geometries(self, B: int, trunk2: bool):
        """(images, cin, h, w, cout, dilation) of every 3x3 stride-1 convolution this plan can launch"""
        out = []
        for l in range(1, 7):                                  # pyramid: the stride-1 layers of each level on both images
            h, w = self.size[l]
            for n in self.pyramid_names[l - 1][1:]:
                if n is not None:
                    out.append((self._slots(B), PYRAMID_CH[l], h, w, PYRAMID_CH[l], 1))
        for l in range(2 if trunk2 else 3, 7):                 # dense blocks + flow heads
            h, w = self.size[l]
            cin = level_in_channels(l, self.nd)
            for co in DENSE_OUT + (2, 10):                     # 10: flow head + upfeat phases as one convolution (small levels)
                out.append((B, cin, h, w, co, 1))
                out.append((B, cin, h, w, min(co, 64), 1))     # the F(4x4) part of a 96-cout layer run as 64 + 32
                cin += co if co not in (2, 10) else 0
        if trunk2:
            h, w = self.size[2]
            cin = level_in_channels(2, self.nd) + DENSE_TOTAL
            for co, dil in CONTEXT:
                out.append((B, cin, h, w, co, dil))
                cin = co
            if getattr(self, "ctx_lattice", False):            # the same layers on their pixel lattices (see __init__)
                c = [co for co, _ in CONTEXT]
                out += [(4 * B, c[0], h // 2, w // 2, c[1], 1), (16 * B, c[1], h // 4, w // 4, c[2], 1),
                        (64 * B, c[2], h // 8, w // 8, c[3], 1), (64 * B, c[2], h // 8, w // 8, 64, 1), (64 * B, c[2], h // 8, w // 8, 32, 1),
                        (64 * B, c[3], h // 8, w // 8, c[4], 2)]
        return out

    def _workspace_bytes(self, B: int, trunk2: bool) -> int:
        need = 0
        for n, cin, h, w, co, dil in self._launch_geometries(B, trunk2):
            need = max(need, ops.conv3x3_workspace_bytes(n, cin, h, w, co, 1, dil))
            if co == 2 and dil == 1:                            # predict_flowL + upfeatL as one streaming pass, cut along Cin when small
                need = max(need, ops.head_upfeat_workspace_bytes(n, cin, h, w))
            if self.wino and co >= 32 and dil <= 8:
                need = max(need, ops.conv3x3_wino_workspace_bytes(n, cin, h, w, co, dil))
            if self.wino and self.wino4 and co >= 32 and dil == 1 and w % 4 == 0:
                need = max(need, ops.conv3x3_wino4_workspace_bytes(n, cin, h, w, co))
        return need

    @staticmethod
    def _slots(B: int) -> int:
        return 2 * B

    @staticmethod
    def _pair_views(buf: torch.Tensor, B: int):
        """first / second image's level features as dense views of the pyramid buffer"""
        return buf[:B], buf[B:]

    # ---- 96-cout layers as 64 couts on F(4x4) + 32 couts on F(2x2) -----------------------------------------------------
    # A 32-cout F(4x4) launch is four tile groups per workgroup: on a map that gives the 64-cout launch (two groups) enough workgroups it
    # leaves half the chip idle (conv3_2 at batch 16: 128 workgroups; dc_conv4 on its 14x32 lattice images), so the rule says no to the
    # whole layer.  Its first 64 couts still fill F(4x4); the last 32 go to F(2x2), whose 32-cout workgroups are smaller.
    @staticmethod
    def _split96_wanted(b: int, cin: int, h: int, w: int, cout: int) -> bool:
        return bool(cout == 96 and not ops.conv3x3_wino4_preferred(b, cin, h, w, 96) and ops.conv3x3_wino4_preferred(b, cin, h, w, 64)
                    and ops.conv3x3_wino_preferred(b, cin, h, w, 32, 1))

    def _pack_split96(self, key: str) -> None:
        w, b = self.p[key + ".weight"], self.p[key + ".bias"]
        self.wino4_packed[key + "[:64]"] = ops.pack_conv3x3_wino4(w[:64].contiguous())
        self.wino_packed[key + "[64:]"] = ops.pack_conv3x3_wino(w[64:].contiguous())
        self.split96_bias[key] = (b[:64].contiguous(), b[64:].contiguous())

    def _conv_split96(self, key: str, x: torch.Tensor, out: torch.Tensor, act: bool = True) -> None:
        macs = x.shape[0] * 96 * x.shape[1] * 9 * x.shape[2] * x.shape[3]
        self.conv_macs["direct"] += macs
        self.conv_macs["executed"] += macs * 2 // 3 * 36 // 144 + macs // 3 * 16 // 36
        b0, b1 = self.split96_bias[key]
        slope = LEAKY if act else None
        ops.conv3x3_wino4(x, self.wino4_packed[key + "[:64]"], b0, 64, leaky_slope=slope, out=out[:, :64], workspace=self.workspace)
        ops.conv3x3_wino(x, self.wino_packed[key + "[64:]"], b1, 32, leaky_slope=slope, out=out[:, 64:], workspace=self.workspace)

    # ---- layer primitives -----------------------------------------------------------------------
    def _conv(self, name: str, x: torch.Tensor, out: torch.Tensor, stride: int = 1, dilation: int = 1,
              act: bool = True, residual: Optional[torch.Tensor] = None) -> None:
        key = name + ".0" if (name + ".0.weight") in self.p else name
        w, b = self.p[key + ".weight"], self.p[key + ".bias"]
        if (key + "[:64]") in self.wino4_packed and stride == 1 and dilation == 1 and residual is None and self._split96_wanted(
                x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0]):
            self._conv_split96(key, x, out, act)
            return
        macs = x.shape[0] * w.shape[0] * w.shape[1] * 9 * ((x.shape[2] - 1) // stride + 1) * ((x.shape[3] - 1) // stride + 1)
        self.conv_macs["direct"] += macs                          # what the layer costs as a direct convolution
        if (self.wino and self.wino4 and key in self.wino_packed and self.conv_backend == "hip" and stride == 1 and dilation == 1 and residual is None
                and x.dtype == torch.float32 and ops.conv3x3_wino4_preferred(x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0])):
            # Winograd F(4x4,3x3): 4x fewer multiplications than the direct form for the large, well-filled layers (gated: DESIGN.md 4b)
            if key not in self.wino4_packed:
                self.wino4_packed[key] = ops.pack_conv3x3_wino4(w)      # first (eager, warm-up) run; a captured replay finds it packed
            ops.conv3x3_wino4(x, self.wino4_packed[key], b, w.shape[0], leaky_slope=LEAKY if act else None, out=out, workspace=self.workspace)
            self.conv_macs["executed"] += macs * 36 // 144
            return
        if (self.wino and key in self.wino_packed and self.conv_backend == "hip" and stride == 1 and residual is None and x.dtype == torch.float32
                and ops.conv3x3_wino_preferred(x.shape[0], x.shape[1], x.shape[2], x.shape[3], w.shape[0], dilation)):
            # Winograd F(2x2,3x3) on the matrix cores: 2.25x fewer MFMA passes for the same fp32 convolution
            ops.conv3x3_wino(x, self.wino_packed[key], b, w.shape[0], leaky_slope=LEAKY if act else None, out=out, dilation=dilation,
                             workspace=self.workspace)
            self.conv_macs["executed"] += macs * 16 // 36         # multiplications the matrix cores actually perform
            return
        self.conv_macs["executed"] += macs
        if self.conv_backend == "hip":
            ops.conv3x3(x, self.packed[key], b, w.shape[0], stride=stride, dilation=dilation,
                        leaky_slope=LEAKY if act else None, residual=residual, out=out, workspace=self.workspace)
        else:
            # BASELINE config[1]: convolutions by PyTorch-ROCm (MIOpen); correlation/warp stay HIP.
            y = F.conv2d(x, w, b, stride=stride, padding=dilation, dilation=dilation)
            if act:
                y = F.leaky_relu(y, LEAKY)
            if residual is not None:
                y = y + residual
            out.copy_(y)

    def _deconv(self, name: str, x: torch.Tensor, out: torch.Tensor) -> None:
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        if self.conv_backend == "hip":
            ops.deconv4x4s2(x, w, b, out=out)
        else:
            out.copy_(F.conv_transpose2d(x, w, b, stride=2, padding=1))

    # ---- the forward ----------------------------------------------------------------------------
    def run(self, x: torch.Tensor) -> torch.Tensor:
        B = self.B
        if tuple(x.shape) != (B, 6, self.H, self.W) or x.dtype != self.dtype or x.device != self.device:
            raise ValueError("plan built for %s %s on %s, got %s %s on %s" % (
                (B, 6, self.H, self.W), self.dtype, self.device, tuple(x.shape), x.dtype, x.device))
        x = ops.densify(x)
        self.conv_macs = {"direct": 0, "executed": 0}           # 3x3 layers of this run (bench.py: utilisation of the matrix cores)
        # -- feature pyramid, both images as one 2B batch (PWCNet.py:184-195) ----------------------
        self._pyramid([(x[:, :3], 0, B), (x[:, 3:], B, 2 * B)], 0, 2 * B)
        return self._decode()

    def _pyramid(self, images, lo: int, hi: int) -> None:
        """conv1a..conv6b over the batch slots [lo,hi) of the pyramid buffers; `images` lists
        (rgb [n,3,H,W], slot_lo, slot_hi) sources for the first conv."""
        prev = None
        for l in range(1, 7):
            na, naa, nb = self.pyramid_names[l - 1]
            a, bb = self.pyr_a[l][lo:hi], self.pyr_b[l][lo:hi]
            # three convs a -> bb -> a (PWCNet.py:184-195); two for PWCDCNet_old, bb -> a (PWCNet.py:411-422)
            first = self.pyr_a[l] if naa is not None else self.pyr_b[l]
            if l == 1:
                for img, s0, s1 in images:
                    self._conv(na, img, first[s0:s1], stride=2)
            else:
                self._conv(na, prev, first[lo:hi], stride=2)
            if naa is not None:
                self._conv(naa, a, bb)
            self._conv(nb, bb, a)
            prev = a

    def _decode(self) -> torch.Tensor:
        """Coarse-to-fine decoder + context network over self.c1[l] / self.c2[l] (PWCNet.py:198-268)."""
        for l in (6, 5, 4, 3, 2):
            self._level_entry(l)
            self._dense(l)
            self._heads(l)
        return self._context()

    def _level_entry(self, l: int) -> None:
        """Cost volume of level l into its arena slot (PWCNet.py:198-199, 212-214 ...): plain correlation at level 6; below, c1
        into its slot and warp(c2, up_flow * scale) + correlation + LeakyReLU as one kernel."""
        nd, base = self.nd, self.arena_base[l]
        ar = self.arena[l]
        c = PYRAMID_CH[l]
        corr_slot = ar[:, base:base + nd]
        if l == 6:
            ops.correlation(self.c1[6], self.c2[6], self.md, 1, self.md, 1, 1, 1.0,
                            normalize=self.normalize_corr, leaky_slope=LEAKY, out=corr_slot)
            return
        off = base + nd
        # first image's level features go into the level's arena slot (they are part of the
        # dense block's input, PWCNet.py:215); c2 is only ever read by the warp
        if not self.c1_in_arena:
            ar[:, off:off + c].copy_(self.c1[l])
        up_flow = ar[:, off + c:off + c + 2]
        # warp + correlation + LeakyReLU as one kernel (the warped features live in LDS only) where the geometry
        # allows it (md = 4, W % 4 == 0) and the map is more than a few tiles; otherwise the two operators
        fused = None
        if self.fuse_warp and self.md == 4 and ops.warp_correlation_preferred(ar.shape[0], c, *self.size[l]):
            fused = ops.warp_correlation(ar[:, off:off + c], self.c2[l], up_flow, flow_scale=WARP_SCALE[l],
                                         align_corners=self.align_corners, mask_threshold=self.mask_threshold,
                                         normalize=self.normalize_corr, leaky_slope=LEAKY, out=corr_slot)
        if fused is None:
            ops.warp(self.c2[l], up_flow, flow_scale=WARP_SCALE[l], align_corners=self.align_corners,
                     mask_threshold=self.mask_threshold, out=self.warped[l])
            ops.correlation(ar[:, off:off + c], self.warped[l], self.md, 1, self.md, 1, 1, 1.0,
                            normalize=self.normalize_corr, leaky_slope=LEAKY, out=corr_slot)

    def _dense(self, l: int) -> None:
        """DenseNet block convL_0..convL_4 (PWCNet.py:202-206 ...): each layer reads a channel suffix of the arena and writes
        its slice in front of it."""
        ar = self.arena[l]
        lo = self.arena_base[l]
        for i, (co, off_i) in enumerate(zip(DENSE_OUT, DENSE_OFF)):
            self._conv("conv%d_%d" % (l, i), ar[:, lo:], ar[:, off_i:off_i + co])
            lo = off_i

    def _heads(self, l: int) -> None:
        """predict_flowL, and for l > 2 deconvL / upfeatL into the next level's arena (PWCNet.py:207-209 ...)."""
        ar = self.arena[l]
        if l == 2:
            self._conv("predict_flow2", ar, self.flow[2], act=False)
            return
        nxt = self.arena[l - 1]
        cn = PYRAMID_CH[l - 1]
        o = self.arena_base[l - 1] + self.nd + cn
        h, w = self.size[l]
        if l in self.head10:
            self._conv("head10_%d" % l, ar, self.head10[l], act=False)
            ops.upsample_entry(self.head10[l], self.p["deconv%d.weight" % l], self.p["deconv%d.bias" % l], nxt[:, o:o + 4])
            return
        if self.stream_head[l]:
            # predict_flowL and upfeatL read the same 3x3 windows of the same arena: one pass
            ops.head_upfeat(ar, self.packed["predict_flow%d" % l], self.p["predict_flow%d.bias" % l],
                            self.p["upfeat%d.weight" % l], self.p["upfeat%d.bias" % l],
                            self.flow[l], nxt[:, o + 2:o + 4], workspace=self.workspace)
        else:
            self._conv("predict_flow%d" % l, ar, self.flow[l], act=False)
            self._deconv("upfeat%d" % l, ar, nxt[:, o + 2:o + 4])
        self._deconv("deconv%d" % l, self.flow[l], nxt[:, o:o + 2])

    def _context(self) -> torch.Tensor:
        """Context network + residual (PWCNet.py:267-268)."""
        if self.ctx_lattice:
            return self._context_lattice()
        t = self.arena[2]
        for i, (_, dil) in enumerate(CONTEXT):
            self._conv("dc_conv%d" % (i + 1), t, self.ctx[i], dilation=dil)
            t = self.ctx[i]
        self._conv("dc_conv7", t, self.flow_out, act=False, residual=self.flow[2])
        return self.flow_out

    def _context_lattice(self) -> torch.Tensor:
        """The same seven layers with dc_conv1..5 in lattice-major layout (see __init__): every layer is the reference's layer on the
        same values, only the order in which pixels are stored between them differs."""
        B = self.B

        def w4(name, x, out, split2):
            key = name + ".0"
            w, b = self.p[key + ".weight"], self.p[key + ".bias"]
            macs = x.shape[0] * w.shape[0] * w.shape[1] * 9 * x.shape[2] * x.shape[3]
            self.conv_macs["direct"] += macs
            self.conv_macs["executed"] += macs * 36 // 144
            ops.conv3x3_wino4(x, self.wino4_packed[key], b, w.shape[0], leaky_slope=LEAKY, out=out, split2=split2, workspace=self.workspace)

        w4("dc_conv1", self.arena[2], self.ctx[0], True)          # [B,565,H,W]      -> 4B  x [128,H/2,W/2]   (lattices of dilation 2)
        w4("dc_conv2", self.ctx[0], self.ctx[1], True)            # dilation 2 = 1 on those -> 16B x [128,H/4,W/4]
        w4("dc_conv3", self.ctx[1], self.ctx[2], True)            # dilation 4 = 1          -> 64B x [128,H/8,W/8]
        if getattr(self, "dc4_split", False):                     # dilation 8 = 1 on the dilation-8 lattices: 64 couts F(4x4) + 32 couts F(2x2)
            self._conv_split96("dc_conv4.0", self.ctx[2], self.ctx[3])
        else:
            self._conv("dc_conv4", self.ctx[2], self.ctx[3], dilation=1)
        self._conv("dc_conv5", self.ctx[3], self.ctx4_lat, dilation=2)      # dilation 16 = 2 on them
        ops.lattice_unsplit(self.ctx4_lat, B, 3, out=self.ctx[4])
        self._conv("dc_conv6", self.ctx[4], self.ctx[5], dilation=1)
        self._conv("dc_conv7", self.ctx[5], self.flow_out, act=False, residual=self.flow[2])
        return self.flow_out

    def flows(self) -> Tuple[torch.Tensor, ...]:
        """(flow2, flow3, flow4, flow5, flow6) of the last run -- the training-mode return (PWCNet.py:270-271)."""
        return (self.flow_out, self.flow[3], self.flow[4], self.flow[5], self.flow[6])

    def bytes_allocated(self) -> int:
        tot = 0
        for group in (self.pyr_a, self.pyr_b, self.warped, self.arena, self.flow):
            tot += sum(t.numel() * t.element_size() for t in group.values())
        if self.c1_in_arena:                                    # pyr_a[2..5] are views of the arenas, which hold 2B items
            for l in range(2, 6):
                tot += (self.arena[l].numel() - self.pyr_a[l].numel()) * self.arena[l].element_size()
        tot += sum(t.numel() * t.element_size() for t in self.ctx + ([self.flow_out] if self.flow_out is not None else []))
        if getattr(self, "ctx_lattice", False):
            tot += self.ctx4_lat.numel() * self.ctx4_lat.element_size()
        tot += sum(t.numel() * t.element_size() for t in self.packed.values())
        tot += sum(t.numel() * t.element_size() for t in self.wino_packed.values())
        tot += sum(t.numel() * t.element_size() for t in self.wino4_packed.values())
        if self.workspace is not None:
            tot += self.workspace.numel() * self.workspace.element_size()
        return tot


class PwcVideoPlan(PwcPlan):
    """Plan for consecutive frame pairs (f[t], f[t+1]) of one video, the loop of
    ``pwc_extract_flow_video.py:262-305`` (``frame1 = frame2`` each iteration).

    The reference runs the feature pyramid on both frames of every pair, i.e. twice per frame.  Here each
    level's pyramid buffer has B+1 batch slots: slot 0 carries the last frame of the previous step, slots
    1..B receive the pyramid of the B new frames, and the decoder's first/second-image features are the
    overlapping views [0:B] / [1:B+1] of the same buffer -- no copy, one pyramid pass per frame.
    """

    def __init__(self, params, B, H, W, device, dtype=torch.float32, md=4, normalize_corr=False,
                 align_corners=False, conv_backend="hip", variant="dc"):
        super().__init__(params, B, H, W, device, dtype, md, normalize_corr, align_corners, conv_backend, variant)
        self.primed = False

    @staticmethod
    def _slots(B: int) -> int:
        return B + 1

    @staticmethod
    def _pair_views(buf: torch.Tensor, B: int):
        return buf[:B], buf[1:]

    def _check(self, frames: torch.Tensor, n: int) -> torch.Tensor:
        if tuple(frames.shape) != (n, 3, self.H, self.W) or frames.dtype != self.dtype or frames.device != self.device:
            raise ValueError("expected frames %s %s on %s, got %s %s on %s" % (
                (n, 3, self.H, self.W), self.dtype, self.device, tuple(frames.shape), frames.dtype, frames.device))
        return ops.densify(frames)

    def _carry(self) -> None:
        for l in range(2, 7):
            self.pyr_a[l][0].copy_(self.pyr_a[l][self.B])

    def prime(self, frame: torch.Tensor) -> None:
        """Pyramid of the very first frame [1,3,H,W] -> carry slot."""
        f = self._check(frame, 1)
        self._pyramid([(f, self.B, self.B + 1)], self.B, self.B + 1)
        self._carry()
        self.primed = True

    def push(self, frames: torch.Tensor) -> torch.Tensor:
        """B new frames [B,3,H,W] -> flows of the pairs (previous, frames[0]), (frames[0], frames[1]), ..."""
        if not self.primed:
            raise RuntimeError("PwcVideoPlan.push before prime(first_frame)")
        f = self._check(frames, self.B)
        self._pyramid([(f, 1, self.B + 1)], 1, self.B + 1)
        out = self._decode()
        self._carry()
        return out

    def run(self, x):
        raise RuntimeError("PwcVideoPlan is driven by prime()/push(), not run()")
