"""PWCDCNet with the reference's constructor, state-dict and forward()/warp() surface, executed by
hand-written gfx950 kernels (libpwc_hip.so) instead of cuDNN + the CUDA correlation extension.

Reference: ``models/PWCNet.py:40-273`` (class), ``:497-506`` (factory ``pwc_dc_net``).

The module owns ordinary ``nn.Conv2d`` / ``nn.ConvTranspose2d`` children purely as parameter
containers, registered under the reference's names and in the reference's order, so

  * ``state_dict()`` has the reference's 128 keys (``conv1a.0.weight`` ... ``dc_conv7.bias``,
    including the never-used ``deconv2.*``, PWCNet.py:124) and ``load_state_dict(strict=True)``
    accepts a reference ``.pth.tar`` unchanged;
  * ``forward(x: [B,6,H,W]) -> flow2: [B,2,H/4,W/4]`` in eval mode, the 5-tuple
    ``(flow2, flow3, flow4, flow5, flow6)`` in training mode (PWCNet.py:270-273);
  * ``warp(x, flo)`` is the fused HIP warp.

Additive constructor keywords: ``normalize_corr`` / ``align_corners`` (SURVEY.md section 0, facts 3
and 4).  The CONSTRUCTOR defaults reproduce the reference's CPU fallback as executed on a current torch
(un-normalised cost volume, ``grid_sample(align_corners=False)``) -- this project's parity definition;
the checkpoint-loading factories ``pwc_dc_net(path)`` / ``pwc_dc_net_old(path)`` default to the native
semantics a checkpoint was trained with (``normalize_corr=True``, see ``_checkpoint_kwargs``), ``conv_backend`` ('hip' = MFMA implicit-GEMM
kernels, 'torch' = BASELINE config[1], convolutions left to PyTorch-ROCm), ``use_graph``.

Inference only: the HIP path does not build an autograd graph (returned flows have
``requires_grad=False``).  Tensors must live on a ROCm device; there is no CPU execution path.
"""
from __future__ import annotations

import warnings
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import PwcHipError
from .correlation import Correlation, onnx_correlation_enabled
from .engine import (CONTEXT, DENSE_OUT, PYRAMID_CH, PYRAMID_NAMES, PYRAMID_NAMES_OLD, PwcPlan,
                     level_in_channels)
from .weights import load_checkpoint, synthetic_state_dict

__all__ = ["PWCDCNet", "PWCDCNet_old", "pwc_dc_net", "pwc_dc_net_old"]


def _conv_block(cin: int, cout: int, stride: int = 1, dilation: int = 1) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation, dilation=dilation, bias=True),
                         nn.LeakyReLU(0.1))


class PWCDCNet(nn.Module):
    variant = "dc"
    _pyramid_names = PYRAMID_NAMES

    def __init__(self, md: int = 4, normalize_corr: bool = False, align_corners: bool = False,
                 conv_backend: str = "hip", use_graph: bool = False, precision: str = "fp32", borrow_output: bool = False):
        super().__init__()
        if precision not in ("fp32", "fp16", "fp16-strict"):
            raise ValueError("precision must be 'fp32', 'fp16' or 'fp16-strict'")
        # 'fp16' (BASELINE configs 3-4): float32 parameters and float32 input/output as in the reference's interface,
        # half-precision activations and filters inside, fp32 accumulation (engine_f16.PwcPlanF16): ~1.2e-3 of mean |flow|
        # from the fp32 result.  'fp16-strict' (engine_strict.PwcPlanStrict): the variant that meets north_star's 1e-3 mean
        # EPE -- pyramid / levels 6..3 / every warp in fp32, the level-2 block and the context network (79 % of the
        # multiplications) in half with split (hi + lo) filters
        self.precision = precision
        self.md = md
        self.normalize_corr = normalize_corr
        self.align_corners = align_corners
        self.conv_backend = conv_backend
        self.use_graph = use_graph
        # borrow_output: eval-mode forwards return the plan's own flow buffer instead of a copy of it -- valid until the next forward
        # of the same geometry overwrites it (one 5 us copy less per forward; for callers that consume the flow at once, like the
        # sharded benchmark loop, whose gather copies it into its own staging buffer)
        self.borrow_output = bool(borrow_output)

        # registration order == reference order (PWCNet.py:52-132)
        for lvl, (na, naa, nb) in enumerate(self._pyramid_names, start=1):
            cin, cout = PYRAMID_CH[lvl - 1], PYRAMID_CH[lvl]
            self.add_module(na, _conv_block(cin, cout, stride=2))
            if naa is not None:
                self.add_module(naa, _conv_block(cout, cout))
            self.add_module(nb, _conv_block(cout, cout))
        self.corr = Correlation(pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1,
                                corr_multiply=1, normalize=normalize_corr)
        self.leakyRELU = nn.LeakyReLU(0.1)
        nd = (2 * md + 1) ** 2
        for lvl in (6, 5, 4, 3, 2):
            od = level_in_channels(lvl, nd)
            grown = od
            for i, cout in enumerate(DENSE_OUT):
                self.add_module("conv%d_%d" % (lvl, i), _conv_block(grown, cout))
                grown += cout
            self.add_module("predict_flow%d" % lvl, nn.Conv2d(grown, 2, 3, padding=1, bias=True))
            self.add_module("deconv%d" % lvl, nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
            if lvl > 2:
                self.add_module("upfeat%d" % lvl, nn.ConvTranspose2d(grown, 2, 4, 2, 1, bias=True))
        cin = level_in_channels(2, nd) + sum(DENSE_OUT)
        for i, (cout, dil) in enumerate(CONTEXT, start=1):
            self.add_module("dc_conv%d" % i, _conv_block(cin, cout, dilation=dil))
            cin = cout
        self.dc_conv7 = nn.Conv2d(cin, 2, 3, padding=1, bias=True)

        # same initialisation as the reference (PWCNet.py:134-138)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight.data, mode="fan_in")
                if m.bias is not None:
                    m.bias.data.zero_()

        # plans (buffers + packed filters, ~0.2 GB per pair at 1024x448) and captured graphs are cached per
        # (batch, size, ...) key, least recently used first; at most `max_cached_plans` are kept so that a caller with
        # varying image sizes (harness.estimate_flow) cannot exhaust HBM
        self.max_cached_plans = 4
        self._plans: "OrderedDict[Tuple, PwcPlan]" = OrderedDict()
        self._graphs: Dict[Tuple, Tuple] = {}
        self._warned_detached = False
        self._versions: Optional[Tuple[int, ...]] = None
        self._param_list = None

    # ---- reference surface --------------------------------------------------------------------
    def warp(self, x: torch.Tensor, flo: torch.Tensor) -> torch.Tensor:
        """warp an image/tensor (im2) back to im1 according to the optical flow (PWCNet.py:141-177)."""
        thr = 0.9999 if self.variant == "dc" else 0.999
        if torch.is_grad_enabled() and (x.requires_grad or flo.requires_grad):
            return ops.WarpFunction.apply(x, flo, 1.0, self.align_corners, thr)
        return ops.warp(ops.densify(x), ops.densify(flo), 1.0, self.align_corners, thr)

    def forward(self, x: torch.Tensor):
        if self.training and torch.is_grad_enabled() and not self._warned_detached:
            # the convolutions have no backward kernels (Correlation and warp do): say so instead of silently handing a
            # training loop flows that do not require grad
            warnings.warn("PWCDCNet (HIP path) is inference-only: the training-mode 5-tuple is returned DETACHED "
                          "(requires_grad=False), so the reference's train*.py loops cannot back-propagate through it. "
                          "Call it under torch.no_grad() to silence this warning.", stacklevel=2)
            self._warned_detached = True
        with torch.no_grad():
            return self._forward(x)

    def _forward(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != 6:
            raise ValueError("expected [B,6,H,W] (two stacked 3-channel images), got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise PwcHipError("PWCDCNet.forward needs a tensor on the ROCm device (got %s): the HIP path has no "
                              "CPU fallback" % x.device)
        plan = self._plan_for(x)
        key = self._key(x)
        if self.use_graph and not self.training:
            out = self._run_graph(key, plan, x)
        else:
            out = plan.run(x)
        if self.training:
            return tuple(t.clone() for t in plan.flows())
        return out if self.borrow_output else out.clone()

    # ---- plan management -----------------------------------------------------------------------
    def _normalize_now(self) -> bool:
        """The reference's net switches its correlation with the module global USE_ONNX_CORRELATION
        (correlation.py:103-110: flag on -> the un-normalised torch expression, flag off -> the native operator).  The
        forward honours the flag the same way: with it on, the cost volumes are un-normalised whatever
        ``normalize_corr`` says (computed by the same HIP kernels -- the two expressions differ by the factor 1/C only)."""
        return bool(self.normalize_corr) and not onnx_correlation_enabled()

    def _key(self, x):
        return (x.shape[0], x.shape[2], x.shape[3], x.dtype, x.device, self.conv_backend,
                self._normalize_now(), self.align_corners, self.precision)

    def _param_versions(self):
        # in-place updates bump _version; re-homing (.to/.cuda/load_state_dict) goes through _apply /
        # load_state_dict below, which drop the plans explicitly
        if self._param_list is None:
            self._param_list = list(self.parameters())
        return tuple(p._version for p in self._param_list)

    def invalidate_plans(self):
        self._plans.clear()
        self._graphs.clear()
        self._versions = None
        self._param_list = None

    def _plan_for(self, x) -> PwcPlan:
        ver = self._param_versions()
        if ver != self._versions:
            self.invalidate_plans()
            self._versions = ver
        key = self._key(x)
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
        if plan is None:
            params = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
            for k, v in params.items():
                if v.device != x.device:
                    raise PwcHipError("parameter %s is on %s but the input is on %s: call net.to(device) first"
                                      % (k, v.device, x.device))
                if v.dtype != torch.float32:
                    raise NotImplementedError("parameters must be float32 (got %s for %s)" % (v.dtype, k))
            if self.precision in ("fp16", "fp16-strict"):
                if self.conv_backend != "hip" or x.dtype != torch.float32:
                    raise NotImplementedError("precision='%s' is built for conv_backend='hip' and float32 input" % self.precision)
                if self.precision == "fp16":
                    from .engine_f16 import PwcPlanF16 as Plan16
                else:
                    from .engine_strict import PwcPlanStrict as Plan16
                plan = Plan16(params, x.shape[0], x.shape[2], x.shape[3], x.device, self.md,
                              self._normalize_now(), self.align_corners, self.variant)
            else:
                plan = PwcPlan(params, x.shape[0], x.shape[2], x.shape[3], x.device, x.dtype, self.md,
                               self._normalize_now(), self.align_corners, self.conv_backend, self.variant)
            self._plans[key] = plan
            while len(self._plans) > max(1, int(self.max_cached_plans)):
                old, _ = self._plans.popitem(last=False)
                self._graphs.pop(old, None)
        return plan

    def _run_graph(self, key, plan: PwcPlan, x: torch.Tensor) -> torch.Tensor:
        entry = self._graphs.get(key)
        if entry is None:
            static_in = torch.empty_like(x, memory_format=torch.contiguous_format)
            static_in.copy_(x)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side):           # warm-up outside capture (first-launch attributes)
                plan.run(static_in)
            torch.cuda.current_stream(x.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                plan.run(static_in)
            entry = (graph, static_in)
            self._graphs[key] = entry
        graph, static_in = entry
        if x.data_ptr() != static_in.data_ptr():        # a caller that fills graph_input() in place skips this copy
            static_in.copy_(x)
        graph.replay()
        return plan.flow_out

    @torch.no_grad()
    def graph_input(self, batch: int, height: int, width: int, device=None) -> torch.Tensor:
        """The captured forward's own input buffer [B,6,H,W] (requires ``use_graph``): write the next pairs into it
        (e.g. as the destination of the H2D copy or of a decoder) and pass it to ``forward`` -- the 11 MB-per-pair
        device-to-device staging copy of a foreign input tensor is then skipped."""
        if not self.use_graph:
            raise RuntimeError("graph_input() needs use_graph=True")
        device = device or next(self.parameters()).device
        x = torch.zeros((batch, 6, height, width), device=device)
        key = self._key(x)
        if key not in self._graphs:
            training, self.training = self.training, False
            try:
                self.forward(x)
            finally:
                self.training = training
        return self._graphs[key][1]

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_plans()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        self.invalidate_plans()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def manifest(self) -> List[Tuple[str, Tuple[int, ...]]]:
        return [(k, tuple(v.shape)) for k, v in self.state_dict().items()]


class PWCDCNet_old(PWCDCNet):
    """The reference's earlier variant (``models/PWCNet.py:277-491``): 116-key state-dict (no ``conv*aa``),
    dense-block concatenation order ``cat(x, conv_0(x)); cat(conv_1(x), x); cat(x, conv_2(x)); ...`` and warp mask
    threshold 0.999.  Runs on the same kernels and arena: only the filters' input-channel order is re-mapped
    when the plan packs them (engine.old_variant_perm)."""
    variant = "old"
    _pyramid_names = PYRAMID_NAMES_OLD


def _checkpoint_kwargs(path: Optional[str], kwargs: dict) -> dict:
    """A checkpoint was trained against the reference's NATIVE correlation, which divides by kernel_size^2 * C
    (correlation_cuda_kernel.cu:104,143; reached through Correlation.forward -> CorrelationFunction whenever
    USE_ONNX_CORRELATION is off, correlation.py:103-117).  The factories therefore default to ``normalize_corr=True``
    when they load a file; without a file they keep the constructor's default (un-normalised = the semantics of the
    reference's CPU fallback, this project's parity definition).  Asking for the un-normalised cost volume WITH a
    checkpoint is allowed but warned about: the cost volume would be C (32..196) times larger than in training."""
    if path is None:
        return kwargs
    kwargs = dict(kwargs)
    if "normalize_corr" not in kwargs:
        kwargs["normalize_corr"] = True
    elif not kwargs["normalize_corr"]:
        warnings.warn("loading %s with normalize_corr=False: the reference's native correlation (the one checkpoints are "
                      "trained with) divides by C; un-normalised cost volumes are 32-196x larger and published weights "
                      "give wrong flow with them" % path, stacklevel=3)
    return kwargs


def pwc_dc_net_old(path: Optional[str] = None, **kwargs) -> PWCDCNet_old:
    """Factory with the reference's name and argument (PWCNet.py:511-520); see _checkpoint_kwargs for the default
    correlation semantics when ``path`` is given."""
    model = PWCDCNet_old(**_checkpoint_kwargs(path, kwargs))
    if path is not None:
        model.load_state_dict(load_checkpoint(path))
    return model


def pwc_dc_net(path: Optional[str] = None, **kwargs) -> PWCDCNet:
    """Factory with the reference's name and argument (PWCNet.py:497-506); see _checkpoint_kwargs for the default
    correlation semantics when ``path`` is given."""
    model = PWCDCNet(**_checkpoint_kwargs(path, kwargs))
    if path is not None:
        model.load_state_dict(load_checkpoint(path))
    return model
