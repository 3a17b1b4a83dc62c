"""Consecutive-frame optical flow for video, the loop of ``pwc_extract_flow_video.py:219-305``.

The reference reads frame t+1, runs the whole network on (frame t, frame t+1) and then does
``frame1 = frame2`` (pwc_extract_flow_video.py:300), so every frame passes through the feature
pyramid twice.  ``FlowStream`` keeps frame t's pyramid resident in HBM (engine.PwcVideoPlan) and
runs the pyramid once per frame; results are the same numbers as ``model(cat(frame_t, frame_t+1))``.

Pre/post-processing helpers follow the reference script:
  * ``frame_to_tensor``   BGR uint8 HWC -> RGB float32 CHW / 255        (pwc_extract_flow_video.py:27-34)
  * ``pad_to_multiple_of_64`` replicate-pad bottom/right                 (:36-42, same as kitti.pad_to_64)
  * ``unpad``             the reference crops the QUARTER-resolution flow by the FULL-resolution pad
                          (:44-47, :213) -- reproduced as is (kitti.model_infer documents the same quirk)
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Tuple

import numpy as np
import torch

from .engine import PwcVideoPlan
from .kitti import pad_to_64, unpad as _unpad

__all__ = ["FlowStream", "frame_to_tensor", "flow_video"]


def frame_to_tensor(frame: np.ndarray) -> torch.Tensor:
    """numpy (H,W,3) BGR uint8 -> tensor (3,H,W) RGB float32 in [0,1] (pwc_extract_flow_video.py:27-34)."""
    if frame.ndim != 3 or frame.shape[2] != 3:
        raise ValueError("expected an (H,W,3) BGR frame, got %s" % (frame.shape,))
    rgb = frame[:, :, ::-1].astype(np.float32) / 255.0
    return torch.from_numpy(np.ascontiguousarray(np.transpose(rgb, (2, 0, 1))))


class FlowStream:
    """flow(frame[t] -> frame[t+1]) for a running sequence of equally sized frames.

    ``net`` is an ``opticalflow_amd.PWCDCNet`` on the ROCm device.  ``batch`` frames are consumed per
    ``push``; ``batch=1`` is the reference's loop.  With ``use_graph`` each push is one HIP-graph replay.
    """

    def __init__(self, net, batch: int, height: int, width: int, use_graph: bool = True):
        params = {k: v.detach() for k, v in net.state_dict(keep_vars=True).items()}
        p0 = next(iter(params.values()))
        if not p0.is_cuda:
            from ._lib import PwcHipError
            raise PwcHipError("FlowStream needs the model on the ROCm device (parameters are on %s)" % p0.device)
        self.device = p0.device
        self.batch, self.height, self.width = batch, height, width
        if getattr(net, "precision", "fp32") == "fp16-strict":
            raise NotImplementedError("FlowStream has plans for precision 'fp32' and 'fp16'; the strict half-precision mode is built "
                                      "for PWCDCNet.forward (pairs / KITTI stream)")
        if getattr(net, "precision", "fp32") == "fp16":
            from .engine_f16 import PwcVideoPlanF16
            self.plan = PwcVideoPlanF16(params, batch, height, width, self.device, net.md, net.normalize_corr,
                                        net.align_corners, getattr(net, "variant", "dc"))
        else:
            self.plan = PwcVideoPlan(params, batch, height, width, self.device, torch.float32, net.md,
                                     net.normalize_corr, net.align_corners, net.conv_backend,
                                     getattr(net, "variant", "dc"))
        self.use_graph = use_graph
        self._graph = None
        self._static = torch.empty((batch, 3, height, width), device=self.device, dtype=torch.float32)

    def prime(self, frame: torch.Tensor) -> None:
        """First frame of the sequence, [1,3,H,W] or [3,H,W]."""
        if frame.dim() == 3:
            frame = frame.unsqueeze(0)
        with torch.no_grad():
            self.plan.prime(frame.to(self.device, torch.float32))

    def push(self, frames: torch.Tensor) -> torch.Tensor:
        """``batch`` further frames [B,3,H,W]; returns [B,2,H/4,W/4], row i = flow from the frame before
        frames[i] to frames[i] (a view of the plan's output buffer: clone it to keep it across pushes)."""
        if frames.dim() == 3:
            frames = frames.unsqueeze(0)
        with torch.no_grad():
            if not self.use_graph:
                return self.plan.push(frames.to(self.device, torch.float32))
            if not self.plan.primed:
                raise RuntimeError("FlowStream.push before prime(first_frame)")
            if tuple(frames.shape) != tuple(self._static.shape):
                raise ValueError("expected frames %s, got %s" % (tuple(self._static.shape), tuple(frames.shape)))
            self._static.copy_(frames)
            if self._graph is None:
                # warm-up outside capture would advance the carry slot: save and restore it around the capture
                keep = {l: self.plan.pyr_a[l][0].clone() for l in range(2, 7)}
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self.plan.push(self._static)
                torch.cuda.current_stream(self.device).wait_stream(side)
                for l, t in keep.items():
                    self.plan.pyr_a[l][0].copy_(t)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.plan.push(self._static)
                for l, t in keep.items():
                    self.plan.pyr_a[l][0].copy_(t)
                self._graph = graph
            self._graph.replay()
            return self.plan.flow_out


def flow_video(net, frames: Iterable[np.ndarray], use_graph: bool = True) -> Iterator[np.ndarray]:
    """Generator over BGR uint8 frames yielding one (h,w,2) float32 flow per consecutive pair, exactly
    what ``process_frame_pair`` returns for (frame_t, frame_t+1) (pwc_extract_flow_video.py:192-216)."""
    stream: Optional[FlowStream] = None
    pads: Tuple[int, int] = (0, 0)
    for frame in frames:
        t = frame_to_tensor(frame).unsqueeze(0)
        tp, pad_h, pad_w = pad_to_64(t)
        if stream is None:
            dev = next(net.parameters()).device
            stream = FlowStream(net, 1, tp.shape[2], tp.shape[3], use_graph=use_graph)
            pads = (pad_h, pad_w)
            stream.prime(tp.to(dev))
            continue
        if tuple(tp.shape[2:]) != (stream.height, stream.width):
            raise ValueError("frame size changed mid-stream")
        flow = stream.push(tp.to(stream.device))
        flow = _unpad(flow, pads[0], pads[1])     # full-resolution pad, like the reference
        yield flow.squeeze(0).permute(1, 2, 0).contiguous().cpu().numpy()
