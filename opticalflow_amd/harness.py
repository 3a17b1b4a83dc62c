"""Single-pair inference harness: the pre/post-processing of the reference's canonical CLI
(`script_pwc.py:43-83`) around PWCDCNet.forward, on the device.

    pre : drop alpha, resize both images to ceil(H/64)*64 x ceil(W/64)*64 (bilinear), RGB->BGR, /255,
          HWC->CHW, stack to [1,6,H_,W_]                                        (script_pwc.py:43-65)
    net : flow2 = net(x)                                                         (script_pwc.py:71)
    post: flo = flow2[0]*20, resize u and v to (W,H) (bilinear), u *= W/W_, v *= H/H_   (script_pwc.py:72-81)
    out : Middlebury .flo                                                        (script_pwc.py:12-27,83)

`cv2.resize(..., INTER_LINEAR)` samples at half-pixel centres without antialiasing, i.e.
`F.interpolate(mode="bilinear", align_corners=False, antialias=False)`; cv2 itself is not available in
this project's environments, so that equivalence is pinned by an independent numpy statement in
tests/test_harness_cpu.py rather than against cv2 (documented: "parity unpinned" w.r.t. cv2's fixed-point
arithmetic on uint8 images, which rounds the resized image to integers -- see `quantize_like_cv2`).

CLI:  python -m opticalflow_amd.harness im1.png im2.png out.flo [--weights pwc_net.pth.tar]
"""
from __future__ import annotations

import math
import sys
from typing import Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .flowio import write_flo

DIVISOR = 64.0


def padded_size(h: int, w: int) -> Tuple[int, int]:
    """script_pwc.py:48-53."""
    return int(math.ceil(h / DIVISOR) * DIVISOR), int(math.ceil(w / DIVISOR) * DIVISOR)


def _resize_bilinear(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
    if t.shape[-2:] == (h, w):
        return t
    return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False, antialias=False)


def preprocess(im1: torch.Tensor, im2: torch.Tensor, quantize_like_cv2: bool = True) -> torch.Tensor:
    """im1, im2: [H,W,3+] uint8 (or float 0..255) RGB images (any device) -> [1,6,H_,W_] float32 BGR in [0,1].

    cv2.resize on a uint8 image returns uint8 (rounded); `quantize_like_cv2` reproduces that rounding step
    (round-half-up of the interpolated value) so that an image which needs resizing goes through the same
    256-level quantisation as in the reference.
    """
    outs = []
    h, w = im1.shape[:2]
    h_, w_ = padded_size(h, w)
    for im in (im1, im2):
        if im.shape[:2] != (h, w):
            raise ValueError("image sizes differ: %s vs %s" % (tuple(im1.shape), tuple(im.shape)))
        t = im[:, :, :3].to(torch.float32).permute(2, 0, 1).unsqueeze(0)       # drop alpha, HWC->1CHW
        if (h_, w_) != (h, w):
            t = _resize_bilinear(t, h_, w_)
            if quantize_like_cv2:
                t = torch.floor(t + 0.5).clamp_(0, 255)
        t = t.flip(1) / 255.0                                                    # RGB->BGR, /255
        outs.append(t)
    return torch.cat(outs, 1).contiguous()


def postprocess(flow2: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """flow2: [1,2,H_/4,W_/4] network output -> [H,W,2] flow in pixels of the original image."""
    h_, w_ = padded_size(h, w)
    flo = flow2[:1] * 20.0
    flo = _resize_bilinear(flo, h, w)
    flo[:, 0].mul_(w / float(w_))          # python scalars: a per-call device tensor would be a blocking pageable upload
    flo[:, 1].mul_(h / float(h_))
    return flo[0].permute(1, 2, 0).contiguous()


@torch.no_grad()
def estimate_flow(net, im1: torch.Tensor, im2: torch.Tensor) -> torch.Tensor:
    """Full script_pwc.py pipeline for one pair; images may live on the host (moved to the net's device)."""
    dev = next(net.parameters()).device
    x = preprocess(im1.to(dev), im2.to(dev))
    return postprocess(net(x), im1.shape[0], im1.shape[1])


def read_image(path: str) -> torch.Tensor:
    """PNG/JPEG -> [H,W,C] uint8 RGB(A) tensor (PIL; the reference uses imageio.imread, script_pwc.py:43)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA"):
            im = im.convert("RGB")
        return torch.from_numpy(np.array(im))


def main(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="PWC-Net flow for one image pair (drop-in for the reference's script_pwc.py)")
    ap.add_argument("im1", nargs="?", default="data/frame_0010.png")      # script_pwc.py:30-32 defaults
    ap.add_argument("im2", nargs="?", default="data/frame_0011.png")
    ap.add_argument("out", nargs="?", default="./tmp/frame_0010.flo")
    ap.add_argument("--weights", default="./pwc_net.pth.tar")             # script_pwc.py:41
    ap.add_argument("--cpu-fallback-semantics", action="store_true",
                    help="un-normalised correlation (the reference's USE_ONNX_CORRELATION fallback, this project's parity "
                         "mode) instead of the native /C correlation that checkpoints are trained with")
    ap.add_argument("--align-corners", action="store_true",
                    help="warp with grid_sample(align_corners=True) (torch < 1.3 behaviour, what the published weights saw)")
    args = ap.parse_args(argv)
    from .pwcnet import pwc_dc_net
    net = pwc_dc_net(args.weights, normalize_corr=not args.cpu_fallback_semantics, align_corners=args.align_corners)
    net = net.cuda().eval()
    flo = estimate_flow(net, read_image(args.im1), read_image(args.im2))
    write_flo(args.out, flo.cpu())
    return 0


if __name__ == "__main__":
    sys.exit(main())
