"""Single-pair inference harness: the pre/post-processing of the reference's canonical CLI
(`script_pwc.py:43-83`) around PWCDCNet.forward, on the device.

    pre : drop alpha, resize both images to ceil(H/64)*64 x ceil(W/64)*64 (bilinear), RGB->BGR, /255,
          HWC->CHW, stack to [1,6,H_,W_]                                        (script_pwc.py:43-65)
    net : flow2 = net(x)                                                         (script_pwc.py:71)
    post: flo = flow2[0]*20, resize u and v to (W,H) (bilinear), u *= W/W_, v *= H/H_   (script_pwc.py:72-81)
    out : Middlebury .flo                                                        (script_pwc.py:12-27,83)

`cv2.resize(src, (W, H))` (INTER_LINEAR, script_pwc.py:54,77-78) is restated here operation for operation from
OpenCV's published algorithm (modules/imgproc/src/resize.cpp, `resizeGeneric_` with `HResizeLinear` / `VResizeLinear`):
  * geometry: fx = (float)((dx + 0.5) * (1 / (dst / src)) - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, 0);
    sx >= src - 1 -> (src - 1, 0);
  * uint8 images: 11-bit fixed-point coefficients a = round_half_even((1 - fx) * 2048), round_half_even(fx * 2048);
    horizontal pass D = S[sx] * a0 + S[sx+1] * a1 (int32, scale 2^11); vertical pass
    dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2  (the FixedPtCast<int, uchar, 22> specialisation);
  * float32 images (the flow): the same two passes in float32, coefficients (1.f - fx, fx), no rounding.
cv2 itself is not installed in this project's environments (and no cv2 output ships with the reference), so the
restatement is pinned by hand-computed known-answer vectors and an independent scalar-loop statement
(tests/test_harness.py), not against cv2: "parity unpinned" w.r.t. an actual cv2 build (whose SIMD / IPP variants
are documented to match the scalar code above).

CLI:  python -m opticalflow_amd.harness im1.png im2.png out.flo [--weights pwc_net.pth.tar]
"""
from __future__ import annotations

import math
import sys
from typing import Tuple

import numpy as np
import torch
from .flowio import write_flo

DIVISOR = 64.0


def padded_size(h: int, w: int) -> Tuple[int, int]:
    """script_pwc.py:48-53."""
    return int(math.ceil(h / DIVISOR) * DIVISOR), int(math.ceil(w / DIVISOR) * DIVISOR)


COEF_BITS = 11                       # INTER_RESIZE_COEF_BITS
COEF_SCALE = 1 << COEF_BITS          # INTER_RESIZE_COEF_SCALE = 2048


def _cv2_axis(src: int, dst: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Per destination index: (first tap s0, second tap s1 = min(s0+1, src-1), fractional weight f of s1) exactly as
    cv::resize builds xofs / alpha for INTER_LINEAR (double scale, float fx, clamps set fx = 0)."""
    scale = 1.0 / (float(dst) / float(src))                                    # scale_x = 1. / inv_scale_x, in double
    d = torch.arange(dst, dtype=torch.float64, device=device)
    fx = ((d + 0.5) * scale - 0.5).to(torch.float32)                           # (float)((dx+0.5)*scale_x - 0.5)
    sx = torch.floor(fx)
    fx = fx - sx                                                               # float32
    sx = sx.to(torch.int64)
    lo, hi = sx < 0, sx >= src - 1
    fx = torch.where(lo | hi, torch.zeros_like(fx), fx)
    sx = torch.where(lo, torch.zeros_like(sx), torch.where(hi, torch.full_like(sx, src - 1), sx))
    return sx, torch.clamp(sx + 1, max=src - 1), fx


def cv2_resize_linear(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """cv2.resize(img, (out_w, out_h)) with the default INTER_LINEAR for a uint8 or float32 [H,W] / [H,W,C] tensor
    (any device); returns the input's dtype.  See the module docstring for the arithmetic."""
    if img.dim() not in (2, 3) or img.dtype not in (torch.uint8, torch.float32):
        raise ValueError("expected a uint8 or float32 [H,W] or [H,W,C] image, got %s %s" % (img.dtype, tuple(img.shape)))
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img.clone()                                                     # cv::resize copies when the size matches
    x0, x1, fx = _cv2_axis(w, out_w, img.device)
    y0, y1, fy = _cv2_axis(h, out_h, img.device)
    shape_x = (1, out_w) + (1,) * (img.dim() - 2)
    shape_y = (out_h, 1) + (1,) * (img.dim() - 2)
    if img.dtype == torch.float32:
        a0, a1 = (1.0 - fx).view(shape_x), fx.view(shape_x)
        b0, b1 = (1.0 - fy).view(shape_y), fy.view(shape_y)
        rows = img[:, x0] * a0 + img[:, x1] * a1                               # HResizeLinear (float)
        return rows[y0] * b0 + rows[y1] * b1                                   # VResizeLinear (float)
    # saturate_cast<short>(coef * 2048) = cvRound = round half to even
    a0 = torch.round((1.0 - fx) * COEF_SCALE).to(torch.int32).view(shape_x)
    a1 = torch.round(fx * COEF_SCALE).to(torch.int32).view(shape_x)
    b0 = torch.round((1.0 - fy) * COEF_SCALE).to(torch.int32).view(shape_y)
    b1 = torch.round(fy * COEF_SCALE).to(torch.int32).view(shape_y)
    src = img.to(torch.int32)
    rows = src[:, x0] * a0 + src[:, x1] * a1                                   # int32, scale 2^11
    s0, s1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0 * s0) >> 16) + ((b1 * s1) >> 16) + 2) >> 2
    return out.clamp_(0, 255).to(torch.uint8)


def preprocess(im1: torch.Tensor, im2: torch.Tensor) -> torch.Tensor:
    """im1, im2: [H,W,3+] uint8 (or float32 0..255) RGB images (any device) -> [1,6,H_,W_] float32 BGR in [0,1]
    (script_pwc.py:43-65: drop alpha, cv2.resize to multiples of 64, [:, :, ::-1], / 255, HWC -> CHW, stack)."""
    outs = []
    h, w = im1.shape[:2]
    h_, w_ = padded_size(h, w)
    for im in (im1, im2):
        if im.shape[:2] != (h, w):
            raise ValueError("image sizes differ: %s vs %s" % (tuple(im1.shape), tuple(im.shape)))
        t = im[:, :, :3]
        if t.dtype not in (torch.uint8, torch.float32):
            t = t.to(torch.float32)
        t = cv2_resize_linear(t.contiguous(), h_, w_)                            # uint8 stays uint8 (256 levels), like cv2
        t = t.to(torch.float32).permute(2, 0, 1).unsqueeze(0)                   # HWC -> 1CHW
        outs.append(t.flip(1) / 255.0)                                          # RGB -> BGR, / 255
    return torch.cat(outs, 1).contiguous()


def postprocess(flow2: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """flow2: [1,2,H_/4,W_/4] network output -> [H,W,2] flow in pixels of the original image (script_pwc.py:72-81:
    x20, cv2.resize of u and v to (W, H), u *= W/W_, v *= H/H_)."""
    h_, w_ = padded_size(h, w)
    flo = (flow2[0] * 20.0).to(torch.float32)
    u = cv2_resize_linear(flo[0].contiguous(), h, w) * (w / float(w_))          # python scalars: no per-call device upload
    v = cv2_resize_linear(flo[1].contiguous(), h, w) * (h / float(h_))
    return torch.stack((u, v), dim=-1).contiguous()


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
