"""script_pwc.py-style harness (opticalflow_amd/harness.py).  cv2.resize (INTER_LINEAR) is restated from OpenCV's
published algorithm (resize.cpp: float geometry, 11-bit fixed-point coefficients and the two-pass
((b*(D>>4))>>16 ... +2)>>2 arithmetic for uint8, plain float32 passes for the flow) and pinned here by
  * hand-computed known-answer vectors (worked out in the comments below), and
  * an independent scalar-loop statement of the same published algorithm (`cv2_linear_scalar`),
cv2 itself being absent from this project's environments ("parity unpinned" against an actual cv2 build)."""
import math

import numpy as np
import pytest
import torch

from conftest import seeded_rand
from oracle import pwc_oracle as O
from opticalflow_amd import harness


def _axis_scalar(src, dst):
    """(s0, s1, fx float32) per destination index, plain Python, per cv::resize's xofs/alpha loop."""
    scale = 1.0 / (float(dst) / float(src))
    out = []
    for d in range(dst):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            sx, fx = 0, np.float32(0)
        if sx >= src - 1:
            sx, fx = src - 1, np.float32(0)
        out.append((sx, min(sx + 1, src - 1), fx))
    return out


def _round_half_even(v):
    return int(np.rint(np.float32(v)))                   # cvRound


def cv2_linear_scalar(img, out_h, out_w):
    """[H,W] uint8 or float32 -> [out_h,out_w]; scalar loops over OpenCV's two passes."""
    h, w = img.shape
    ax, ay = _axis_scalar(w, out_w), _axis_scalar(h, out_h)
    if img.dtype == np.uint8:
        rows = np.zeros((h, out_w), np.int64)
        for y in range(h):
            for x, (s0, s1, f) in enumerate(ax):
                a0, a1 = _round_half_even((np.float32(1) - f) * np.float32(2048)), _round_half_even(f * np.float32(2048))
                rows[y, x] = int(img[y, s0]) * a0 + int(img[y, s1]) * a1
        out = np.zeros((out_h, out_w), np.uint8)
        for y, (s0, s1, f) in enumerate(ay):
            b0, b1 = _round_half_even((np.float32(1) - f) * np.float32(2048)), _round_half_even(f * np.float32(2048))
            for x in range(out_w):
                out[y, x] = (((b0 * (int(rows[s0, x]) >> 4)) >> 16) + ((b1 * (int(rows[s1, x]) >> 4)) >> 16) + 2) >> 2
        return out
    rows = np.zeros((h, out_w), np.float32)
    for y in range(h):
        for x, (s0, s1, f) in enumerate(ax):
            rows[y, x] = np.float32(img[y, s0] * (np.float32(1) - f)) + np.float32(img[y, s1] * f)
    out = np.zeros((out_h, out_w), np.float32)
    for y, (s0, s1, f) in enumerate(ay):
        for x in range(out_w):
            out[y, x] = np.float32(rows[s0, x] * (np.float32(1) - f)) + np.float32(rows[s1, x] * f)
    return out


def test_cv2_resize_known_answers():
    """Worked by hand from the published algorithm.
    Row [10, 20] widened 2 -> 4 (scale 0.5): fx = -0.25, 0.25, 0.75, 1.25 -> taps/weights (0; 2048,0), (0,1; 1536,512),
    (0,1; 512,1536), (1; 2048,0) -> D = 20480, 25600, 35840, 40960; one source row so b = (2048, 0):
    ((2048 * (D >> 4)) >> 16) = 40, 50, 70, 80 -> (v + 2) >> 2 = 10, 13, 18, 20   (12.5 and 17.5 round UP)."""
    row = torch.tensor([[10, 20]], dtype=torch.uint8)
    assert harness.cv2_resize_linear(row, 1, 4).tolist() == [[10, 13, 18, 20]]
    # column [0, 255] heightened 2 -> 3 (scale 2/3): fy = -1/6, 0.5, 7/6 -> rows (0), (0,1; 1024,1024), (1)
    # D = v * 2048; middle: ((1024 * (0 >> 4)) >> 16) + ((1024 * (522240 >> 4)) >> 16) = 0 + 510 -> (510 + 2) >> 2 = 128
    col = torch.tensor([[0], [255]], dtype=torch.uint8)
    assert harness.cv2_resize_linear(col, 3, 1).tolist() == [[0], [128], [255]]
    # shrinking 4 -> 2 (scale 2): fx = 0.5, 2.5 -> taps (0,1; 1024,1024), (2,3; 1024,1024): plain 2-tap means, NOT an area
    # average: [0, 100, 200, 50] -> (0+100)/2 = 50, (200+50)/2 = 125
    assert harness.cv2_resize_linear(torch.tensor([[0, 100, 200, 50]], dtype=torch.uint8), 1, 2).tolist() == [[50, 125]]
    # float32 path, same geometry, no rounding: [1, 3] -> [1, 1.5, 2.5, 3]
    f = harness.cv2_resize_linear(torch.tensor([[1.0, 3.0]]), 1, 4)
    assert f.dtype == torch.float32 and f.tolist() == [[1.0, 1.5, 2.5, 3.0]]
    # same size: a copy
    a = torch.arange(12, dtype=torch.uint8).reshape(3, 4)
    r = harness.cv2_resize_linear(a, 3, 4)
    assert torch.equal(r, a) and r.data_ptr() != a.data_ptr()
    with pytest.raises(ValueError):
        harness.cv2_resize_linear(torch.zeros(3, 4, dtype=torch.float64), 6, 8)


@pytest.mark.parametrize("shape,out", [((13, 17), (64, 64)), ((436 // 4, 1024 // 8), (448 // 4, 128)), ((20, 31), (9, 14)),
                                       ((1, 5), (3, 11))])
def test_cv2_resize_vectorised_equals_scalar_statement(shape, out):
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = harness.cv2_resize_linear(torch.from_numpy(u8), *out).numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, cv2_linear_scalar(u8, *out))
    f32 = rng.normal(0, 10, size=shape).astype(np.float32)
    gotf = harness.cv2_resize_linear(torch.from_numpy(f32), *out).numpy()
    assert np.array_equal(gotf, cv2_linear_scalar(f32, *out))              # same float32 operations in the same order
    # multi-channel = per channel
    u3 = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    got3 = harness.cv2_resize_linear(torch.from_numpy(u3), *out).numpy()
    for c in range(3):
        assert np.array_equal(got3[..., c], cv2_linear_scalar(np.ascontiguousarray(u3[..., c]), *out))


def np_resize_bilinear(img, h2, w2):
    """img [H,W,C] float64 -> [h2,w2,C]: dst pixel centre (i+0.5)*H/h2 - 0.5, clamped taps (the real-valued geometry)."""
    H, W = img.shape[:2]
    ys = (np.arange(h2) + 0.5) * H / h2 - 0.5
    xs = (np.arange(w2) + 0.5) * W / w2 - 0.5
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    wy = ys - y0; wx = xs - x0
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    top = img[y0c][:, x0c] * (1 - wx)[None, :, None] + img[y0c][:, x1c] * wx[None, :, None]
    bot = img[y1c][:, x0c] * (1 - wx)[None, :, None] + img[y1c][:, x1c] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


def test_padded_size():
    assert harness.padded_size(436, 1024) == (448, 1024)      # Sintel
    assert harness.padded_size(375, 1242) == (384, 1280)      # KITTI
    assert harness.padded_size(64, 128) == (64, 128)


def test_preprocess_matches_numpy_statement():
    rng = np.random.default_rng(0)
    im1 = rng.integers(0, 256, size=(100, 150, 4), dtype=np.uint8)     # RGBA: alpha must be dropped
    im2 = rng.integers(0, 256, size=(100, 150, 4), dtype=np.uint8)
    x = harness.preprocess(torch.from_numpy(im1), torch.from_numpy(im2))
    assert x.shape == (1, 6, 128, 192) and x.dtype == torch.float32
    for k, im in enumerate((im1, im2)):
        got = x[0, 3 * k:3 * k + 3].permute(1, 2, 0).numpy()
        # exactly the fixed-point resize, channel by channel, then BGR and / 255
        for c in range(3):
            ref = cv2_linear_scalar(np.ascontiguousarray(im[:, :, 2 - c]), 128, 192).astype(np.float32) / np.float32(255.0)
            assert np.array_equal(got[:, :, c], ref)
        # and within one grey level of the real-valued bilinear interpolation (11-bit coefficients + two roundings)
        real = np_resize_bilinear(im[:, :, :3].astype(np.float64), 128, 192)[:, :, ::-1] / 255.0
        assert np.abs(got - real).max() <= 1.0 / 255 + 1e-6
    # already a multiple of 64: no resize, exact
    im = rng.integers(0, 256, size=(64, 128, 3), dtype=np.uint8)
    x = harness.preprocess(torch.from_numpy(im), torch.from_numpy(im))
    assert torch.equal(x[0, :3], torch.from_numpy(im[:, :, ::-1].copy()).permute(2, 0, 1).float() / 255.0)
    with pytest.raises(ValueError):
        harness.preprocess(torch.zeros(10, 10, 3), torch.zeros(10, 11, 3))


def test_postprocess_matches_numpy_statement():
    f2 = seeded_rand((1, 2, 32, 48), 3, -1, 1)                        # network output for a 128x192 padded input
    out = harness.postprocess(f2, 100, 150)
    assert out.shape == (100, 150, 2)
    ref = np_resize_bilinear((f2[0] * 20.0).permute(1, 2, 0).double().numpy(), 100, 150)
    ref[:, :, 0] *= 150 / 192.0
    ref[:, :, 1] *= 100 / 128.0
    assert np.abs(out.numpy() - ref).max() < 2e-4          # |flow*20| up to 20: fp32 interpolation rounding


@pytest.mark.gpu
def test_estimate_flow_end_to_end(gpu_device, tmp_path):
    from opticalflow_amd import PWCDCNet, read_flo, write_flo
    from opticalflow_amd.weights import synthetic_state_dict
    net = PWCDCNet()
    sd = synthetic_state_dict(net.manifest(), seed=0, gain=0.85, bias_std=0.02)
    net.load_state_dict(sd)
    net = net.to(gpu_device).eval()
    rng = np.random.default_rng(1)
    im1 = torch.from_numpy(rng.integers(0, 256, size=(100, 150, 3), dtype=np.uint8))
    im2 = torch.from_numpy(rng.integers(0, 256, size=(100, 150, 3), dtype=np.uint8))
    flo = harness.estimate_flow(net, im1, im2).cpu()
    assert flo.shape == (100, 150, 2)
    x = harness.preprocess(im1, im2)
    with torch.no_grad():
        ref = harness.postprocess(O.pwc_forward(sd, x), 100, 150)
    assert (flo - ref).abs().mean().item() < 1e-3
    path = str(tmp_path / "o.flo")
    write_flo(path, flo)
    assert np.array_equal(read_flo(path), flo.numpy())
