"""script_pwc.py-style harness (opticalflow_amd/harness.py): pre/post-processing pinned by an independent
numpy statement of cv2's INTER_LINEAR geometry (half-pixel centres, edge clamp, no antialias); cv2 itself is
absent from this project's environments, so its uint8 fixed-point rounding is only approximated
(round-half-up of the float interpolation) -- "parity unpinned" for that last-bit detail."""
import numpy as np
import pytest
import torch

from conftest import seeded_rand
from oracle import pwc_oracle as O
from opticalflow_amd import harness


def np_resize_bilinear(img, h2, w2):
    """img [H,W,C] float64 -> [h2,w2,C]: dst pixel centre (i+0.5)*H/h2 - 0.5, clamped taps."""
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
        ref = np.floor(np_resize_bilinear(im[:, :, :3].astype(np.float64), 128, 192) + 0.5).clip(0, 255)
        ref = ref[:, :, ::-1] / 255.0                                   # BGR, /255
        got = x[0, 3 * k:3 * k + 3].permute(1, 2, 0).numpy()
        # float32 interpolation may land on the other side of a .5 rounding boundary for a few pixels
        diff = np.abs(got - ref)
        assert (diff > 1e-6).mean() < 1e-3 and diff.max() <= 1.0 / 255 + 1e-6
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
