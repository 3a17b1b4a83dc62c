"""Video frame-pair loop (pwc_extract_flow_video.py:192-305): pyramid-reusing FlowStream vs. the per-pair forward."""
import numpy as np
import pytest
import torch

from conftest import seeded_rand


def test_frame_to_tensor_bgr_to_rgb_unit_range():
    from opticalflow_amd.video import frame_to_tensor
    f = np.zeros((5, 7, 3), np.uint8)
    f[..., 0], f[..., 1], f[..., 2] = 255, 128, 0          # B, G, R
    t = frame_to_tensor(f)
    assert t.shape == (3, 5, 7) and t.dtype == torch.float32
    assert t[0].max().item() == 0.0 and abs(t[1, 0, 0].item() - 128 / 255.0) < 1e-7 and t[2].min().item() == 1.0
    with pytest.raises(ValueError):
        frame_to_tensor(np.zeros((5, 7), np.uint8))


def test_flow_stream_needs_device_model():
    from opticalflow_amd import PWCDCNet, PwcHipError
    from opticalflow_amd.video import FlowStream
    with pytest.raises(PwcHipError):
        FlowStream(PWCDCNet(), 1, 64, 64)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,use_graph", [(1, False), (1, True), (3, True)])
def test_flow_stream_matches_pairwise_forward(gpu_device, batch, use_graph):
    """Same numbers as model(cat(frame_t, frame_t+1)) for every consecutive pair.  Tolerance: the pair-wise
    forward runs the pyramid at batch 2B and the stream at batch B, which can select a different conv tile
    (different fp32 summation order) -- 1e-4 px/20 absolute on flows of O(1)."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.video import FlowStream
    from opticalflow_amd.weights import synthetic_state_dict
    dev = gpu_device
    net = PWCDCNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=3, gain=0.85, bias_std=0.02))
    H, W, n = 128, 192, 1 + 2 * batch
    frames = seeded_rand((n, 3, H, W), 77, 0, 1).to(dev)
    stream = FlowStream(net, batch, H, W, use_graph=use_graph)
    with pytest.raises(RuntimeError):
        stream.push(frames[1:1 + batch])
    stream.prime(frames[0])
    got = [stream.push(frames[1 + k * batch:1 + (k + 1) * batch]).clone() for k in range(2)]
    got = torch.cat(got, 0)
    ref = net(torch.cat([frames[:-1], frames[1:]], 1))
    assert got.shape == ref.shape == (n - 1, 2, H // 4, W // 4)
    assert ref.abs().max().item() > 1e-2
    assert (got - ref).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_flow_stream_fp16(gpu_device):
    """the half-precision stream plan reuses the pyramid too and agrees with the fp16 per-pair forward."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.video import FlowStream
    from opticalflow_amd.weights import synthetic_state_dict
    dev = gpu_device
    net = PWCDCNet(precision="fp16").to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=3, gain=0.85, bias_std=0.02))
    H, W, batch = 128, 192, 2
    frames = seeded_rand((1 + 2 * batch, 3, H, W), 78, 0, 1).to(dev)
    stream = FlowStream(net, batch, H, W, use_graph=True)
    stream.prime(frames[0])
    got = torch.cat([stream.push(frames[1 + k * batch:1 + (k + 1) * batch]).clone() for k in range(2)], 0)
    ref = net(torch.cat([frames[:-1], frames[1:]], 1))
    assert got.shape == ref.shape and ref.abs().max().item() > 1e-2
    # same kernels on the same half-precision features; only the pyramid's batch size (tile choice) may differ
    assert (got - ref).abs().max().item() < 2e-2 * ref.abs().max().item()


@pytest.mark.gpu
def test_flow_video_generator_matches_reference_loop(gpu_device):
    """flow_video() == the reference loop's process_frame_pair on each pair, incl. its pad/unpad quirk."""
    from opticalflow_amd import PWCDCNet
    from opticalflow_amd.kitti import pad_to_64
    from opticalflow_amd.video import flow_video, frame_to_tensor
    from opticalflow_amd.weights import synthetic_state_dict
    dev = gpu_device
    net = PWCDCNet().to(dev).eval()
    net.load_state_dict(synthetic_state_dict(net.manifest(), seed=4, gain=0.85, bias_std=0.02))
    rng = np.random.RandomState(5)
    frames = [rng.randint(0, 256, (100, 150, 3)).astype(np.uint8) for _ in range(4)]
    flows = list(flow_video(net, frames))
    assert len(flows) == 3
    for i, fl in enumerate(flows):
        t1, ph, pw = pad_to_64(frame_to_tensor(frames[i]).unsqueeze(0))
        t2, _, _ = pad_to_64(frame_to_tensor(frames[i + 1]).unsqueeze(0))
        ref = net(torch.cat([t1, t2], 1).to(dev))
        ref = ref[:, :, :ref.shape[2] - ph, :ref.shape[3] - pw]        # pwc_extract_flow_video.py:44-47,213
        ref = ref.squeeze(0).permute(1, 2, 0).cpu().numpy()
        assert fl.shape == ref.shape and fl.dtype == np.float32
        assert np.abs(fl - ref).max() < 1e-4
