"""CPU-only tests: host logic of the drop-in interface and the C-ABI library's loadability.
No compute call is made (there is no GPU here); kernels are exercised by tests/test_gpu_parity.py."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from oracle import pwc_oracle as O


# ------------------------------------------------------------------ C ABI
def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pwc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pwc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from opticalflow_amd import _lib
    names = _declared_symbols()
    assert "pwc_corr_fwd" in names and "pwc_warp_fwd" in names and "pwc_conv2d_fwd" in names
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and include/pwc_hip.h disagree"
    assert os.path.exists(_lib.LIB_PATH), "libpwc_hip.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert _lib.load().pwc_abi_version() == _lib.ABI_VERSION
    assert _lib.load().pwc_experiment_mask() == 0          # the product is never a timing-experiment build (-DPWC_*_EXP: results invalid)


def test_abi_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it can be checked without a device."""
    from opticalflow_amd import _lib
    lib = _lib.load()
    assert lib.pwc_corr_fwd(None, None, None, 1, 1, 1, 1, 4, 1, 4, 1, 1, 1.0, 0, 0, 0.0, 1, 1, 1, None) == -1
    assert b"null pointer" in lib.pwc_last_error()
    assert lib.pwc_conv3x3_packed_bytes(565, 128, 0) == 71 * 8 * 9 * 128 * 4
    assert lib.pwc_conv3x3_packed_bytes(32, 2, 0) == (4 * 8 * 9 * 32 + 32 * 20) * 4      # MFMA image + [Cin][20] head tail
    assert lib.pwc_conv3x3_packed_bytes(0, 2, 0) == -1
    assert lib.pwc_conv3x3_packed_bytes(8, 8, 1) == -1          # f16 weights not supported
    # decoder level entry: the level must be exactly twice the level above (PWCNet.py:208-212 needs even H, W)
    fake = [ctypes.c_void_p(4096)] * 9
    assert lib.pwc_level_entry_c8_f16(*fake, 1, 32, 7, 16, 1.25, 0, 0.9999, *([8] * 7), None) == -1
    assert b"even" in lib.pwc_last_error()
    assert lib.pwc_level_entry_c8_f16(None, *fake[1:], 1, 32, 8, 16, 1.25, 0, 0.9999, *([8] * 7), None) == -1
    # Winograd route: filter image = [chunks of 4 cin][16 positions][2][CoutP][2] floats; argument checks; the preference rule
    assert lib.pwc_conv3x3_wino_packed_bytes(565, 128) == 142 * 64 * 128 * 4
    assert lib.pwc_conv3x3_wino_packed_bytes(5, 7) == 2 * 64 * 32 * 4 and lib.pwc_conv3x3_wino_packed_bytes(0, 7) == -1
    assert lib.pwc_conv3x3_wino_fwd(None, None, None, None, 1, 4, 8, 8, 32, 1, 0, 0.0, 1024, 2048, None, 0, None) == -1
    assert b"null pointer" in lib.pwc_last_error()
    four = [ctypes.c_void_p(4096)] * 4
    assert lib.pwc_conv3x3_wino_fwd(*four, 1, 4, 8, 8, 32, 0, 0, 0.0, 1024, 2048, None, 0, None) == -1 and b"dilation" in lib.pwc_last_error()
    assert lib.pwc_conv3x3_wino_fwd(*four, 1, 4, 8, 8, 32, 1, 0, 0.0, 8, 2048, None, 0, None) == -1 and b"batch stride" in lib.pwc_last_error()
    pref = lib.pwc_conv3x3_wino_preferred
    assert pref(16, 565, 112, 256, 128, 1) == 1 and pref(16, 128, 112, 256, 128, 4) == 1          # dc_conv1, dc_conv3
    assert pref(16, 497, 7, 16, 32, 1) == 0 and pref(32, 16, 224, 512, 16, 1) == 0                # level 6; Cout 16
    wsb = lib.pwc_conv3x3_wino_workspace_bytes
    assert pref(16, 533, 28, 64, 64, 1) == 1 and wsb(16, 533, 28, 64, 64, 1) == 3 * 16 * 64 * 28 * 64 * 4        # conv4_3: 128 tiles x 3 Cin slices
    assert pref(16, 341, 14, 32, 128, 1) == 0 and wsb(16, 341, 14, 32, 128, 1) == 0                 # level 5 (64 tiles) stays on the direct kernel
    assert wsb(16, 565, 112, 256, 128, 1) == 0                                                    # a full grid does not split
    assert pref(16, 96, 112, 256, 64, 16) == 0 and pref(1, 565, 112, 256, 128, 1) == 1            # 7x16 lattices; batch 1 at level 2
    # Winograd F(4x4,3x3) (round 3): packed size = 36 / 9 of the filter with Cout padded to 32 and Cin to 4; the measured rule; argument errors
    assert lib.pwc_conv3x3_wino4_packed_bytes(565, 128) == 142 * 36 * 4 * 128 * 4 and lib.pwc_conv3x3_wino4_packed_bytes(0, 8) == -1
    assert lib.pwc_conv3x3_wino4_packed_bytes(5, 7) == 2 * 36 * 4 * 32 * 4
    p4 = lib.pwc_conv3x3_wino4_preferred
    assert p4(16, 565, 112, 256, 128, 1) == 1 and p4(16, 373, 112, 256, 96, 1) == 1 and p4(64, 128, 56, 128, 128, 1) == 1
    assert p4(16, 128, 112, 256, 128, 2) == 0        # dilated layers: F(2x2) on lattices, or the lattice-major layout (engine)
    # launches smaller than the chip count with the input-channel slices the launcher cuts them into (round 4, option "w4_smallsplit"):
    # conv3_2's 32-cout launch (128 workgroups -> 2 slices) and dc_conv1 at batch 1 (112 workgroups -> 2 slices) are taken ...
    assert p4(16, 405, 56, 128, 96, 1) == 1 and p4(1, 565, 112, 256, 128, 1) == 1
    assert lib.pwc_conv3x3_wino4_workspace_bytes(1, 565, 112, 256, 128) == 2 * 56 * 128 * 8 * 64 * 4      # 2 slices x 56 tiles x 128 couts x 8x64 pixels
    assert p4(1, 64, 14, 32, 64, 1) == 0              # ... a launch whose slices would be shorter than 6 chunks is not
    assert lib.pwc_set_option(b"w4_smallsplit", 0) == 0
    try:                                              # option off: the round-3 rule
        assert p4(16, 405, 56, 128, 96, 1) == 0 and p4(1, 565, 112, 256, 128, 1) == 0
        assert lib.pwc_conv3x3_wino4_workspace_bytes(1, 565, 112, 256, 128) == 0
    finally:
        assert lib.pwc_set_option(b"w4_smallsplit", 1) == 0
    assert lib.pwc_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.pwc_last_error()
    assert p4(1024, 128, 14, 32, 64, 1) == 1         # narrow maps: 2 x 8 tile groups
    assert p4(16, 16, 224, 512, 16, 1) == 0 and p4(16, 64, 112, 254, 64, 1) == 0
    vp = ctypes.c_void_p
    assert lib.pwc_conv3x3_wino4_fwd(None, vp(4096), vp(4096), vp(4096), 1, 32, 8, 64, 32, 1, 0, 0.0, 32 * 512, 32 * 512, None, 0, None) == -1
    assert lib.pwc_conv3x3_wino4_fwd(vp(4096), vp(4096), vp(4096), vp(4096), 1, 32, 8, 62, 32, 1, 0, 0.0, 32 * 496, 32 * 496, None, 0, None) == -2   # W % 4
    assert lib.pwc_conv3x3_wino4_fwd(vp(4096), vp(4096), vp(4096), vp(4096), 1, 32, 8, 64, 32, 2, 0, 0.0, 32 * 512, 32 * 512, None, 0, None) == -2   # dilation
    assert lib.pwc_conv3x3_wino4_fwd(vp(4100), vp(4096), vp(4096), vp(4096), 1, 32, 8, 64, 32, 1, 0, 0.0, 32 * 512, 32 * 512, None, 0, None) == -3   # alignment
    # KITTI pre / post kernels (csrc/pwc_kitti.hip): argument checks come before any launch
    m3 = (ctypes.c_float * 3)(0.485, 0.456, 0.406)
    s3 = (ctypes.c_float * 3)(0.229, 0.224, 0.225)
    z3 = (ctypes.c_float * 3)(0.229, 0.0, 0.225)
    assert lib.pwc_kitti_ingest_u8(None, vp(4096), 1, 8, 8, m3, s3, 6 * 64 * 64, None) == -1
    assert lib.pwc_kitti_ingest_u8(vp(4096), vp(4100), 1, 8, 8, m3, s3, 6 * 64 * 64, None) == -3          # alignment
    assert lib.pwc_kitti_ingest_u8(vp(4096), vp(4096), 1, 8, 8, m3, s3, 6 * 64 * 64 - 4, None) == -3      # batch stride < 6*Hp*Wp
    assert lib.pwc_kitti_ingest_u8(vp(4096), vp(4096), 1, 8, 8, m3, z3, 6 * 64 * 64, None) == -1          # zero std
    assert lib.pwc_flow_upsample_f32(vp(4096), vp(4096), 1, 16, 16, 17, 16, 64, 64, 2 * 256, None) == -1   # crop larger than the map
    assert lib.pwc_flow_upsample_f32(vp(4096), None, 1, 16, 16, 16, 16, 64, 64, 2 * 256, None) == -1
    # tail split of the F(4x4) launches: 896 workgroups (64 couts at level 2, batch 16) = 3.5 rounds on 256 CUs -> the last 128 tiles as two
    # input-channel slices; 1792 workgroups (128 couts) = 7 whole rounds -> nothing to split; short K (8 chunks) -> no split
    wsb = lib.pwc_conv3x3_wino4_workspace_bytes
    assert wsb(16, 469, 112, 256, 64) == 2 * 128 * 64 * 8 * 64 * 4
    assert wsb(16, 565, 112, 256, 128) == 0 and wsb(32, 32, 112, 256, 32) == 0 and wsb(0, 32, 8, 64, 32) == 0
    assert lib.pwc_conv3x3_wino4_fwd(vp(4096), vp(4096), vp(4096), vp(4096), 1, 32, 7, 64, 32, 1, 32, 0.0, 32 * 448, 32 * 448, None, 0, None) == -1  # SPLIT2: odd H
    assert lib.pwc_lattice_unsplit_f32(vp(4096), vp(4096), 1, 8, 4, 5, 1, 8 * 8 * 10, None) == -3                                               # W % 4 after unsplit
    assert lib.pwc_lattice_unsplit_f32(vp(4096), vp(4096), 1, 8, 4, 6, 0, 8 * 8 * 12, None) == -1
    # split filters: 16 couts per 32-row tile -> twice the rows beyond Cout = 16, the plain size up to there (ABI v8)
    assert lib.pwc_conv3x3_f16_packed_bytes_split(64, 32) == 2 * lib.pwc_conv3x3_f16_packed_bytes(64, 32)
    assert lib.pwc_conv3x3_f16_packed_bytes_split(64, 9) == lib.pwc_conv3x3_f16_packed_bytes(64, 9)
    assert lib.pwc_conv3x3_f16_packed_bytes_split(565, 96) == 36 * 18 * 192 * 16
    assert lib.pwc_conv3x3_f16_packed_bytes_split(0, 9) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from opticalflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PwcHipError):
        _lib.load()


# ------------------------------------------------------------------ model surface
def test_state_dict_matches_reference_manifest():
    from opticalflow_amd import PWCDCNet
    g = load_golden("g5_manifest.npz")
    torch.manual_seed(0)
    net = PWCDCNet()
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert net.manifest() == O.state_dict_manifest()
    # same construction order + same init recipe as the reference -> same weights for the same seed
    sums = np.array([float(v.double().abs().sum()) for v in sd.values()])
    assert np.allclose(sums, g["seed0_abs_sums"], rtol=1e-12, atol=0)


def test_checkpoint_layouts(tmp_path):
    from opticalflow_amd import PWCDCNet, pwc_dc_net
    from opticalflow_amd.weights import load_checkpoint, synthetic_state_dict
    sd = synthetic_state_dict(PWCDCNet().manifest(), seed=3, gain=0.5, bias_std=0.1)
    layouts = {"bare": sd, "state_dict": {"state_dict": sd, "epoch": 3},
               "model": {"model": {("module." + k): v for k, v in sd.items()}, "optimizer": {}}}
    for name, obj in layouts.items():
        path = str(tmp_path / (name + ".pth.tar"))
        torch.save(obj, path)
        got = load_checkpoint(path)
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd), name
        net = pwc_dc_net(path)
        assert torch.equal(net.state_dict()["dc_conv7.bias"], sd["dc_conv7.bias"])
    bad = dict(sd)
    bad.pop("deconv2.weight")           # defined but unused by forward; still required by strict loading
    with pytest.raises(RuntimeError):
        PWCDCNet().load_state_dict(bad)


def test_cpu_tensors_are_rejected_not_silently_computed():
    from opticalflow_amd import Correlation, PWCDCNet, PwcHipError
    net = PWCDCNet().eval()
    with pytest.raises(PwcHipError):
        net(torch.zeros(1, 6, 64, 64))
    with pytest.raises(PwcHipError):
        net.warp(torch.zeros(1, 4, 8, 8), torch.zeros(1, 2, 8, 8))
    with pytest.raises(PwcHipError):
        Correlation(4, 1, 4, 1, 1)(torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 5, 64, 64))


def test_plan_constants_match_reference_channel_counts():
    from opticalflow_amd import engine
    # od per level (PWCNet.py:77,87,97,107,117) and dense growth (PWCNet.py:75)
    assert [engine.level_in_channels(l) for l in (6, 5, 4, 3, 2)] == [81, 213, 181, 149, 117]
    assert engine.DENSE_TOTAL == 448 and list(engine.DENSE_OFF) == [320, 192, 96, 32, 0]
    for l in (6, 5, 4, 3, 2):
        od = engine.level_in_channels(l)
        lo = engine.DENSE_TOTAL
        for i, (co, off) in enumerate(zip(engine.DENSE_OUT, engine.DENSE_OFF)):
            cin = engine.DENSE_TOTAL + od - lo          # channels a suffix starting at `lo` holds
            assert cin == od + sum(engine.DENSE_OUT[:i])
            assert off + co == lo                         # the new block sits right in front of its input
            lo = off
    with pytest.raises(ValueError):
        engine.PwcPlan({}, 1, 100, 128, torch.device("cpu"))


def test_correlation_shape_contract():
    from opticalflow_amd import ops
    assert ops.corr_output_shape(32, 112, 256, 4, 1, 4, 1, 1) == (81, 112, 256)
    assert ops.corr_output_shape(6, 10, 12, 4, 1, 4, 1, 2) == (25, 10, 12)
    assert ops.corr_output_shape(4, 15, 17, 3, 3, 6, 2, 2) == O.corr_output_shape(4, 15, 17, 3, 3, 6, 2, 2)
    assert ops.corr_output_shape(16, 64, 64, 20, 3, 20, 1, 2) == (441, 62, 62)   # FlowNetC-style config


def test_traceable_export_expression_matches_oracle():
    """The opt-in exporter expression (reference's USE_ONNX_CORRELATION switch) states the same cost volume."""
    import opticalflow_amd.correlation as C
    a = torch.randn(1, 5, 9, 11, generator=torch.Generator().manual_seed(1))
    b = torch.randn(1, 5, 9, 11, generator=torch.Generator().manual_seed(2))
    got = C.correlation_traceable(a, b, 4, 1, 4, 1, 1, 1)
    assert torch.allclose(got, O.correlation(a, b, 4, 1, 4, 1, 1, 1), atol=1e-5)
    assert C.USE_ONNX_CORRELATION is False


# ------------------------------------------------------------------ .flo
def test_flo_roundtrip_and_known_bytes(tmp_path):
    from opticalflow_amd import read_flo, write_flo
    g = load_golden("g4_flo.npz")
    path = str(tmp_path / "a.flo")
    write_flo(path, g["uv"])
    assert open(path, "rb").read() == g["blob"].tobytes()
    assert np.array_equal(read_flo(path), g["uv"])
    write_flo(path, torch.from_numpy(g["uv"]))
    assert np.array_equal(read_flo(path), g["uv"])
    open(path, "wb").write(b"XXXX" + g["blob"].tobytes()[4:])
    with pytest.raises(ValueError):
        read_flo(path)
    open(path, "wb").write(g["blob"].tobytes()[:-8])
    with pytest.raises(ValueError):
        read_flo(path)
    with pytest.raises(ValueError):
        write_flo(path, np.zeros((3, 5, 3), np.float32))


def test_drop_in_import_paths():
    import models
    from models.correlation_package.correlation import Correlation
    import correlation_cuda
    assert callable(models.pwc_dc_net) and callable(correlation_cuda.forward) and callable(correlation_cuda.backward)
    m = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    assert (m.pad_size, m.kernel_size, m.max_displacement, m.stride1, m.stride2, m.corr_multiply) == (4, 1, 4, 1, 1, 1)
    old = models.pwc_dc_net_old()                      # PWCNet.py:511-520
    assert type(old).__name__ == "PWCDCNet_old" and len(old.state_dict()) == 116
    # the drop-in module keeps the reference's semantics: native (normalised) unless USE_ONNX_CORRELATION is set
    assert m.normalize is True
    import inspect
    assert "normalize" not in inspect.signature(Correlation.__init__).parameters     # the reference's exact signature
    import opticalflow_amd.correlation as impl
    a = torch.randn(1, 6, 10, 12, generator=torch.Generator().manual_seed(3))
    b = torch.randn(1, 6, 10, 12, generator=torch.Generator().manual_seed(4))
    # the reference's callers set the flag on the module they import under THIS path (pth2onnx.py:44-46,
    # onnx_pth_compare.py:91-93): `corr_mod.USE_ONNX_CORRELATION = True`
    import models.correlation_package.correlation as corr_mod
    assert corr_mod.USE_ONNX_CORRELATION is False and impl.onnx_correlation_enabled() is False
    corr_mod.USE_ONNX_CORRELATION = True               # reference fallback (correlation.py:103-110): un-normalised, any device
    try:
        assert impl.onnx_correlation_enabled() is True
        assert torch.allclose(m(a, b), O.correlation(a, b, 4, 1, 4, 1, 1, 1), atol=1e-5)
        # the implementation module's own Correlation follows the same switch (CPU tensors: only the traceable expression can run)
        assert torch.allclose(impl.Correlation(4, 1, 4, 1, 1)(a, b), O.correlation(a, b, 4, 1, 4, 1, 1, 1), atol=1e-5)
        # ... and so does the net: its plans are keyed on the EFFECTIVE normalisation (PWCDCNet._normalize_now)
        from models.PWCNet import PWCDCNet as ShimNet
        net = ShimNet()
        assert net.normalize_corr is True and net._normalize_now() is False
    finally:
        corr_mod.USE_ONNX_CORRELATION = False
    assert net._normalize_now() is True
    with pytest.raises(Exception):                     # flag off: the native operator, which has no CPU path
        m(a, b)
    impl.USE_ONNX_CORRELATION = True                   # setting it on the implementation module works as well
    try:
        assert torch.allclose(m(a, b), O.correlation(a, b, 4, 1, 4, 1, 1, 1), atol=1e-5)
    finally:
        impl.USE_ONNX_CORRELATION = False


def test_models_pwcnet_classes_default_to_native_correlation():
    """the reference's scripts build the net as `from models.PWCNet import PWCDCNet; PWCDCNet();
    load_state_dict(ckpt)` (inference_kitti.py:301, inference.py:328, pwc_extract_flow.py:129): the classes exported under
    the reference's import path mirror the reference's module, whose correlation is the native one (/C); the package's
    own classes keep the parity default."""
    from models.PWCNet import PWCDCNet, PWCDCNet_old
    import models
    import opticalflow_amd as pkg
    from opticalflow_amd.weights import synthetic_state_dict
    for cls, base in ((PWCDCNet, pkg.pwcnet.PWCDCNet), (PWCDCNet_old, pkg.pwcnet.PWCDCNet_old)):
        net = cls()
        assert isinstance(net, base) and net.normalize_corr is True and net.corr.normalize is True
        net.load_state_dict(synthetic_state_dict(net.manifest(), seed=2), strict=True)
        assert cls(normalize_corr=False).normalize_corr is False           # still selectable
        assert base().normalize_corr is False
    assert models.pwc_dc_net().normalize_corr is True and type(models.pwc_dc_net()) is PWCDCNet
    assert models.pwc_dc_net_old().normalize_corr is True
    assert [k for k, _ in PWCDCNet().manifest()] == [k for k, _ in pkg.PWCDCNet().manifest()]


def test_checkpoint_factories_default_to_native_correlation(tmp_path):
    """a checkpoint is trained against the reference's native correlation (/C, correlation_cuda_kernel.cu:104,143),
    so pwc_dc_net(path) defaults to normalize_corr=True; without a file the constructor's parity default (un-normalised)
    stays; an explicit normalize_corr=False with a checkpoint warns."""
    import warnings
    import models
    from opticalflow_amd import PWCDCNet, pwc_dc_net, pwc_dc_net_old
    from opticalflow_amd.weights import synthetic_state_dict
    path = str(tmp_path / "w.pth.tar")
    torch.save(synthetic_state_dict(PWCDCNet().manifest(), seed=1), path)
    assert PWCDCNet().normalize_corr is False and pwc_dc_net().normalize_corr is False
    for factory in (pwc_dc_net, models.pwc_dc_net):
        net = factory(path)
        assert net.normalize_corr is True and net.corr.normalize is True
    assert pwc_dc_net(path, normalize_corr=True, align_corners=True).align_corners is True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net = pwc_dc_net(path, normalize_corr=False)
        assert net.normalize_corr is False and any("normalize_corr=False" in str(x.message) for x in w)
    from opticalflow_amd import pwcnet
    path_old = str(tmp_path / "old.pth.tar")
    torch.save(synthetic_state_dict(pwcnet.PWCDCNet_old().manifest(), seed=1), path_old)
    assert pwc_dc_net_old(path_old).normalize_corr is True


def test_old_variant_state_dict_and_channel_permutation():
    """PWCDCNet_old: the reference's 116 keys in its order (golden g6), strict load; the filter re-mapping
    used by the plan is a true permutation that sends the reference's concatenation order to the arena's."""
    from conftest import load_golden
    from opticalflow_amd import pwcnet
    from opticalflow_amd.engine import DENSE_OUT, old_variant_perm
    g = load_golden("g6_old.npz")
    net = pwcnet.PWCDCNet_old()
    assert [k for k, _ in net.manifest()] == [str(k) for k in g["keys"]]
    assert [",".join(map(str, s)) for _, s in net.manifest()] == [str(s) for s in g["shapes"]]
    net.load_state_dict({k: torch.zeros(s) for k, s in net.manifest()}, strict=True)
    od = 81 + 32 + 4
    for k in range(1, 6):
        perm = old_variant_perm(k, od)
        n = od + sum(DENSE_OUT[:k])
        assert sorted(perm.tolist()) == list(range(n))
    # k=2: reference order [conv_1 | base | conv_0] -> arena [conv_1 | conv_0 | base]
    p2 = old_variant_perm(2, od).tolist()
    assert p2[:128] == list(range(0, 128)) and p2[128:256] == list(range(128 + od, 256 + od)) and p2[256:] == list(range(128, 128 + od))
    # functional check on CPU: conv over the arena order with permuted filters == conv over the reference order
    xs = {n: torch.randn(1, c, 5, 6) for n, c in (("base", od), ("c0", 128), ("c1", 128), ("c2", 96))}
    w = torch.randn(8, od + 128 + 128 + 96, 3, 3)
    ref = torch.nn.functional.conv2d(torch.cat([xs["c1"], xs["base"], xs["c0"], xs["c2"]], 1), w, padding=1)
    got = torch.nn.functional.conv2d(torch.cat([xs["c2"], xs["c1"], xs["c0"], xs["base"]], 1),
                                     w.index_select(1, old_variant_perm(3, od)), padding=1)
    assert (ref - got).abs().max().item() < 1e-3


def test_bench_self_launches_its_ranks(monkeypatch):
    """VERDICT r2 missing #2: `python bench.py --gpus N` (no WORLD_SIZE) starts torch.distributed.run as a CHILD process with the
    driver's flags and the same arguments, and exits with the child's code; under torch.distributed.run it does not re-launch."""
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        """first launch: dies before any rank prints (a rendezvous failure: the port was taken) -> ONE retry with a new port;
        second launch: a rank printed, exit code 7 is passed on"""
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen.setdefault("cmds", []).append(cmd)
            seen["cmd"], seen["env"] = cmd, env
            self.returncode = 1 if len(seen["cmds"]) == 1 else 7

        def communicate(self):
            return ("" if len(seen["cmds"]) == 1 else '{"metric": "x"}\n'), None

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "kitti"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7 and len(seen["cmds"]) == 2
    a, b = seen["cmds"]
    ip = a.index("--master-port") + 1
    assert a[:ip] + a[ip + 1:] == b[:ip] + b[ip + 1:]           # the same command; the port may differ
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "kitti"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # a rank whose WORLD_SIZE disagrees with --gpus is an error, not a second launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    seen.clear()
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert not seen and "WORLD_SIZE=2" in str(ei.value.code)


def test_strict_mode_default_filter_policy(monkeypatch):
    """The strict half mode's default since round 4: split (hi + lo) filters on EVERY level-2 / context layer (the error of the mode is
    relative to the flow magnitude, so the faster ladder step's 13 % margin was a property of the test inputs); PWC_STRICT_PLAIN=fast
    opts into the measured ladder step, a list names layers, an unknown name is an error.  Ladder: engine_strict.py, DESIGN.md 7a."""
    import importlib
    monkeypatch.delenv("PWC_STRICT_PLAIN", raising=False)
    from opticalflow_amd import engine_strict
    importlib.reload(engine_strict)
    assert engine_strict.PLAIN_FILTERS == frozenset()
    monkeypatch.setenv("PWC_STRICT_PLAIN", "none")
    importlib.reload(engine_strict)
    assert engine_strict.PLAIN_FILTERS == frozenset()
    monkeypatch.setenv("PWC_STRICT_PLAIN", "fast")
    importlib.reload(engine_strict)
    assert engine_strict.PLAIN_FILTERS == {"dc_conv4", "dc_conv5", "dc_conv6", "conv2_3", "conv2_4"}
    monkeypatch.setenv("PWC_STRICT_PLAIN", "dc_conv6")
    importlib.reload(engine_strict)
    assert engine_strict.PLAIN_FILTERS == {"dc_conv6"}
    monkeypatch.setenv("PWC_STRICT_PLAIN", "dc_conv6,conv9_9")
    with pytest.raises(ValueError, match="conv9_9"):
        importlib.reload(engine_strict)
    monkeypatch.delenv("PWC_STRICT_PLAIN")
    importlib.reload(engine_strict)


def test_deconv_as_conv3x3_is_the_transposed_convolution():
    """ops.deconv_as_conv3x3: ConvTranspose2d(k4, s2, p1) (upfeatL / deconvL, models/PWCNet.py:35-36) restated as a 3x3 convolution with
    four output phases per channel + pixel shuffle -- the form the small levels run on the matrix cores together with the flow head
    (pwc_upsample_entry_f32 does the shuffle).  Pure host arithmetic, checked against torch's conv_transpose2d in fp64."""
    import torch.nn.functional as F
    from opticalflow_amd import ops
    g = torch.Generator().manual_seed(3)
    for cin, cout, h, w in ((5, 2, 7, 16), (3, 3, 4, 5), (8, 2, 1, 1)):
        x = torch.randn(2, cin, h, w, generator=g, dtype=torch.float64)
        wt = torch.randn(cin, cout, 4, 4, generator=g, dtype=torch.float64)
        b = torch.randn(cout, generator=g, dtype=torch.float64)
        ref = F.conv_transpose2d(x, wt, b, stride=2, padding=1)
        k = ops.deconv_as_conv3x3(wt)                                     # [cout*4, cin, 3, 3], channel co*4 + py*2 + px
        assert tuple(k.shape) == (cout * 4, cin, 3, 3)
        ph = F.conv2d(x, k, b.repeat_interleave(4), padding=1)            # [B, cout*4, h, w]
        got = F.pixel_shuffle(ph, 2)                                       # channel co*4 + py*2 + px -> (co, 2y+py, 2x+px)
        assert got.shape == ref.shape and (got - ref).abs().max().item() < 1e-12
