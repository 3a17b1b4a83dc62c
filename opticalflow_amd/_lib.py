"""ctypes binding of libpwc_hip.so (the C ABI declared in include/pwc_hip.h).

The library is the product: if it is missing or fails to load, every operator
raises -- there is no CPU or eager-PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# PWC_HIP_LIB: alternative build of the same C ABI (kernel experiments); default = the in-tree library
LIB_PATH = os.environ.get("PWC_HIP_LIB") or os.path.join(_HERE, "libpwc_hip.so")

ABI_VERSION = 12
PWC_F32, PWC_F16 = 0, 1
FLAG_CORR_NORMALIZE = 1
FLAG_ACT_LEAKY = 2
FLAG_CONV_RESIDUAL = 4
FLAG_CONV_OUT_F32 = 8
FLAG_CONV_SPLIT_W = 16
FLAG_CONV_SPLIT2 = 32

# name -> (restype, argtypes); mirrors include/pwc_hip.h one to one
SIGNATURES = {
    "pwc_abi_version": (c_int, []),
    "pwc_experiment_mask": (c_int, []),
    "pwc_last_error": (c_char_p, []),
    "pwc_last_conv_kernel": (c_char_p, []),
    "pwc_set_option": (c_int, [c_char_p, c_int]),
    "pwc_get_option": (c_int, [c_char_p, ctypes.POINTER(c_int)]),
    "pwc_corr_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_uint, c_float,
                             c_int64, c_int64, c_int64, c_void_p]),
    "pwc_warp_corr81_preferred": (c_int, [c_int, c_int, c_int, c_int]),
    "pwc_warp_corr81_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_float,
                                    c_float, c_uint, c_float, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "pwc_corr_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_uint, c_void_p]),
    "pwc_warp_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_float, c_int, c_float, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "pwc_warp_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "pwc_warp_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_float, c_int, c_float, c_int, c_void_p, c_int64, c_void_p]),
    "pwc_conv3x3_packed_bytes": (c_int64, [c_int, c_int, c_int]),
    "pwc_conv3x3_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pwc_conv2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_float,
                               c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "pwc_conv2d_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pwc_conv3x3_f16_packed_bytes": (c_int64, [c_int, c_int]),
    "pwc_conv3x3_f16_packed_bytes_split": (c_int64, [c_int, c_int]),
    "pwc_conv3x3_f16_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pwc_conv3x3_f16_pack_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pwc_conv3x3_wino_packed_bytes": (c_int64, [c_int, c_int]),
    "pwc_conv3x3_wino_preferred": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pwc_conv3x3_wino_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pwc_conv3x3_wino_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_float,
                             c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "pwc_conv3x3_wino_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pwc_conv3x3_wino4_packed_bytes": (c_int64, [c_int, c_int]),
    "pwc_conv3x3_wino4_preferred": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "pwc_conv3x3_wino4_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "pwc_conv3x3_wino4_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_float,
                                     c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "pwc_conv3x3_wino4_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "pwc_kitti_ingest_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "pwc_flow_upsample_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "pwc_lattice_unsplit_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "pwc_conv2d_f16_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_uint, c_float, c_int64, c_int64, c_void_p]),
    "pwc_nchw_to_c8_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "pwc_nchw_to_c8_f16_hilo": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "pwc_c8_f16_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "pwc_image_conv_s2_c8_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                         c_int64, c_int64, c_void_p]),
    "pwc_pyramid1_f16_packed_bytes": (c_int64, []),
    "pwc_pyramid1_fused_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                       c_int64, c_int64, c_void_p]),
    "pwc_corr81_c8_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_uint, c_float,
                                  c_int64, c_int64, c_int64, c_void_p]),
    "pwc_warp_c8_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float,
                                c_int64, c_int64, c_int64, c_void_p]),
    "pwc_level_entry_c8_f16": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_float, c_int, c_float] + [c_int64] * 7 + [c_void_p]),
    "pwc_level_corr81_c8_f16": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_float, c_uint, c_float]
                                + [c_int64] * 7 + [c_void_p]),
    "pwc_calib_lds_dma_read": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "pwc_upsample_entry_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "pwc_deconv4x4s2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p]),
    "pwc_head_upfeat_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "pwc_head_upfeat_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "pwc_head_upfeat_ws_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
}

_lib = None


class PwcHipError(RuntimeError):
    """Raised when libpwc_hip.so is unavailable or a call returns non-zero."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PwcHipError(
            "libpwc_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C opticalflow_amd/csrc`; the HIP path has no fallback" % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise PwcHipError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PwcHipError("libpwc_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    got = lib.pwc_abi_version()
    if got != ABI_VERSION:
        raise PwcHipError("libpwc_hip.so ABI %d, binding expects %d -- rebuild" % (got, ABI_VERSION))
    exp = lib.pwc_experiment_mask()
    if exp and not os.environ.get("PWC_HIP_LIB"):
        # a timing-experiment build (-DPWC_*_EXP: work skipped, results invalid) must never pass as the product
        raise PwcHipError("%s was built with timing-experiment switches (mask %d): results would be invalid -- rebuild it "
                          "(make -C opticalflow_amd/csrc) or select an experiment library explicitly with PWC_HIP_LIB" % (LIB_PATH, exp))
    _lib = lib
    return lib


def set_option(name: str, value: int) -> None:
    """pwc_set_option: flip a kernel-selection switch of the library at run time (see include/pwc_hip.h for the names)."""
    check(load().pwc_set_option(name.encode(), int(value)), "pwc_set_option(%s)" % name)


def get_option(name: str) -> int:
    v = c_int(0)
    check(load().pwc_get_option(name.encode(), ctypes.byref(v)), "pwc_get_option(%s)" % name)
    return v.value


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pwc_last_error()
        raise PwcHipError("%s failed (code %d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else "?"))
