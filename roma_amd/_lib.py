"""ctypes binding of libroma_hip.so (include/roma_hip.h).  Loading never falls back: a missing library is an error."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libroma_hip.so")

ROMA_F32, ROMA_F16, ROMA_BF16 = 0, 1, 2
ROMA_NCHW, ROMA_NHWC = 0, 1
LC_VARIANTS = {"auto": 0, "tile8x4": 1, "tile8x8": 2, "rows8": 3}
ABI_VERSION = 5
ROMA_E_ARG, ROMA_E_DTYPE, ROMA_E_SHAPE, ROMA_E_ALIGN, ROMA_E_UNSUPPORTED = -1, -2, -3, -4, -5      # include/roma_hip.h

# name -> argtypes; restype is c_int unless listed in _RESTYPES.  Mirrors include/roma_hip.h one to one.
SIGNATURES = {
    "roma_abi_version": [],
    "roma_last_error": [],
    "roma_local_corr": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_warp_bilinear": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                           c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_disp_emb": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_void_p],
    "roma_interp_bilinear": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_flow_update": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p],
    "roma_cls_to_flow_refine": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_long, c_long, c_int, c_void_p],
    "roma_cos_kernel": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                        c_float, c_void_p],
    "roma_add_layernorm": [c_void_p, c_int, c_long, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long,
                           c_long, c_int, c_float, c_void_p],
    "roma_attention_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_long] * 12 + [c_float, c_int, c_void_p],
    "roma_chol_diag_block": [c_void_p, c_int, c_long, c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_int, c_void_p],
    "roma_chol_step": [c_void_p, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_int, c_long,
                       c_void_p, c_int, c_long, c_void_p, c_int, c_int, c_void_p],
    "roma_chol_subst_step": [c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int, c_int,
                             c_void_p, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_match_finalize": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_kde_density": [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p],
    "roma_nn_argmin": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "roma_race_keys": [c_void_p, c_void_p, c_void_p, c_long, c_float, ctypes.c_uint, ctypes.c_uint, c_void_p],
    "roma_dwconv5x5_bn_relu": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_resample_u8": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "roma_normalize_u8": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "roma_bias_relu_nchw": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "roma_bias_relu_pool2_nchw": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_bias_relu_nhwc": [c_void_p, c_void_p, c_long, c_int, c_int, c_void_p],
    "roma_bias_relu_pool2_nhwc": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_pointwise_mfma": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "roma_refiner_block": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                           c_int, c_int, c_void_p],
    "roma_refiner_wide_pack": [c_void_p, c_void_p, c_int],
    "roma_refiner_wide_taps": [c_void_p, c_void_p, c_int],
    "roma_jpeg_info": [c_void_p, c_long, c_void_p],
    "roma_jpeg_entropy_decode": [c_void_p, c_long, c_void_p, c_void_p],
    "roma_jpeg_reconstruct": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "roma_refiner_block_wide": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_void_p],
    "roma_refiner_head": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_float, c_float, c_void_p],
    "roma_pointwise_small": [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p],
    "roma_tiny_corr_posembed": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
}
_RESTYPES = {"roma_last_error": c_char_p}

_lib = None


class RomaHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libroma_hip.so (once).  Raises if it has not been built — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RomaHipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C roma_amd/csrc`.  roma_amd has no CPU/PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.roma_abi_version() != ABI_VERSION:
        raise RomaHipError(f"libroma_hip.so ABI {lib.roma_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = load().roma_last_error().decode(errors="replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise RomaHipError(f"{what}: HIP error {rc}: {msg}")
