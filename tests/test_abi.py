"""The C-ABI library loads on a CPU-only box and exports every symbol include/roma_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from roma_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "roma_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(roma_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.roma_abi_version() == 5
    assert isinstance(lib.roma_last_error(), bytes)


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    rc = lib.roma_local_corr(None, None, None, None, 1, 8, 4, 4, 2, 0, 0, 8, 8, 0, 25, 0, 0, None)
    assert rc == -1 and b"null pointer" in lib.roma_last_error()
    rc = lib.roma_kde_density(None, None, 0, 1, 0.1, 0, None)
    assert rc < 0
    rc = lib.roma_add_layernorm(None, 0, 8, None, 0, 8, None, None, None, None, 0, 8, 1, 8, 1e-5, None)
    assert rc == -1 and b"null pointer" in lib.roma_last_error()
    rc = lib.roma_race_keys(None, None, None, 4, 0.05, 1, 0, None)
    assert rc == -1
    # the fused Cholesky solve (round 2): null pointers, then shapes, are rejected before anything touches a device
    rc = lib.roma_chol_step(None, 8, 64, 8, 8, 0, 8, None, 8, 64, None, 8, 64, None, 8, 64, None, 0, 1, None)
    assert rc == -1 and b"roma_chol_step: null pointer" in lib.roma_last_error()
    buf = (ctypes.c_float * 4)()
    a = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.roma_chol_step(a, 8, 64, 8, 8, 0, 65, a, 8, 64, a, 8, 64, None, 8, 64, a, 0, 1, None)     # nb > 64
    assert rc < 0 and b"bad shape" in lib.roma_last_error()
    rc = lib.roma_chol_subst_step(-1, None, 8, 64, None, 64, 64, 8, None, 64, 64, 8, 1, None, 8, 64, 8, 8, 8, 0, 1, None)
    assert rc == -1 and b"roma_chol_subst_step: null pointer" in lib.roma_last_error()
    rc = lib.roma_chol_subst_step(0, a, 8, 64, a, 64, 64, 8, a, 64, 64, 8, 1, a, 8, 64, 8, 8, 8, 0, 1, None)   # dir == 0
    assert rc < 0 and b"bad shape" in lib.roma_last_error()
    rc = lib.roma_chol_subst_step(1, a, 32, 1024, a, 64, 64, 200, a, 64, 64, 200, 0, a, 8, 64, 100, 8, 32, 1, 1, None)   # 32-row blocks, 4 of them
    assert rc < 0 and b"64-row blocks" in lib.roma_last_error()
    rc = lib.roma_dwconv5x5_bn_relu(a, a, a, a, a, 1, 12, 4, 4, 1, 12, 12, None)                      # C not a multiple of 8
    assert rc < 0 and b"multiples of 8" in lib.roma_last_error()


def test_refiner_wide_host_packers_lay_out_what_the_header_says():
    """roma_refiner_wide_taps / roma_refiner_wide_pack are HOST functions (no device): their outputs against the layouts written in
    include/roma_hip.h, restated here with numpy."""
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(5)
    D = 64
    w25 = rng.integers(1, 60000, size=(25, D)).astype(np.uint16)                 # opaque 16-bit patterns: the packers only move them
    out = np.zeros(60 * D, dtype=np.uint16)
    assert lib.roma_refiner_wide_taps(w25.ctypes.data, out.ctypes.data, D) == 0
    got = out.reshape(D // 32, 5, 4, 2, 6, 4, 2)
    lo, hi = (0, 2, 4, None, 1, 3), (1, 3, None, 0, 2, 4)                        # tap column of (low half, high half) per pair set
    for kp in range(D // 32):
        for dy in range(5):
            for kq in range(4):
                for h in range(2):
                    c0 = kp * 32 + kq * 8 + h * 4
                    for st in range(6):
                        want_lo = np.zeros(4, np.uint16) if lo[st] is None else w25[dy * 5 + lo[st], c0:c0 + 4]
                        want_hi = np.zeros(4, np.uint16) if hi[st] is None else w25[dy * 5 + hi[st], c0:c0 + 4]
                        assert (got[kp, dy, kq, h, st, :, 0] == want_lo).all() and (got[kp, dy, kq, h, st, :, 1] == want_hi).all()
    assert lib.roma_refiner_wide_taps(w25.ctypes.data, out.ctypes.data, 48) < 0 and b"multiple of 32" in lib.roma_last_error()
    wt = rng.integers(0, 60000, size=(D, D)).astype(np.uint16)
    wp = np.zeros(D * D, dtype=np.uint16)
    assert lib.roma_refiner_wide_pack(wt.ctypes.data, wp.ctypes.data, D) == 0
    gotp = wp.reshape(D // 32, D, 4, 8)
    for kp in range(D // 32):
        for n in range(D):
            for slot in range(4):
                kgl = slot ^ ((n >> 1) & 3)
                assert (gotp[kp, n, slot] == wt[n, kp * 32 + kgl * 8: kp * 32 + kgl * 8 + 8]).all()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from roma_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.kde(torch.zeros(8, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cls_to_flow_refine(torch.zeros(1, 64, 2, 2))
