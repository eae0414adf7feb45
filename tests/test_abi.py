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
    assert lib.roma_abi_version() == 4
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


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from roma_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.kde(torch.zeros(8, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cls_to_flow_refine(torch.zeros(1, 64, 2, 2))
