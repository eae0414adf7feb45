"""Deterministic input / weight recipes shared by the golden-vector generator and the tests.

Everything here is numpy ``RandomState`` based (bit-stable across numpy versions and machines), so a
fixture only has to store the *outputs* the reference produced; the inputs are regenerated from the
recipe on whichever box the test runs.  Nothing in this file comes from the reference: it is data
plumbing for the parity tests.
"""
from __future__ import annotations

import math
import zlib

import numpy as np


def rs_for(name: str, seed: int = 0) -> np.random.RandomState:
    """A RandomState that depends only on (name, seed), not on call order."""
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def normal(name: str, shape, seed: int = 0, scale: float = 1.0) -> np.ndarray:
    return (rs_for(name, seed).standard_normal(size=shape) * scale).astype(np.float32)


def uniform(name: str, shape, lo: float, hi: float, seed: int = 0) -> np.ndarray:
    return rs_for(name, seed).uniform(lo, hi, size=shape).astype(np.float32)


def identity_grid(b: int, h: int, w: int) -> np.ndarray:
    """(b,2,h,w) grid of pixel centres in [-1,1], channel 0 = x, channel 1 = y."""
    ys = np.linspace(-1 + 1 / h, 1 - 1 / h, h, dtype=np.float32)
    xs = np.linspace(-1 + 1 / w, 1 - 1 / w, w, dtype=np.float32)
    g = np.stack(np.broadcast_arrays(xs[None, :], ys[:, None]), axis=0)
    return np.broadcast_to(g[None], (b, 2, h, w)).astype(np.float32).copy()


def coherent_flow(name: str, b: int, h: int, w: int, seed: int = 0, rot_deg: float = 10.0,
                  scale: float = 1.1, shift: float = 0.05, jitter_px: float = 0.5) -> np.ndarray:
    """identity grid ∘ affine (rotation, scale, shift) + N(0, jitter_px²) — the headline flow of SURVEY §8(d)."""
    g = identity_grid(b, h, w)
    c, s = math.cos(math.radians(rot_deg)) * scale, math.sin(math.radians(rot_deg)) * scale
    x = c * g[:, 0] - s * g[:, 1] + shift
    y = s * g[:, 0] + c * g[:, 1] - shift
    f = np.stack((x, y), axis=1)
    jit = rs_for(name, seed).standard_normal(size=f.shape).astype(np.float32)
    jit[:, 0] *= jitter_px * 2 / w
    jit[:, 1] *= jitter_px * 2 / h
    return (f + jit).astype(np.float32)


def adversarial_flow(name: str, b: int, h: int, w: int, seed: int = 0, lim: float = 1.2) -> np.ndarray:
    """U(-lim, lim): incoherent targets, partly outside the image (zero padding path)."""
    return uniform(name, (b, 2, h, w), -lim, lim, seed)


# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------

def fill_tensor(key: str, shape, seed: int = 0, gains: dict | None = None) -> np.ndarray:
    """Deterministic value for one state-dict entry, chosen by the entry's name and rank."""
    shape = tuple(int(s) for s in shape)
    gain = 1.0
    for pat, g in (gains or {}).items():
        if pat in key:
            gain = g
    last = key.rsplit(".", 1)[-1]
    if last == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if last == "running_var":
        return uniform(key, shape, 0.5, 1.5, seed)
    if last == "running_mean":
        return normal(key, shape, seed, 0.1)
    if last == "gamma":  # LayerScale
        return uniform(key, shape, 0.5, 1.5, seed)
    if last in ("cls_token", "pos_embed", "mask_token"):
        return normal(key, shape, seed, 0.02)
    if last == "bias":
        return normal(key, shape, seed, 0.05 * gain)
    if len(shape) == 1:  # norm scale
        return uniform(key, shape, 0.5, 1.5, seed)
    fan_in = int(np.prod(shape[1:]))
    return normal(key, shape, seed, gain / math.sqrt(fan_in))


def fill_state_dict(shapes: dict, seed: int = 0, gains: dict | None = None) -> dict:
    """shapes: {key: shape}.  Returns {key: ndarray}; each entry depends only on (key, shape, seed)."""
    return {k: fill_tensor(k, s, seed, gains) for k, s in shapes.items()}


def checksum(a: np.ndarray) -> np.ndarray:
    """A few order-independent statistics used to pin big tensors without storing them."""
    a64 = np.asarray(a, dtype=np.float64)
    return np.array([a64.sum(), np.abs(a64).sum(), (a64 * a64).sum(), a64.min(), a64.max()], dtype=np.float64)
