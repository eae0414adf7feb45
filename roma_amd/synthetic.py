"""Deterministic synthetic weights and image pairs for bench.py / smoke() (no datasets or checkpoints offline).
Variance-preserving random parameters (N(0, 1/fan_in) weights, non-trivial BatchNorm statistics), a pure function of
(parameter name, shape, seed) so that every rank and the CPU oracle hold bit-identical values."""
from __future__ import annotations

import math

import torch


def _rs(name: str, seed: int):
    import zlib
    import numpy as np
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synthetic_tensor(key: str, shape, seed: int = 0, gains: dict | None = None) -> torch.Tensor:
    """One state-dict entry, a function of (key, shape, seed) only (numpy RandomState: bit-stable everywhere).
    Same recipe as the parity tests use (tests/golden/recipes.py), so bench.py and the tests run the same model."""
    import numpy as np
    shape = tuple(int(s) for s in shape)
    gain = 1.0
    for pat, g in (gains or {}).items():
        if pat in key:
            gain = g
    last = key.rsplit(".", 1)[-1]
    rs = _rs(key, seed)
    if last == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if last == "running_var":
        a = rs.uniform(0.5, 1.5, size=shape)
    elif last == "running_mean":
        a = rs.standard_normal(size=shape) * 0.1
    elif last == "gamma":
        a = rs.uniform(0.5, 1.5, size=shape)
    elif last in ("cls_token", "pos_embed", "mask_token"):
        a = rs.standard_normal(size=shape) * 0.02
    elif last == "bias":
        a = rs.standard_normal(size=shape) * (0.05 * gain)
    elif len(shape) == 1:
        a = rs.uniform(0.5, 1.5, size=shape)
    else:
        a = rs.standard_normal(size=shape) * (gain / math.sqrt(float(np.prod(shape[1:]))))
    return torch.from_numpy(a.astype(np.float32))


def synthetic_state_dict(shapes: dict, seed: int = 0, gains: dict | None = None) -> dict:
    return {k: synthetic_tensor(k, s, seed, gains) for k, s in shapes.items()}


SYNTH_GAINS = {"to_out": 6.0, "pos_conv": 4.0}


def load_synthetic_weights(model, seed: int = 0):
    """Fill a roma model (product or oracle: same key layout) and its DINOv2 trunk in place."""
    vit = model.encoder.dinov2_vitl14[0]
    model.load_state_dict(synthetic_state_dict({k: v.shape for k, v in model.state_dict().items()}, seed, SYNTH_GAINS))
    vd = synthetic_state_dict({"dinov2." + k: v.shape for k, v in vit.state_dict().items()}, seed)
    vit.load_state_dict({k[len("dinov2."):]: v for k, v in vd.items()})
    return model


def _texture(g, size, octaves=6):
    """Smooth multi-scale random texture (3,size,size): a sum of bilinearly up-sampled noise octaves, unit variance."""
    import torch.nn.functional as F
    img = torch.zeros(1, 3, size, size)
    for o in range(octaves):
        n = max(4, size >> (octaves - 1 - o))
        img = img + F.interpolate(torch.randn(1, 3, n, n, generator=g), size=(size, size), mode="bilinear", align_corners=False) * (0.6 ** o)
    img = img - img.mean(dim=(2, 3), keepdim=True)
    return img / img.std(dim=(2, 3), keepdim=True)


def synthetic_pair(index: int, lo=(560, 560), hi=(864, 864), rot_deg=8.0, scale=1.08, shift=0.04):
    """One synthetic image pair with a KNOWN relation, generator seed 1234 + pair index: A is a smooth random texture
    with ImageNet-normalised statistics (zero mean, unit variance per channel), B is A seen through a similarity warp
    (rotation, scale, shift) plus 2 % sensor noise — so the matcher has real correspondences to find and the predicted
    flow is spatially coherent, as for photographs.  Returns (A_lo, B_lo, A_hi, B_hi), each (1,3,h,w) fp32."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1234 + index)
    base = max(hi[0], hi[1], lo[0], lo[1])
    A = _texture(g, base)
    c, s_ = math.cos(math.radians(rot_deg)) * scale, math.sin(math.radians(rot_deg)) * scale
    theta = torch.tensor([[[c, -s_, shift], [s_, c, -shift]]], dtype=torch.float32)
    B = F.grid_sample(A, F.affine_grid(theta, (1, 3, base, base), align_corners=False), mode="bilinear",
                      padding_mode="reflection", align_corners=False)
    B = B + 0.02 * torch.randn(B.shape, generator=g)

    def rs(img, size):
        return img if tuple(img.shape[-2:]) == tuple(size) else F.interpolate(img, size=size, mode="bilinear", align_corners=False, antialias=True)

    return rs(A, lo), rs(B, lo), rs(A, hi), rs(B, hi)
