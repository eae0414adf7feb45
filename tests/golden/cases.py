"""Test-case definitions (shapes, deterministic inputs, reduced-model dimensions) shared by the golden-vector
generator (make_golden.py, runs the reference) and the parity tests (run the oracle / the HIP path).
Nothing here comes from the reference; it only fixes WHAT is fed to both sides."""
from __future__ import annotations

import numpy as np
import torch.nn as nn

try:
    from . import recipes as R
except ImportError:  # executed as a script directory
    import recipes as R


LOCAL_CORR_CASES = {
    # name: (B, C, h, w, r, flow kind)
    "r7_coh": (2, 32, 12, 10, 7, "coherent"),
    "r3_coh": (2, 48, 14, 18, 3, "coherent"),
    "r2_coh": (1, 24, 20, 16, 2, "coherent"),
    "r7_adv": (1, 16, 9, 11, 7, "adversarial"),
    "r3_adv": (2, 40, 13, 13, 3, "adversarial"),
    "r2_adv": (2, 8, 16, 12, 2, "adversarial"),
    "r2_none": (1, 16, 10, 10, 2, "none"),
    "r3_edge": (1, 16, 8, 8, 3, "edge"),
}


def local_corr_inputs(name):
    B, C, h, w, r, kind = LOCAL_CORR_CASES[name]
    f0 = R.normal(f"lc.{name}.f0", (B, C, h, w))
    f1 = R.normal(f"lc.{name}.f1", (B, C, h, w))
    if kind == "coherent":
        flow = R.coherent_flow(f"lc.{name}.flow", B, h, w)
    elif kind == "adversarial":
        flow = R.adversarial_flow(f"lc.{name}.flow", B, h, w)
    elif kind == "edge":  # exact pixel centres, exact borders, exactly +-1: zero bilinear fractions and half-in taps
        flow = R.identity_grid(B, h, w)
        flow[:, 0, 0, :] = -1.0
        flow[:, 1, 1, :] = 1.0
        flow[:, 0, 2, :] = 1.0 - 1.0 / w
        flow[:, :, 3, :] = 0.0
    else:
        flow = None
    return f0, f1, flow, r


def cls_inputs(res=64, B=2, H=8, W=8):
    C = res * res
    x = R.normal("cls.logits", (B, C, H, W), scale=1.0)
    # force modes onto anchor-grid borders / corners / row ends (clamp-not-wrap behaviour)
    forced = [0, res - 1, res, C - res, C - 1, res * 5 + res - 1, res * 6, (res // 4) * res + 3]
    for i, m in enumerate(forced):
        x[0, m, i // W, i % W] = 9.0 + i
    # near-tie between two far-apart classes
    x[B - 1, C // 40, 0, 0] = 8.0
    x[B - 1, (3 * C) // 4 + 1, 0, 0] = 8.0 - 1e-3
    return x


REFINER_CASES = {
    # name: (feat_dim, emb_dim, radius, b, h, w, scale_factor)
    "with_corr_r2": (16, 8, 2, 2, 12, 14, 1.0),
    "with_corr_r3": (12, 4, 3, 1, 10, 10, 1.5429),
    "no_corr": (9, 6, None, 2, 16, 12, 1.0),
}


def refiner_inputs(name):
    fd, ed, r, b, h, w, sf = REFINER_CASES[name]
    x = R.normal(f"ref.{name}.x", (b, fd, h, w))
    y = R.normal(f"ref.{name}.y", (b, fd, h, w))
    flow = R.coherent_flow(f"ref.{name}.flow", b, h, w, rot_deg=5.0, scale=0.95)
    return x, y, flow, sf


# reduced model: DINO-dim 32 @ "scale 16", VGG dims (8,16,24,32) @ scales 1,2,4,8; gp/feat dim 16; 8x8 anchor classes
RED = dict(dino=32, vgg={1: 8, 2: 16, 4: 24, 8: 32}, gp=16, feat16=16, feat={8: 16, 4: 12, 2: 8, 1: 4},
           emb={16: 8, 8: 6, 4: 4, 2: 4, 1: 2}, rad={16: 3, 8: 2, 4: 2, 2: None, 1: None}, cls_res=8, heads=2, nblk=2)


DEC_GAINS = {"to_out": 6.0, "pos_conv": 4.0}


def decoder_pyramids(tag, b, base, upsample):
    """Feature pyramids f1,f2: scale s map has size base/s (scale 16 uses base/14 like DINOv2)."""
    sizes = {1: base, 2: base // 2, 4: base // 4, 8: base // 8}
    dims = dict(RED["vgg"])
    if not upsample:
        sizes[16] = base // 14
        dims[16] = RED["dino"]
    f1 = {s: R.normal(f"dec.{tag}.f1.{s}", (b, dims[s], sizes[s], sizes[s])) for s in sizes}
    f2 = {s: R.normal(f"dec.{tag}.f2.{s}", (b, dims[s], sizes[s], sizes[s])) for s in sizes}
    return f1, f2


def post_inputs(hs=20, ws=24, up=(30, 36)):
    b = 2  # symmetric pair
    c16 = R.normal("post.c16", (b, 1, 5, 6), scale=2.0)
    flow = R.identity_grid(b, up[0], up[1]) * 1.05 + R.normal("post.flow", (b, 2, up[0], up[1]), scale=0.02)
    cert = R.normal("post.cert", (b, 1, up[0], up[1]), scale=2.0)
    return c16, flow, cert


def kde_inputs(n=512):
    return R.uniform("kde.x", (n, 4), -1, 1)


E2E_GAINS = {"to_out": 6.0, "pos_conv": 4.0}


class StubXFeat(nn.Module):
    """Interface-compatible stand-in for the (absent) XFeat backbone: the attributes TinyRoMa.forward_single
    touches (tiny.py:146-159), 24-ch stride-4 and 64-ch stride-8 outputs.  XFeat parity itself is unpinned."""

    def __init__(self):
        super().__init__()
        self.norm = nn.InstanceNorm2d(1)
        self.skip1 = nn.Sequential(nn.AvgPool2d(4, stride=4), nn.Conv2d(1, 24, 1, stride=1, padding=0))
        self.block1 = nn.Sequential(nn.Conv2d(1, 8, 3, stride=2, padding=1), nn.ReLU(), nn.Conv2d(8, 24, 3, stride=2, padding=1), nn.ReLU())
        self.block2 = nn.Sequential(nn.Conv2d(24, 24, 3, padding=1), nn.ReLU())
        self.block3 = nn.Sequential(nn.Conv2d(24, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block4 = nn.Sequential(nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block5 = nn.Sequential(nn.Conv2d(64, 64, 3, stride=2, padding=1), nn.ReLU())
        self.block_fusion = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(), nn.Conv2d(64, 64, 1))
        self.heatmap_head = nn.Identity()
        self.keypoint_head = nn.Identity()
        self.fine_matcher = nn.Identity()


