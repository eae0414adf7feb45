"""Shared plumbing for the parity tests: golden-fixture loading, recipe weights, reduced-model builders.

The builders take a namespace `ns` (the oracle module, or the product package) exposing ConvRefiner, GP, Block,
TransformerDecoder, Decoder with the constructor signatures of oracle/roma_oracle.py, so the same reduced model
can be instantiated on either side and loaded with the same recipe weights (reference state-dict key layout).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from tests.golden import cases, recipes as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def T(a, device=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if device is None else t.to(device)


def load_recipe_weights(module: nn.Module, prefix: str, seed=0, gains=None, strict=True):
    sd = module.state_dict()
    new = {k: T(R.fill_tensor(prefix + k, v.shape, seed, gains)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=strict)
    return module


def build_refiner(ns, name):
    fd, ed, r, *_ = cases.REFINER_CASES[name]
    D = 2 * fd + ed + ((2 * r + 1) ** 2 if r else 0)
    return ns.ConvRefiner(D, D, 3, hidden_blocks=2, displacement_emb_dim=ed, local_corr_radius=r).eval()


def build_reduced_decoder(ns):
    RED = cases.RED
    dd = RED["gp"] + RED["feat16"]
    td = ns.TransformerDecoder(nn.Sequential(*[ns.Block(dd, RED["heads"]) for _ in range(RED["nblk"])]), dd, RED["cls_res"] ** 2 + 1)
    feat = dict(RED["feat"])
    feat[16] = RED["feat16"]
    refiners = nn.ModuleDict()
    for s in (16, 8, 4, 2, 1):
        r = RED["rad"][s]
        D = 2 * feat[s] + RED["emb"][s] + ((2 * r + 1) ** 2 if r else 0)
        refiners[str(s)] = ns.ConvRefiner(D, D, 3, hidden_blocks=2, displacement_emb_dim=RED["emb"][s], local_corr_radius=r)
    gps = nn.ModuleDict({"16": ns.GP(RED["gp"])})
    vgg = RED["vgg"]
    dims = {"16": (RED["dino"], feat[16]), "8": (vgg[8], feat[8]), "4": (vgg[4], feat[4]), "2": (vgg[2], feat[2]), "1": (vgg[1], feat[1])}
    proj = nn.ModuleDict({s: nn.Sequential(nn.Conv2d(i, o, 1, 1), nn.BatchNorm2d(o)) for s, (i, o) in dims.items()})
    return ns.Decoder(td, gps, proj, refiners).eval()


def full_model_weights(model_state_shapes, vit_state_shapes, seed=0):
    """Recipe weights of the shipped architecture, keyed like the reference's two state dicts."""
    w = R.fill_state_dict(model_state_shapes, seed, cases.E2E_GAINS)
    v = R.fill_state_dict({"dinov2." + k: s for k, s in vit_state_shapes.items()}, seed)
    return w, {k[len("dinov2."):]: a for k, a in v.items()}


def asset(name):
    return os.path.join(GOLDEN, "assets", name)
