"""Model zoo with the reference's entry points and hyper-parameters — romatch/models/model_zoo/__init__.py:16-73,
roma_models.py:9-170.  Weights are NOT bundled; like the reference, omitted weights are fetched with torch.hub
(which needs network access) — pass `weights=` / `dinov2_weights=` state dicts when offline."""
from __future__ import annotations

import warnings
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .encoders import CNNandDinov2
from .matcher import ConvRefiner, Decoder, GP, RegressionMatcher
from .transformer import Block, DinoViT, TransformerDecoder

weight_urls = {
    "romatch": {
        "outdoor": "https://github.com/Parskatt/storage/releases/download/roma/roma_outdoor.pth",
        "indoor": "https://github.com/Parskatt/storage/releases/download/roma/roma_indoor.pth",
    },
    "tiny_roma_v1": {"outdoor": "https://github.com/Parskatt/storage/releases/download/roma/tiny_roma_v1_outdoor.pth"},
    "dinov2": "https://dl.fbaipublicfiles.com/dinov2/dinov2_vitl14/dinov2_vitl14_pretrain.pth",
}

# scale -> (projected feature dim, displacement-embedding dim, local-correlation radius)   roma_models.py:40-114
REFINER_SPEC = {"16": (512, 128, 7), "8": (512, 64, 3), "4": (256, 32, 2), "2": (64, 16, None), "1": (9, 6, None)}
PROJ_SPEC = {"16": (1024, 512), "8": (512, 512), "4": (256, 256), "2": (128, 64), "1": (64, 9)}   # roma_models.py:131-142


def build_roma(resolution=(560, 560), upsample_preds=True, amp_dtype=torch.float16, dinov2: Optional[DinoViT] = None):
    """The shipped architecture, randomly initialised (no weights touched)."""
    gp_dim, feat_dim, cls_res = 512, 512, 64
    dd = gp_dim + feat_dim
    td = TransformerDecoder(nn.Sequential(*[Block(dd, 8) for _ in range(5)]), dd, cls_res ** 2 + 1, is_classifier=True)
    refiners = nn.ModuleDict()
    for s, (fd, ed, r) in REFINER_SPEC.items():
        D = 2 * fd + ed + ((2 * r + 1) ** 2 if r else 0)
        refiners[s] = ConvRefiner(D, D, 2 + 1, hidden_blocks=8, displacement_emb_dim=ed, local_corr_radius=r, amp_dtype=amp_dtype)
    gps = nn.ModuleDict({"16": GP(gp_dim, T=0.2)})
    proj = nn.ModuleDict({s: nn.Sequential(nn.Conv2d(i, o, 1, 1), nn.BatchNorm2d(o)) for s, (i, o) in PROJ_SPEC.items()})
    decoder = Decoder(td, gps, proj, refiners, detach=True, scales=["16", "8", "4", "2", "1"], amp_dtype=amp_dtype)
    encoder = CNNandDinov2(dinov2, amp_dtype=amp_dtype)
    h, w = resolution
    return RegressionMatcher(encoder, decoder, h=h, w=w, upsample_preds=upsample_preds, symmetric=True, attenuate_cert=True,
                             sample_mode="threshold_balanced")


def roma_model(resolution, upsample_preds, device=None, weights=None, dinov2_weights=None, amp_dtype: torch.dtype = torch.float16, **kwargs):
    """roma_models.py:18-170: build, move to `device`, load both state dicts (strict)."""
    warnings.filterwarnings("ignore", category=UserWarning, message="TypedStorage is deprecated")
    vit = DinoViT()
    if dinov2_weights is not None:
        vit.load_state_dict(dinov2_weights)
    model = build_roma(resolution, upsample_preds, amp_dtype, vit.eval()).to(device)
    model.load_state_dict(weights)
    return model.eval()


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _roma(kind, device, weights, dinov2_weights, coarse_res, upsample_res, amp_dtype):
    coarse_res, upsample_res = _pair(coarse_res), _pair(upsample_res)
    assert coarse_res[0] % 14 == 0, "Needs to be multiple of 14 for backbone"
    assert coarse_res[1] % 14 == 0, "Needs to be multiple of 14 for backbone"
    if weights is None:
        weights = torch.hub.load_state_dict_from_url(weight_urls["romatch"][kind], map_location=device)
    if dinov2_weights is None:
        dinov2_weights = torch.hub.load_state_dict_from_url(weight_urls["dinov2"], map_location=device)
    model = roma_model(resolution=coarse_res, upsample_preds=True, weights=weights, dinov2_weights=dinov2_weights,
                       device=device, amp_dtype=amp_dtype)
    model.upsample_res = upsample_res
    print(f"Using coarse resolution {coarse_res}, and upsample res {model.upsample_res}")
    return model


def roma_outdoor(device, weights=None, dinov2_weights=None, coarse_res: Union[int, Tuple[int, int]] = 560,
                 upsample_res: Union[int, Tuple[int, int]] = 864, amp_dtype: torch.dtype = torch.float16):
    """model_zoo/__init__.py:30-52."""
    return _roma("outdoor", device, weights, dinov2_weights, coarse_res, upsample_res, amp_dtype)


def roma_indoor(device, weights=None, dinov2_weights=None, coarse_res: Union[int, Tuple[int, int]] = 560,
                upsample_res: Union[int, Tuple[int, int]] = 864, amp_dtype: torch.dtype = torch.float16):
    """model_zoo/__init__.py:54-73."""
    return _roma("indoor", device, weights, dinov2_weights, coarse_res, upsample_res, amp_dtype)


def tiny_roma_v1_model(weights=None, freeze_xfeat=False, exact_softmax=False, xfeat=None):
    from .tiny import TinyRoMa
    model = TinyRoMa(xfeat=xfeat, freeze_xfeat=freeze_xfeat, exact_softmax=exact_softmax)
    if weights is not None:
        model.load_state_dict(weights)
    return model


def tiny_roma_v1_outdoor(device, weights=None, xfeat=None):
    """model_zoo/__init__.py:16-28.  The XFeat backbone is an external hub repository: pass it in when offline."""
    if weights is None:
        weights = torch.hub.load_state_dict_from_url(weight_urls["tiny_roma_v1"]["outdoor"], map_location=device)
    if xfeat is None:
        xfeat = torch.hub.load("verlab/accelerated_features", "XFeat", pretrained=True, top_k=4096).net
    return tiny_roma_v1_model(weights=weights, xfeat=xfeat).to(device)
