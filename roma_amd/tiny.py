"""TinyRoMa (XFeat backbone) on MI355X — reference: romatch/models/tiny.py:93-367.

The backbone and the small matcher conv stacks are library convolutions; the full correlation volume + soft-argmax
(`corr_volume` + `pos_embed`, tiny.py:241-254,178-203) is ONE fused MFMA kernel that never writes the volume
(ops.tiny_corr_posembed), and the warps are the same bilinear-gather kernel the big model uses."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .matcher import pixel_grid


class BasicLayer(nn.Module):
    """Conv2d -> BatchNorm(affine=False) -> ReLU, tiny.py:76-91."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, bias=False, relu=True):
        super().__init__()
        self.layer = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, stride=stride, dilation=dilation, bias=bias),
                                   nn.BatchNorm2d(out_channels, affine=False), nn.ReLU(inplace=True) if relu else nn.Identity())

    def forward(self, x):
        return self.layer(x)


class TinyRoMa(nn.Module):
    def __init__(self, xfeat=None, freeze_xfeat=True, sample_mode="threshold_balanced", symmetric=False, exact_softmax=False):
        super().__init__()
        for name in ("heatmap_head", "keypoint_head", "fine_matcher"):
            if hasattr(xfeat, name):
                delattr(xfeat, name)
        if freeze_xfeat:
            xfeat.train(False)
            self.xfeat = [xfeat]
        else:
            self.xfeat = nn.ModuleList([xfeat])
        self.freeze_xfeat = freeze_xfeat
        md, fd = 256, 64
        self.coarse_matcher = nn.Sequential(BasicLayer(64 + 64 + 2, md), BasicLayer(md, md), BasicLayer(md, md), BasicLayer(md, md),
                                            nn.Conv2d(md, 3, kernel_size=1, bias=True, padding=0))
        self.fine_matcher = nn.Sequential(BasicLayer(24 + 24 + 2, fd), BasicLayer(fd, fd), BasicLayer(fd, fd), BasicLayer(fd, fd),
                                          nn.Conv2d(fd, 3, kernel_size=1, bias=True, padding=0))
        self.sample_mode = sample_mode
        self.sample_thresh = 0.05
        self.symmetric = symmetric
        self.exact_softmax = exact_softmax

    @property
    def device(self):
        return self.fine_matcher[-1].weight.device

    def preprocess_tensor(self, x):
        H, W = x.shape[-2:]
        _H, _W = (H // 32) * 32, (W // 32) * 32
        return F.interpolate(x, (_H, _W), mode="bilinear", align_corners=False), H / _H, W / _W

    def forward_single(self, x):                                                               # tiny.py:144-162
        xf = self.xfeat[0]
        if next(xf.parameters()).device != x.device:
            xf.to(x.device)
        x = xf.norm(x.mean(dim=1, keepdim=True))
        x1 = xf.block1(x)
        x2 = xf.block2(x1 + xf.skip1(x))
        x3 = xf.block3(x2)
        x4 = xf.block4(x3)
        x5 = xf.block5(x4)
        x4 = F.interpolate(x4, x3.shape[-2:], mode="bilinear")
        x5 = F.interpolate(x5, x3.shape[-2:], mode="bilinear")
        return x2, xf.block_fusion(x3 + x4 + x5)

    def corr_volume(self, feat0, feat1):
        """Materialised volume (tiny.py:241-254) — kept for API parity; forward() never calls it."""
        B, C, H0, W0 = feat0.shape
        _, _, H1, W1 = feat1.shape
        cv = torch.einsum("bci,bcj->bji", feat0.reshape(B, C, -1), feat1.reshape(B, C, -1))
        return cv.reshape(B, H1, W1, H0, W0) / (C ** 0.5)

    def pos_embed_from_features(self, f0, f1):
        """corr_volume + pos_embed fused.  The reference's fast path is only shape-correct for B=1; the
        kernel applies the per-pair B=1 semantics to every batch item."""
        exact = self.training or self.exact_softmax
        return ops.tiny_corr_posembed(f0, f1, exact=exact)

    def to_pixel_coordinates(self, coords, H_A, W_A, H_B=None, W_B=None):
        if coords.shape[-1] == 2:
            return self._to_pixel_coordinates(coords, H_A, W_A)
        if isinstance(coords, (list, tuple)):
            kA, kB = coords[0], coords[1]
        else:
            kA, kB = coords[..., :2], coords[..., 2:]
        return self._to_pixel_coordinates(kA, H_A, W_A), self._to_pixel_coordinates(kB, H_B, W_B)

    def _to_pixel_coordinates(self, coords, H, W):
        return torch.stack((W / 2 * (coords[..., 0] + 1), H / 2 * (coords[..., 1] + 1)), axis=-1)

    @torch.no_grad()
    def forward(self, batch):                                                                  # tiny.py:330-367
        im0, _, _ = self.preprocess_tensor(batch["im_A"])
        im1, _, _ = self.preprocess_tensor(batch["im_B"])
        H1, W1 = im1.shape[-2:]
        to_norm = torch.tensor((2 / W1, 2 / H1, 1.0), device=im0.device)[None, :, None, None]
        if im0.shape[-2:] == im1.shape[-2:]:
            xf, xc = self.forward_single(torch.cat([im0, im1], dim=0))
            f0c, f1c = xc.chunk(2)
            f0f, f1f = xf.chunk(2)
        else:
            f0f, f0c = self.forward_single(im0)
            f1f, f1c = self.forward_single(im1)
        cw = self.pos_embed_from_features(f0c, f1c)
        cm = torch.cat((cw, torch.zeros_like(cw[:, -1:])), dim=1)
        w1 = ops.warp_bilinear(f1c.contiguous(), cm[:, :2].contiguous())
        cm = cm + self.coarse_matcher(torch.cat((f0c, w1, cw), dim=1)) * to_norm
        out = {8: {"flow": cm[:, :2], "certainty": cm[:, 2:]}}
        up = ops.interp_bilinear(cm, f0f.shape[-2:])
        w1f = ops.warp_bilinear(f1f.contiguous(), up[:, :2].contiguous())
        fm = up + self.fine_matcher(torch.cat((f0f, w1f, up[:, :2]), dim=1)) * to_norm
        out[4] = {"flow": fm[:, :2], "certainty": fm[:, 2:]}
        return out

    @torch.inference_mode()
    def match(self, im0, im1, *args, batched=True):                                            # tiny.py:263-295
        from PIL import Image
        import numpy as np
        if isinstance(im0, (str, os.PathLike)):
            im0, im1 = Image.open(im0), Image.open(im1)
        if isinstance(im0, Image.Image):
            batched = False
            conv = lambda im: torch.from_numpy(np.array(im, dtype=np.float32).transpose(2, 0, 1) / 255.0)[None].to(self.device)  # noqa: E731
            im0, im1 = conv(im0), conv(im1)
        B, C, H0, W0 = im0.shape
        self.train(False)
        c = self.forward({"im_A": im0, "im_B": im1})
        flow = ops.interp_bilinear(c[8]["flow"], (H0, W0)).permute(0, 2, 3, 1)
        cert = ops.interp_bilinear(c[8]["certainty"], (H0, W0))
        grid = pixel_grid(B, H0, W0, flow.device).permute(0, 2, 3, 1)
        warp, cert = torch.cat((grid, flow), dim=-1), cert[:, 0].sigmoid()
        return (warp, cert) if batched else (warp[0], cert[0])

    def sample(self, matches, certainty, num=5_000, seed=None):                               # tiny.py:297-327
        """Same exponential-race formulation as RegressionMatcher.sample (roma_amd/matcher.py)."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        thresh = self.sample_thresh if "threshold" in self.sample_mode else -1.0
        matches, certainty = matches.reshape(-1, 4), certainty.reshape(-1)
        expansion = 4 if "balanced" in self.sample_mode else 1
        good = torch.topk(ops.race_keys(certainty, thresh, seed), min(expansion * num, len(certainty))).indices
        gm, gc = matches[good], certainty[good]
        if thresh >= 0:
            gc = torch.where(gc > thresh, torch.ones_like(gc), gc)
        if "balanced" not in self.sample_mode:
            return gm, gc
        density = ops.kde(gm, std=0.1, half=True, down=1)
        p = 1 / (density + 1)
        p[density < 10] = 1e-7
        bal = torch.topk(ops.race_keys(p, -1.0, seed, counter=good, stage=1), min(num, len(gc))).indices
        return gm[bal], gc[bal]


class XFeatBackbone(nn.Module):
    """The XFeat backbone TOPOLOGY ("XFeat: Accelerated Features for Lightweight Image Matching", CVPR 2024) with the
    attributes TinyRoMa.forward_single touches (tiny.py:146-159): 24-channel stride-4 and 64-channel stride-8 outputs.  The
    reference fetches the network from a hub repository (model_zoo/__init__.py:22-26) that is absent offline; its layer
    plan is restated, widened to 32 channels, at experiments/model_tiny1.py:34-117 — this is that plan at the published
    widths.  Used with synthetic weights by bench.py's `tiny` workload; parity of the backbone itself is unpinned."""

    def __init__(self):
        super().__init__()
        L = BasicLayer
        self.norm = nn.InstanceNorm2d(1)
        self.skip1 = nn.Sequential(nn.AvgPool2d(4, stride=4), nn.Conv2d(1, 24, 1, stride=1, padding=0))
        self.block1 = nn.Sequential(L(1, 4, stride=1), L(4, 8, stride=2), L(8, 8, stride=1), L(8, 24, stride=2))
        self.block2 = nn.Sequential(L(24, 24, stride=1), L(24, 24, stride=1))
        self.block3 = nn.Sequential(L(24, 64, stride=2), L(64, 64, stride=1), L(64, 64, 1, padding=0))
        self.block4 = nn.Sequential(L(64, 64, stride=2), L(64, 64, stride=1), L(64, 64, stride=1))
        self.block5 = nn.Sequential(L(64, 128, stride=2), L(128, 128, stride=1), L(128, 128, stride=1), L(128, 64, 1, padding=0))
        self.block_fusion = nn.Sequential(L(64, 64, stride=1), L(64, 64, stride=1), nn.Conv2d(64, 64, 1, padding=0))
