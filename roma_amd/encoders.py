"""Feature pyramids: VGG19-BN (scales 1,2,4,8) + frozen DINOv2 ViT-L/14 ("scale 16").
Reference: romatch/models/encoders.py:61-122.  Library convolutions/GEMMs on PyTorch-ROCm (MIOpen / hipBLASLt),
channels-last, BatchNorm folded into the convolutions at prepare time, in the model's amp dtype."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

from .transformer import DinoViT

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def _ver(t):
    """In-place-update counter used as a cache key (inference tensors do not have one)."""
    return 0 if t.is_inference() else t._version


class VGG19(nn.Module):
    """Parameters live in `layers.{0..39}` exactly like torchvision's vgg19_bn().features[:40] (encoders.py:64)."""

    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in VGG19_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        self.layers = nn.ModuleList(layers[:40])
        self._folded = None

    def fold(self, dtype, nhwc=False):
        """conv+BN(eval) -> one conv: w' = w*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta.  Cached."""
        key = (dtype, nhwc, self.layers[0].weight.device, tuple(_ver(p) for p in self.parameters()))
        if self._folded is not None and self._folded[0] == key:
            return self._folded[1]
        plan = []
        i = 0
        while i < len(self.layers):
            m = self.layers[i]
            if isinstance(m, nn.MaxPool2d):
                plan.append(None)
                i += 1
                continue
            bn = self.layers[i + 1]
            s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w = (m.weight.float() * s[:, None, None, None]).to(dtype).contiguous(memory_format=torch.channels_last if nhwc else torch.contiguous_format)
            b = ((m.bias.float() - bn.running_mean.float()) * s + bn.bias.float()).to(dtype)
            plan.append((w, b))
            i += 3
        self._folded = (key, plan)
        return plan

    @torch.no_grad()
    def forward(self, x, dtype=torch.float16):
        feats, scale = {}, 1
        # channels-last activations (round 3): with MIOpen's solver search on (enable_miopen_find) the fastest 3x3 solvers on gfx950 are
        # CK's NHWC implicit-GEMM kernels, and on planar maps MIOpen wraps them in `batched_transpose` passes (1.3 ms per match);
        # conv + pool chain of the 560 + 864 passes: planar 4.65 ms, channels-last 3.44 ms (tools/scratch/vgg_layout.py).  With the
        # heuristic solver pick (ROMA_MIOPEN_FIND=0) planar is the faster layout (round 2: 5.7 vs 7.0 ms) and is kept.  The decoder's
        # projection GEMM reads either layout in place.
        nhwc = torch.backends.cudnn.benchmark and x.is_cuda
        x = x.to(dtype).contiguous(memory_format=torch.channels_last if nhwc else torch.contiguous_format)
        plan = self.fold(dtype, nhwc)
        for i, step in enumerate(plan):
            if step is None:
                continue                                          # the pool was fused into the layer before it (or is the unused last one)
            pooled_next = i + 1 < len(plan) and plan[i + 1] is None
            x = F.conv2d(x, step[0], None, padding=1)
            if not pooled_next:
                x = ops.bias_relu_(x, step[1])                    # one epilogue pass instead of add_ + relu_
                continue
            feats[scale] = x                                      # captured BEFORE the pool (encoders.py:73-76)
            last = i + 2 >= len(plan)                             # layers[:40] ends on a pool whose output nothing reads
            if last or x.shape[-1] % 2 or x.shape[-2] % 2:
                ops.bias_relu_(x, step[1])
                x = None if last else F.max_pool2d(x, 2, 2)
            else:
                x = ops.bias_relu_pool2_(x, step[1])              # bias + ReLU in place and the 2x2 max in the same pass
            scale *= 2
        return feats


def enable_miopen_find():
    """MIOpen picks its convolution solver per shape either from a heuristic (torch's default: for most of the VGG19 3x3 layers at
    560 / 864 that is im2col + GEMM: 22 `Im2d2Col_v2` launches per match, 0.9 ms) or by timing every applicable solver once per
    (process, shape) — torch.backends.cudnn.benchmark.  Measured on MI355X: VGG19 560 + 864 passes 4.82 -> 4.18 ms, the whole match
    20.18 -> 19.77 ms, for ~30 s of one-time solver search in a fresh process (gpurun_out: tools/scratch/find_test.py).  On unless
    ROMA_MIOPEN_FIND=0."""
    import os
    if os.environ.get("ROMA_MIOPEN_FIND", "1") != "0":
        torch.backends.cudnn.benchmark = True


class CNNandDinov2(nn.Module):
    def __init__(self, dinov2: Optional[DinoViT] = None, amp_dtype=torch.float16):
        super().__init__()
        enable_miopen_find()
        self.cnn = VGG19()
        self.amp_dtype = amp_dtype
        self.dinov2_vitl14 = [dinov2 if dinov2 is not None else DinoViT()]   # outside the module tree (encoders.py:104)

    def train(self, mode: bool = True):
        return self.cnn.train(mode)

    def _vit(self, device):
        """The ViT in the amp dtype on `device`.  The module in `dinov2_vitl14[0]` stays the fp32 master (so switching
        amp_dtype never re-rounds already rounded weights); casts are cached per (device, dtype)."""
        master = self.dinov2_vitl14[0]
        key = (str(device), self.amp_dtype, tuple(_ver(p) for p in master.parameters()))
        cache = self.__dict__.setdefault("_vit_cache", {})
        hit = cache.get(key)
        if hit is None:
            import copy
            cache.clear()
            if self.amp_dtype == torch.float32 and next(master.parameters()).device == device:
                hit = master.eval()
            else:
                with torch.inference_mode(False), torch.no_grad():      # ordinary tensors even when called under inference_mode
                    hit = copy.deepcopy(master).to(device=device, dtype=self.amp_dtype).eval()
            cache[key] = hit
        return hit

    @torch.no_grad()
    def vit_features(self, x):
        """DINOv2 patch tokens as a (B,1024,H/14,W/14) channels-last view (encoders.py:114-121)."""
        B, C, H, W = x.shape
        t = self._vit(x.device).patch_tokens(x.to(self.amp_dtype))
        return t.reshape(B, H // 14, W // 14, -1).permute(0, 3, 1, 2)

    @torch.no_grad()
    def forward(self, x, upsample=False):
        pyr = self.cnn(x, self.amp_dtype)
        if not upsample:                                                      # encoders.py:114
            pyr[16] = self.vit_features(x)
        return pyr
