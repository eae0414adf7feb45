"""Transformer pieces of the hot path: hipBLASLt GEMMs, torch SDPA (or, opt-in, ops.attention: the hand-written
flash-style kernel), ops.add_layernorm at the seams.

Module/parameter names follow the reference so its checkpoints load unchanged:
`Block` = romatch/models/transformer/layers/block.py:36-107 (pre-LN, optional LayerScale),
`TransformerDecoder` = romatch/models/transformer/__init__.py:10-46, `DinoViT` = transformer/dinov2.py:43-237 (vit_large).
These are dense GEMM/attention work that the north star leaves to the vendor libraries; the hand-written HIP kernels
start where the decoder's logits are consumed (ops.cls_rows_to_flow).
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# Attention of the 64-wide DINOv2 heads in the 16-bit modes: torch's scaled_dot_product_attention (aotriton; the default) or
# ops.attention (the hand-written flash-style kernel, roma_attention_fwd; ROMA_ATTENTION=hip).  Both are equally accurate against an
# fp64 attention (identical max / rms error on every shape tried), and stand-alone the hand-written kernel is faster at one
# pair per step (53.2 vs 59.1 us per layer at B = 2, slower at eight pairs: 355 vs 319 us at B = 16).  Round 3 tried it as the default
# up to B = 4 and measured NO step-time gain (18.38 vs 18.28 ms: the 24 layers sit beside the VGG stream, which fills what they free),
# while the fp16 statistics moved inside their pinned bounds but visibly (32 instead of 17 of 3 200 coarse arg-max flips against the
# fp32 mode at 560 -> 864, warp p99 against the reference 0.12 instead of 0.04: the GP amplifies any change of the 16-bit noise into
# different near-tie decisions; torch's `math` SDPA backend gives 34 flips) — so it stays opt-in.
ATTENTION_KERNEL = os.environ.get("ROMA_ATTENTION", "sdpa")       # "sdpa" | "hip" (tests may also set True / False)


def _use_hip_attention(batch):
    return ATTENTION_KERNEL in (True, "hip")


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, n_valid=None):
        """n_valid: the first n_valid tokens are real, the rest is row padding (see DinoViT.patch_tokens): every token
        queries, only real tokens are keys / values, so padding never reaches a real token."""
        B, N, C = x.shape
        hd = C // self.num_heads
        qkv = self.qkv(x).view(B, N, 3, self.num_heads, hd)
        if hd == 64 and qkv.dtype in (torch.float16, torch.bfloat16) and _use_hip_attention(B):
            # the hand-written flash-style kernel reads q / k / v in place and writes the projection's input layout
            return self.proj(ops.attention(qkv, n_valid))
        qkv = qkv.permute(2, 0, 3, 1, 4)
        k, v = (qkv[1], qkv[2]) if n_valid is None else (qkv[1][:, :, :n_valid], qkv[2][:, :, :n_valid])
        o = F.scaled_dot_product_attention(qkv[0], k, v)                      # scale = head_dim^-0.5, as attention.py:53-56
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class LayerScale(nn.Module):
    def __init__(self, dim, init=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=False, init_values=None, ln_eps=1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=ln_eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.ls1 = LayerScale(dim, init_values) if init_values else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=ln_eps)
        self.mlp = Mlp(dim, int(dim * 4.0))
        self.ls2 = LayerScale(dim, init_values) if init_values else nn.Identity()

    @staticmethod
    def _residual(x, y, ls):
        # x + gamma * y in one elementwise pass (the reference's LayerScale + residual are two, layers/block.py:87-107)
        return torch.addcmul(x, y, ls.gamma.to(y.dtype)) if isinstance(ls, LayerScale) else x + y

    def forward(self, x, n_valid=None):
        x = self._residual(x, self.attn(self.norm1(x), n_valid), self.ls1)
        return self._residual(x, self.mlp(self.norm2(x)), self.ls2)


def _seam_params(blocks, final_norm=None):
    """fp32 contiguous copies of what the fused add + LayerNorm seams read (LayerNorm affine, LayerScale gammas), cached on
    the module list owner; the modules passed here are frozen inference copies."""
    out = []
    for blk in blocks:
        f = lambda t: t.detach().float().contiguous()
        out.append(dict(n1=(f(blk.norm1.weight), f(blk.norm1.bias), blk.norm1.eps), n2=(f(blk.norm2.weight), f(blk.norm2.bias), blk.norm2.eps),
                        ls1=f(blk.ls1.gamma) if isinstance(blk.ls1, LayerScale) else None,
                        ls2=f(blk.ls2.gamma) if isinstance(blk.ls2, LayerScale) else None))
    fin = None
    if final_norm is not None:
        fin = (final_norm.weight.detach().float().contiguous(), final_norm.bias.detach().float().contiguous(), final_norm.eps)
    return out, fin


def run_blocks_fused(blocks, params, x, n_valid, compute_dtype, final=None):
    """A stack of pre-LN blocks (layers/block.py:87-107) with every `residual add (+ LayerScale)` fused with the NEXT
    LayerNorm into one kernel (ops.add_layernorm): x is the residual stream — fp32 for the decoder transformer (what autocast
    gives the reference, transformer/__init__.py:30-46), the amp dtype for the DINOv2 trunk (the reference casts that module
    itself, encoders.py:102-103) — and is updated in place; attention and MLP run on `compute_dtype` tensors.  Returns
    final LayerNorm(x) when `final` = (weight, bias, eps), else x cast to compute_dtype (the cast in front of `to_out`)."""
    from . import ops
    h = ops.add_layernorm(x, None, *params[0]["n1"], compute_dtype)
    last = len(blocks) - 1
    for i, blk in enumerate(blocks):
        p = params[i]
        a = blk.attn(h, n_valid)
        h = ops.add_layernorm(x, a, *p["n2"], compute_dtype, ls=p["ls1"])
        m = blk.mlp(h)
        if i < last:
            h = ops.add_layernorm(x, m, *params[i + 1]["n1"], compute_dtype, ls=p["ls2"])
        elif final is not None:
            h = ops.add_layernorm(x, m, *final, compute_dtype, ls=p["ls2"])
        else:
            h = ops.add_layernorm(x, m, None, None, 0.0, compute_dtype, ls=p["ls2"])
    return h


class TransformerDecoder(nn.Module):
    """Returns the token-major logit rows (B, H*W, out_dim) — the (B,out_dim,H,W) permute of the reference
    (transformer/__init__.py:44) is folded into the consumer kernel's strides."""

    def __init__(self, blocks, hidden_dim, out_dim, is_classifier=True):
        super().__init__()
        self.blocks = blocks
        self.to_out = nn.Linear(hidden_dim, out_dim)
        self.hidden_dim, self.out_dim, self.is_classifier = hidden_dim, out_dim, is_classifier
        self._scales = [16]

    def scales(self):
        return self._scales.copy()

    def forward_rows(self, tokens, n_valid=None, owned=False):
        """tokens (B, N, hidden) -> logit rows (B, N, out_dim).  When the module's GEMM weights are 16-bit (the cached
        amp-dtype copy the Decoder makes) the arithmetic is the reference's autocast (transformer/__init__.py:30-46): fp32
        token / residual stream and LayerNorm, 16-bit GEMMs and attention, 16-bit logits.
        n_valid: the caller already row-padded the sequence (rows >= n_valid are padding); owned=True: `tokens` is a fresh
        fp32 contiguous buffer this call may update in place."""
        n = tokens.shape[1] if n_valid is None else n_valid
        wdt = self.to_out.weight.dtype
        if wdt == torch.float32:
            return self.to_out(self.blocks(tokens[:, :n].float()))
        # row-pad the sequence to the attention / GEMM tile (see DinoViT.patch_tokens); padding rows are queries only and are
        # dropped from the returned view
        if owned and tokens.dtype == torch.float32 and tokens.is_contiguous() and tokens.shape[1] % 128 == 0:
            t = tokens
        else:
            t = F.pad(tokens[:, :n].float(), (0, 0, 0, -n % 128)) if n % 128 else tokens.float().clone()
        cache = self.__dict__.get("_seams")
        if cache is None:
            cache = self.__dict__["_seams"] = _seam_params(self.blocks)
        z = run_blocks_fused(list(self.blocks), cache[0], t, n if t.shape[1] != n else None, wdt)
        return self.to_out(z)[:, :n]


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoViT(nn.Module):
    """DINOv2 ViT-L/14 trunk: LN eps 1e-6, qkv bias, LayerScale; returns the normalised patch tokens."""

    def __init__(self, img_size=518, patch_size=14, dim=1024, depth=24, heads=16, init_values=1.0):
        super().__init__()
        self.patch_size = patch_size
        n = (img_size // patch_size) ** 2
        self.patch_embed = PatchEmbed(patch_size, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.blocks = nn.ModuleList([Block(dim, heads, qkv_bias=True, init_values=init_values, ln_eps=1e-6) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self._pos_cache = {}

    def pos_encoding(self, w, h, dtype):
        """dinov2.py:166-190: bicubic resize of the 37x37 table with scale_factor (w0+0.1)/37; cached per input size."""
        key = (w, h, dtype, self.pos_embed.device, 0 if self.pos_embed.is_inference() else self.pos_embed._version)
        hit = self._pos_cache.get(key)
        if hit is not None:
            return hit
        N = self.pos_embed.shape[1] - 1
        w0, h0 = w // self.patch_size, h // self.patch_size
        if w0 * h0 == N and w == h:
            pe = self.pos_embed
        else:
            p = self.pos_embed.float()
            dim = p.shape[-1]
            side = int(math.sqrt(N))
            grid = F.interpolate(p[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2),
                                 scale_factor=((w0 + 0.1) / math.sqrt(N), (h0 + 0.1) / math.sqrt(N)), mode="bicubic")
            assert grid.shape[-2] == w0 and grid.shape[-1] == h0
            pe = torch.cat((p[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)
        pe = pe.to(dtype)
        self._pos_cache = {key: pe}
        return pe

    def patch_tokens(self, x):
        B, _, w, h = x.shape
        t = self.patch_embed(x)
        t = torch.cat((self.cls_token.expand(B, -1, -1), t), dim=1) + self.pos_encoding(w, h, t.dtype)
        n = t.shape[1]
        n_valid = None
        if t.dtype != torch.float32 and n % 128:
            # The flash-attention kernel runs 1.5x faster when the QUERY length is a multiple of its 128-row tile
            # (1601 tokens: 90 us, 1664: 61 us per layer on MI355X).  The sequence is padded once, here; padded rows are
            # queries only (keys / values are sliced to the real tokens in every block), so they cannot influence a real
            # token, and the GEMMs' tile count does not change (3202 -> 3328 rows = 13 tiles of 256 either way).
            n_valid = n
            t = F.pad(t, (0, 0, 0, 128 - n % 128))
        if t.dtype == torch.float32 or not t.is_cuda:
            for blk in self.blocks:
                t = blk(t, n_valid)
            return self.norm(t[:, 1:n])
        # 16-bit modes: residual add + LayerScale + next LayerNorm in one kernel per seam (49 launches instead of 97)
        cache = self.__dict__.get("_seams")
        if cache is None:
            cache = self.__dict__["_seams"] = _seam_params(self.blocks, self.norm)
        t = t.contiguous()
        return run_blocks_fused(list(self.blocks), cache[0], t, n_valid, t.dtype, final=cache[1])[:, 1:n]
