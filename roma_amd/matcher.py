"""RegressionMatcher / Decoder / GP / ConvRefiner of the RoMa hot path, MI355X-native.

Same public surface as the reference's romatch/models/matcher.py (class names, constructor-visible attributes,
`match / sample / to_pixel_coordinates / to_normalized_coordinates / match_keypoints / get_output_resolution /
visualize_warp / forward / forward_symmetric`), same state-dict key layout, so `load_state_dict` of a reference
checkpoint works unchanged.  What is different is underneath: the per-level arithmetic runs in the hand-written
gfx950 kernels of libroma_hip.so (roma_amd.ops), activations are channels-last in the amp dtype, BatchNorm is folded,
and the dense GEMM/conv/attention work goes to the ROCm libraries through PyTorch.  There is no CPU path: every
kernel wrapper raises on a non-GPU tensor.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional
from warnings import warn

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .transformer import Block, TransformerDecoder  # noqa: F401  (re-exported like the reference's module)

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _round_up(n, m):
    return (n + m - 1) // m * m


def pixel_grid(b, h, w, device):
    """(b,2,h,w) fp32 pixel centres (x,y) — Decoder.get_placeholder_flow, matcher.py:303-315."""
    ys = torch.linspace(-1 + 1 / h, 1 - 1 / h, h, device=device)
    xs = torch.linspace(-1 + 1 / w, 1 - 1 / w, w, device=device)
    return torch.stack((xs[None, :].expand(h, w), ys[:, None].expand(h, w)), dim=0)[None].expand(b, 2, h, w).contiguous()


def _params_version(module):
    return tuple((p.data_ptr(), 0 if p.is_inference() else p._version) for p in list(module.parameters()) + list(module.buffers()))


# ------------------------------------------------------------------------------------------------
# ConvRefiner — matcher.py:17-143
# ------------------------------------------------------------------------------------------------
def _zero_pad(t, *shape):
    out = torch.zeros(shape, dtype=t.dtype, device=t.device)
    out[tuple(slice(0, s) for s in t.shape)] = t
    return out


class ConvRefiner(nn.Module):
    """x, warped y, displacement embedding and local correlation are assembled in ONE channels-last buffer (pitch padded to
    8 channels): the decoder's projection GEMM writes x there, three kernels write the other channel slices.  A block
    (depthwise 5x5 + BN + ReLU + 1x1) is, by width: D <= 32 one fused kernel (ops.refiner_block); 32 < D <= 160 the
    depthwise kernel + the streaming MFMA 1x1 (ops.pointwise_mfma); wider, the depthwise kernel + a hipBLASLt GEMM.  fp32
    (parity mode) always takes the separate kernels."""

    def __init__(self, in_dim, hidden_dim, out_dim, hidden_blocks, displacement_emb_dim, local_corr_radius=None,
                 amp_dtype=torch.float16):
        super().__init__()
        assert in_dim == hidden_dim, "depthwise refiner blocks keep the channel count (all shipped models)"

        def block(i, o):
            return nn.Sequential(nn.Conv2d(i, o, 5, 1, 2, groups=i), nn.BatchNorm2d(o), nn.ReLU(inplace=True), nn.Conv2d(o, o, 1, 1, 0))

        self.block1 = block(in_dim, hidden_dim)
        self.hidden_blocks = nn.Sequential(*[block(hidden_dim, hidden_dim) for _ in range(hidden_blocks)])
        self.out_conv = nn.Conv2d(hidden_dim, out_dim, 1, 1, 0)
        self.disp_emb = nn.Conv2d(2, displacement_emb_dim, 1, 1, 0)
        self.has_displacement_emb = True
        self.local_corr_radius = local_corr_radius
        self.in_dim, self.out_dim = in_dim, out_dim
        self.amp_dtype = amp_dtype
        self._prep = None

    def prepare(self, dtype):
        key = (dtype, _params_version(self))
        if self._prep is not None and self._prep[0] == key:
            return self._prep[1]
        D = self.in_dim
        # channel pitch: 16-byte packets; the wide refiners (D = 1144, 1384) are padded to a multiple of 64 because hipBLASLt runs
        # their square 1x1-conv GEMMs 23-34 % faster at K = N = 1152 / 1408 (576 already is a multiple of 64)
        Dp = _round_up(D, 64) if D > 160 else _round_up(D, 8)
        dev = self.out_conv.weight.device
        # whole-block fusion (ops.refiner_block): measured 1.8x faster than dwconv + pointwise at D = 24, on par at D = 144
        # (LDS-read and VALU bound there) -> narrow refiners only unless ROMA_FUSED_BLOCK=160
        fuse_max = int(os.environ.get("ROMA_FUSED_BLOCK", "32"))
        fused = Dp <= min(fuse_max, 160) and dtype in (torch.float16, torch.bfloat16)
        mid = (not fused) and 32 < Dp <= 160 and dtype in (torch.float16, torch.bfloat16) and os.environ.get("ROMA_PW_MFMA", "1") != "0"
        # D = 576 (the scale-4 refiner), fp16: the whole block as one kernel (ops.refiner_block_wide, csrc/refiner_wide.hip) — prepared
        # here, used per call when the map fills the chip (see _body): 160 vs 177 us per block at 216^2 x 2, but 97 vs 80 us at 140^2 x 2
        # (324 tiles of 128 pixels on 256 CUs: one and a quarter rounds)
        wide = Dp == 576 and dtype == torch.float16 and os.environ.get("ROMA_WIDE_BLOCK", "1") != "0"
        blocks = []
        for blk in [self.block1] + list(self.hidden_blocks):
            dw, bn, _, pw = blk
            s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w25 = torch.zeros(25, Dp, device=dev)
            w25[:, :D] = dw.weight.float().reshape(D, 25).t()
            scale = torch.zeros(Dp, device=dev)
            shift = torch.zeros(Dp, device=dev)
            scale[:D] = s
            b_dw = dw.bias.float() if dw.bias is not None else torch.zeros(D, device=dev)
            shift[:D] = (b_dw - bn.running_mean.float()) * s + bn.bias.float()
            wt = torch.zeros(Dp, Dp, device=dev)
            wt[:D, :D] = pw.weight.float().reshape(D, D).t()            # (in, out): X @ wt
            b = torch.zeros(Dp, device=dev)
            b[:D] = pw.bias.float()
            if wide:
                blocks.append((w25.contiguous(), scale, shift, wt.to(dtype).contiguous(), b.to(dtype),
                               ops.refiner_wide_taps(w25, dtype), ops.refiner_wide_pack(wt.t().contiguous().to(dtype)), b.contiguous()))
            elif fused:
                # one kernel per block (ops.refiner_block): weights zero-padded to kpad channels, 1x1 weight as [out][in]
                kp = 32 if Dp <= 32 else 160
                blocks.append((_zero_pad(w25, 25, kp).to(dtype), _zero_pad(scale, kp), _zero_pad(shift, kp),
                               _zero_pad(wt.t(), kp, kp).to(dtype), _zero_pad(b, kp)))
            elif mid:
                # mid widths (D = 144): depthwise kernel + streaming MFMA 1x1 (ops.pointwise_mfma), weight [out][in] padded to 160
                blocks.append((w25.contiguous(), scale, shift, _zero_pad(wt.t(), 160, 160).to(dtype), _zero_pad(b, 160)))
            # narrow refiners (Dp <= 32): the 1x1 conv is a streaming op -> own kernel with fp32 weights; else a library GEMM
            elif Dp <= 32:
                blocks.append((w25.contiguous(), scale, shift, wt.contiguous(), b.contiguous()))
            else:
                blocks.append((w25.contiguous(), scale, shift, wt.to(dtype).contiguous(), b.to(dtype)))
        wo = torch.zeros(Dp, self.out_dim, device=dev)
        wo[:D] = self.out_conv.weight.float().reshape(self.out_dim, D).t()
        prep = dict(D=D, Dp=Dp, fused=fused, mid=mid, wide=wide, blocks=blocks, wo=wo.contiguous(), bo=self.out_conv.bias.float(),
                    we=self.disp_emb.weight.float().reshape(-1, 2).contiguous(), be=self.disp_emb.bias.float())
        self._prep = (key, prep)
        return prep

    def new_buffer(self, B, h, w, dtype, device):
        """The channels-last (B,h,w,Dp) concat buffer of one forward; the caller may fill channels [0, C) with x itself."""
        return torch.empty((B, h, w, self.prepare(dtype)["Dp"]), dtype=dtype, device=device)

    @torch.no_grad()
    def _body(self, x, y, flow, scale_factor, dtype, buf=None, batch_shift=0):
        """Everything up to (not including) out_conv: returns the last block's activation (B,h,w,Dp) and the prepared weights.
        batch_shift: x[b] is matched against y[(b + batch_shift) % B] (forward_symmetric passes y = x, shift = B/2)."""
        P = self.prepare(dtype)
        B, C, h, w = x.shape
        D, Dp = P["D"], P["Dp"]
        E = P["we"].shape[0]
        r = self.local_corr_radius
        K = (2 * r + 1) ** 2 if r else 0
        assert 2 * C + E + K == D, "feature width does not match this refiner"
        x_in_place = buf is not None
        if buf is None:
            buf = torch.empty((B, h, w, Dp), dtype=dtype, device=x.device)
        assert buf.shape == (B, h, w, Dp) and buf.dtype == dtype and buf.is_contiguous()
        d = buf.permute(0, 3, 1, 2)
        if x_in_place:
            assert x.data_ptr() == buf.data_ptr() and x.stride(1) == 1, "x must already be channels [0, C) of buf"
        else:
            d[:, :C].copy_(x)
        if Dp > D:
            buf[..., D:].zero_()
        yy = y.to(dtype)
        ops.warp_bilinear(yy, flow, out=d[:, C:2 * C], batch_shift=batch_shift)               # matcher.py:109
        ops.disp_emb(flow, P["we"], P["be"], 40 / 32 * scale_factor, out=d[:, 2 * C:2 * C + E])  # :111-120
        if r:
            ops.local_correlation(d[:, :C], yy, r, flow=flow, out=d[:, 2 * C + E:D], batch_shift=batch_shift)   # :121-125
        M = B * h * w
        cur = buf
        if P["wide"] and B * ((h + 7) // 8) * ((w + 15) // 16) >= 2 * 256:
            nxt = torch.empty_like(buf)
            for blk in P["blocks"]:                                                            # :139-140
                ops.refiner_block_wide(cur, blk[5], blk[1], blk[2], blk[6], blk[7], out=nxt)
                cur, nxt = nxt, cur
            return cur, P
        if P["fused"]:
            nxt = torch.empty_like(buf)
            for (w25, scale, shift, wt, b) in P["blocks"]:                                     # :139-140
                ops.refiner_block(cur, w25, scale, shift, wt, b, Dp, out=nxt)
                cur, nxt = nxt, cur
            return cur, P
        for blk in P["blocks"]:                                                                # :139-140
            w25, scale, shift, wt, b = blk[:5]
            t = ops.dwconv5x5_bn_relu(cur.permute(0, 3, 1, 2), w25, scale, shift)
            rows = t.permute(0, 2, 3, 1).reshape(M, Dp)
            if P["mid"]:
                cur = ops.pointwise_mfma(rows, wt, b, Dp)
            else:
                cur = ops.pointwise_small(rows, wt, b) if Dp <= 32 else torch.addmm(b, rows, wt)
            cur = cur.view(B, h, w, Dp)
        return cur, P

    @torch.no_grad()
    def forward(self, x, y, flow, scale_factor=1, logits=None, dtype=None):
        """x, y: (B,C,h,w) features; flow (B,2,h,w) fp32.  Returns (delta_flow (B,2,h,w), delta_certainty (B,1,h,w)) fp32
        — matcher.py:105-143."""
        flow = flow.float().contiguous()
        cur, P = self._body(x, y, flow, scale_factor, dtype or self.amp_dtype)
        B, h, w, Dp = cur.shape
        out = torch.addmm(P["bo"], cur.reshape(B * h * w, Dp).float(), P["wo"])                # out_conv in fp32, :141
        out = out.view(B, h, w, self.out_dim).permute(0, 3, 1, 2).contiguous()
        return out[:, :-1], out[:, -1:]

    @torch.no_grad()
    def forward_update(self, x, y, flow, certainty, scale_factor, sx, sy, dtype=None, buf=None, batch_shift=0):
        """The Decoder's use of the refiner (matcher.py:393-402) with out_conv fused into the update kernel:
        flow (B,2,h,w) fp32 contiguous is updated IN PLACE by (sx*dx, sy*dy); returns (flow, certainty + dcert)."""
        cur, P = self._body(x, y, flow, scale_factor, dtype or self.amp_dtype, buf=buf, batch_shift=batch_shift)
        return ops.refiner_head(cur, P["wo"], P["bo"], flow, certainty, sx, sy)


# ------------------------------------------------------------------------------------------------
# GP with the cosine kernel — matcher.py:145-273
# ------------------------------------------------------------------------------------------------
class CosKernel(nn.Module):
    def __init__(self, T=0.2, learn_temperature=False):
        super().__init__()
        self.T = T

    def __call__(self, x, y, eps=1e-6):
        return ops.cos_kernel(x, y, T=self.T, eps=eps)


class GP(nn.Module):
    """Posterior mean of the Fourier positional basis (no_cov=True, the shipped configuration).  The two kernel
    matrices come from the fp32-MFMA CosKernel kernel; `inv(K_yy + sigma I) @ f` (matcher.py:259-263) is a hand-blocked
    Cholesky solve (ops.spd_solve: K_yy + sigma I is SPD; the vendor batched potrf/potrs takes ~10 ms at 1600 x 1600,
    the reference's explicit inverse ~20 ms); the dead K_xx (matcher.py:255) is skipped."""

    def __init__(self, gp_dim=512, T=0.2, sigma_noise=0.1, kernel=None, **_):
        super().__init__()
        self.K = CosKernel(T=T)
        self.sigma_noise = sigma_noise
        self.pos_conv = nn.Conv2d(2, gp_dim, 1, 1)
        self.dim = gp_dim
        self._basis = {}

    def basis(self, b, h, w, device):
        key = (h, w, device, _params_version(self.pos_conv))
        f = self._basis.get(key)
        if f is None:
            g = pixel_grid(1, h, w, device)
            f = torch.cos(8 * math.pi * F.conv2d(g, self.pos_conv.weight.float(), self.pos_conv.bias.float()))   # :222-247
            f = f.flatten(2).transpose(1, 2).contiguous()
            self._basis = {key: f}
        return f.expand(b, -1, -1)

    @torch.no_grad()
    def forward(self, x, y, **kwargs):
        b, c, h1, w1 = x.shape
        _, _, h2, w2 = y.shape
        xs = x.float().flatten(2).transpose(1, 2).contiguous()
        ys = y.float().flatten(2).transpose(1, 2).contiguous()
        K_yy = ops.cos_kernel(ys, ys, T=self.K.T, diag_add=self.sigma_noise)
        K_xy = ops.cos_kernel(xs, ys, T=self.K.T)
        mu = self.posterior_product(K_xy, ops.spd_solve(K_yy, self.basis(1, h2, w2, x.device)))
        return mu.transpose(1, 2).reshape(b, self.dim, h1, w1)

    def posterior_rows(self, xs, ys, h2, w2, fp64=False, batch_shift=0, out=None, check="now"):
        """Token-major variant used by the Decoder: xs, ys (B,N,D) rows (any supported dtype, channels-last slices are read in
        place) -> mu (B,N,gp_dim) fp32.  ys=None with batch_shift=s: y[b] = x[(b+s) % B] (forward_symmetric) — then K_yy of
        item b is K_xx of item (b+s) % B, so ONE self-kernel matrix and one solve serve both directions.

        fp64=True (diagnostic / the `gp_precision="fp64"` parity mode): K_yy + sigma I has condition number ~1e4 on real
        features, so ANY fp32 evaluation — the reference's own CPU path included — carries ~1e-3 error on mu (measured
        against an fp64 run: reference fp32 9e-4, fp32 MFMA kernel + blocked solve 1.9e-3, torch.linalg.inv in fp32 on the
        GPU 2.2e-3).  Evaluating kernel matrices and solve in fp64 leaves only the reference's own error in the comparison.
        The product path (fp64=False) is the fp32 MFMA CosKernel + hand-blocked Cholesky."""
        B, n = xs.shape[0], xs.shape[1]
        s = batch_shift % B
        basis = self.basis(1, h2, w2, xs.device)
        if fp64:
            a = xs.double()
            c = (a.roll(-s, 0) if s else a) if ys is None else ys.double()

            def cosk(u, v):
                g = torch.einsum("bnd,bmd->bnm", u, v) / (u.norm(dim=-1)[..., None] * v.norm(dim=-1)[:, None] + 1e-6)
                return ((g - 1.0) / self.K.T).exp()

            Kyy = cosk(c, c) + self.sigma_noise * torch.eye(c.shape[1], device=c.device, dtype=torch.float64)
            Z = torch.cholesky_solve(basis.double().expand(B, -1, -1), torch.linalg.cholesky(Kyy))
            mu = (cosk(a, c) @ Z).float()
            if out is not None:
                out.copy_(mu)
                return out
            return mu
        if ys is None:
            K_self = ops.cos_kernel(xs, xs, T=self.K.T, diag_add=self.sigma_noise)
            K_xy = ops.cos_kernel(xs, xs, T=self.K.T, batch_shift=s)
            Z = ops.spd_solve(K_self, basis, check=check)            # Z[b] belongs to K_yy of item (b - s) % B
            if s:
                assert 2 * s == B, "only the half swap of forward_symmetric"
                Z = torch.cat((Z[s:], Z[:s]), dim=0)
        else:
            K_yy = ops.cos_kernel(ys, ys, T=self.K.T, diag_add=self.sigma_noise)
            K_xy = ops.cos_kernel(xs, ys, T=self.K.T)
            Z = ops.spd_solve(K_yy, basis, check=check)
        return self.posterior_product(K_xy, Z, out)

    @staticmethod
    def posterior_product(K_xy, Z, out=None):
        """mu = K_xy @ Z (matcher.py:263) accumulated in fp64.  This product is where the fp32 GP loses its accuracy: 1600 terms
        of magnitude up to |K||Z| ~ 16 cancel down to |mu| ~ 2, so fp32 ACCUMULATION alone costs 1.9e-3 on mu — measured with
        exact Z and exactly rounded K_xy (tools/scratch/gp_err.py: fp32-MFMA kernel matrices + fp32 Cholesky + fp32 product
        2.1e-3, the reference's literal fp32 method on the GPU 1.9e-3, its CPU path 9e-4), while the kernel matrices and the
        solve contribute ~2e-4 together.  The operands stay the fp32 kernel outputs; only the accumulator is wider (one library
        fp64 GEMM, 5.2 GFLOP per pair)."""
        mu = torch.bmm(K_xy.double(), Z.double())
        if out is None:
            return mu.float()
        out.copy_(mu)
        return out


# ------------------------------------------------------------------------------------------------
# Decoder — matcher.py:275-422
# ------------------------------------------------------------------------------------------------
class Decoder(nn.Module):
    def __init__(self, embedding_decoder, gps, proj, conv_refiner, detach=True, scales=("16", "8", "4", "2", "1"),
                 amp_dtype=torch.float16, **_):
        super().__init__()
        self.embedding_decoder, self.gps, self.proj, self.conv_refiner = embedding_decoder, gps, proj, conv_refiner
        self.scales = list(scales)
        self.refine_init = 4
        self.amp_dtype = amp_dtype
        # GP arithmetic: "fp32" = the product kernels (fp32-MFMA CosKernel + blocked Cholesky, every amp dtype); "fp64" = torch
        # fp64 (diagnostic: removes this side's share of the ill-conditioned solve's error from a parity comparison)
        self.gp_precision = "fp32"
        self.record = None       # a dict here collects diagnostics of the next forward (scale-16 arg-max indices, GP posterior)
        # DIAGNOSTIC hook (bench.py's roofline_pipeline_coherent): callable(scale, b, h, w, device, upsample) -> (b,2,h,w) fp32 flow
        # or None; when set, the flow entering that level's refiner is REPLACED by it (so warp and local_correlation see the spatially
        # coherent flow of a converged matcher instead of what random weights predict).  Never set in the product path.
        self.flow_override = None
        self._proj = None

    def _embedding_decoder_for(self, dtype):
        if dtype == torch.float32:
            return self.embedding_decoder
        key = (dtype, _params_version(self.embedding_decoder))
        hit = self.__dict__.get("_embed_cast")
        if hit is None or hit[0] != key:
            import copy
            with torch.inference_mode(False), torch.no_grad():
                cp = copy.deepcopy(self.embedding_decoder).eval()
                for m in cp.modules():                      # autocast semantics: 16-bit GEMM weights, fp32 LayerNorm parameters
                    if isinstance(m, nn.Linear):
                        m.to(dtype)
                hit = (key, cp)
            self.__dict__["_embed_cast"] = hit
        return hit[1]

    def get_placeholder_flow(self, b, h, w, device):
        return pixel_grid(b, h, w, device)

    def folded_proj(self, dtype):
        key = (dtype, _params_version(self.proj))
        if self._proj is not None and self._proj[0] == key:
            return self._proj[1]
        out = {}
        for s, seq in self.proj.items():
            conv, bn = seq[0], seq[1]
            g = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            wt = (conv.weight.float().reshape(conv.out_channels, -1) * g[:, None]).t().contiguous().to(dtype)   # (in,out)
            b = ((conv.bias.float() - bn.running_mean.float()) * g + bn.bias.float()).to(dtype)
            out[s] = (wt, b)
            if wt.shape[1] < 16:
                out[s + "/16"] = (F.pad(wt, (0, 16 - wt.shape[1])).contiguous(), F.pad(b, (0, 16 - b.shape[0])).contiguous())
        self._proj = (key, out)
        return out

    def project(self, s, f, dtype, out=None):
        """1x1 conv + BN(eval) as one batched GEMM — matcher.py:366-371.  The NCHW feature map is the GEMM's transposed
        operand (no layout copy) and the result lands channels-last, directly in `out` when given (a (B,h,w,C) slice of the
        refiner's concat buffer).  Returns the (B,C,h,w) channels-last view."""
        wt, b = self.folded_proj(dtype)[s]
        B, C, h, w = f.shape
        a = f.to(dtype).flatten(2).transpose(1, 2)                                            # (B, hw, Cin) view
        n = wt.shape[1]
        if out is None:
            out = torch.empty((B, h, w, n), dtype=dtype, device=f.device)
        tgt = out
        if n < 16 and out.stride(2) >= 16 and out.storage_offset() == 0:
            # the 64 -> 9 projection of scale 1: hipBLASLt runs N = 16 25 % faster than N = 9; the 7 extra (zero) columns
            # land in channels the warp / embedding kernels overwrite afterwards
            tgt = out.as_strided((B, h, w, 16), out.stride())
            wt, b = self.folded_proj(dtype)[s + "/16"]
        torch.baddbmm(b, a, wt.unsqueeze(0).expand(B, -1, -1), out=tgt.view(B, h * w, tgt.shape[-1]))
        return out.permute(0, 3, 1, 2)

    @torch.no_grad()
    def forward(self, f1, f2, gt_warp=None, gt_prob=None, upsample=False, flow=None, certainty=None, scale_factor=1,
                swapped_pair=False, wait_events=None):
        """Coarse-to-fine loop.  `swapped_pair=True` tells the decoder that f2 is f1 with its two batch halves swapped
        (forward_symmetric), so each level is projected once."""
        dtype = self.amp_dtype
        all_scales = self.scales if not upsample else ["8", "4", "2", "1"]                    # matcher.py:335
        sizes = {s: tuple(f1[s].shape[-2:]) for s in f1}
        h, w = sizes[1]
        b = f1[1].shape[0]
        device = f1[1].device
        first = int(all_scales[0])
        if not upsample:
            flow = pixel_grid(b, *sizes[first], device)                                      # :346
            certainty = None                                                                 # the reference's 0.0
        else:
            flow = ops.interp_bilinear(flow, sizes[first])                                   # :349-360
            certainty = ops.interp_bilinear(certainty, sizes[first])
        corresps = {}
        for s in all_scales:
            ins = int(s)
            corresps[ins] = {}
            if wait_events and ins in wait_events:          # features of this level (and below) come from another stream
                torch.cuda.current_stream().wait_event(wait_events[ins])
            hs, ws = sizes[ins]
            refiner = self.conv_refiner[s]
            buf = refiner.new_buffer(b, hs, ws, dtype, device)                               # x lands in its first channels
            x = self.project(s, f1[ins], dtype, out=buf[..., :self.proj[s][0].out_channels])
            shift = 0
            if swapped_pair:
                y, shift = x, b // 2                        # y[i] = x[(i + b/2) % b]: the kernels index the swap, nothing is copied
            else:
                y = self.project(s, f2[ins], dtype)
            if ins in self.embedding_decoder.scales():
                n = hs * ws
                xs = x.permute(0, 2, 3, 1).reshape(b, n, -1)                                   # channels-last rows, read in place
                ys = None if swapped_pair else y.permute(0, 2, 3, 1).reshape(b, n, -1)
                gp = self.gps[s]
                C = xs.shape[2]
                # tokens = cat(gp_posterior [fp32], features) -> fp32 (transformer/__init__.py:33-41): the GP writes mu straight
                # into the (row-padded) fp32 token buffer
                npad = n if dtype == torch.float32 else n + (-n % 128)
                tokens = torch.empty((b, npad, gp.dim + C), dtype=torch.float32, device=device)
                if npad > n:
                    tokens[:, n:].zero_()
                tokens[:, :n, gp.dim:].copy_(xs)
                capturing = torch.cuda.is_current_stream_capturing()
                gp.posterior_rows(xs, ys, hs, ws, fp64=(self.gp_precision == "fp64"), batch_shift=shift,
                                  out=tokens[:, :n, :gp.dim],
                                  check=None if capturing else ("defer" if ops.deferring() else "now"))   # :377
                # The reference runs this transformer under autocast (transformer/__init__.py:31-32): fp32 token / residual
                # stream and LayerNorm, amp-dtype GEMMs / attention / logits.  Autocast re-casts every weight on every call
                # (0.26 ms of copy kernels per pair), so the 16-bit modes run a cached copy of the module whose Linear weights
                # are in the amp dtype; the stream stays fp32 (ops.add_layernorm fuses residual add + LayerNorm + cast).
                rows = self._embedding_decoder_for(dtype).forward_rows(tokens, n_valid=n, owned=True)   # (b, hw, 4097)
                flow, certainty = ops.cls_rows_to_flow(rows, b, hs, ws)                       # :378-385
                if self.record is not None and not upsample:
                    self.record["argmax16"] = rows[..., :-1].float().argmax(dim=-1).cpu()     # (b, hw) anchor index (utils.py:316)
                    self.record["mu16"] = tokens[:, :n, :gp.dim].detach().clone()
            if self.flow_override is not None:
                forced = self.flow_override(ins, b, hs, ws, device, upsample)
                if forced is not None:
                    flow = forced
            flow, certainty = refiner.forward_update(                                         # :393-402
                x, y, flow, certainty, scale_factor, ins / (self.refine_init * w), ins / (self.refine_init * h), dtype=dtype, buf=buf,
                batch_shift=shift)
            corresps[ins].update({"certainty": certainty, "flow": flow})
            if s != "1":
                flow = ops.interp_bilinear(flow, sizes[ins // 2])                             # :408-417
                certainty = ops.interp_bilinear(certainty, sizes[ins // 2])
        return corresps


# ------------------------------------------------------------------------------------------------
# RegressionMatcher — matcher.py:425-766
# ------------------------------------------------------------------------------------------------
def _check_rgb(im):
    if im.mode != "RGB":
        raise NotImplementedError("Can't handle non-RGB images")                              # utils.py:660-662


def _check_not_i16(im):
    if im.mode == "I;16":
        raise NotImplementedError("Can't handle 16 bit images")                               # utils.py:656-658


def preprocess(im, size):
    """PIL RGB -> PIL bicubic resize to (h,w) -> /255 -> ImageNet normalise -> (3,h,w) fp32 (utils.py:165-261)."""
    from PIL import Image
    h, w = size
    a = np.array(im.resize((w, h), Image.BICUBIC), dtype=np.float32).transpose(2, 0, 1) / 255.0
    t = torch.from_numpy(a)
    return (t - torch.tensor(IMAGENET_MEAN)[:, None, None]) / torch.tensor(IMAGENET_STD)[:, None, None]


class RegressionMatcher(nn.Module):
    def __init__(self, encoder, decoder, h=448, w=448, sample_mode="threshold_balanced", upsample_preds=False,
                 symmetric=False, name=None, attenuate_cert=None):
        super().__init__()
        self.attenuate_cert = attenuate_cert
        self.encoder, self.decoder = encoder, decoder
        self.name = name
        self.w_resized, self.h_resized = w, h
        self.sample_mode = sample_mode
        self.upsample_preds = upsample_preds
        self.upsample_res = (14 * 16 * 6, 14 * 16 * 6)
        self.symmetric = symmetric
        self.sample_thresh = 0.05
        self.host_preprocess = False          # True: PIL resize on the host like the reference (same bits, slower)
        self.device_jpeg = True               # JPEG paths: entropy decoding on the host, IDCT / up-sampling / RGB on the device (same bits as PIL)

    def get_output_resolution(self):
        return self.upsample_res if self.upsample_preds else (self.h_resized, self.w_resized)

    def extract_backbone_features(self, batch, batched=True, upsample=False):
        if batched:
            return self.encoder(torch.cat((batch["im_A"], batch["im_B"]), dim=0), upsample=upsample)
        return self.encoder(batch["im_A"], upsample=upsample), self.encoder(batch["im_B"], upsample=upsample)

    def forward(self, batch, batched=True, upsample=False, scale_factor=1):
        pyr = batch.get("pyramid") or self.extract_backbone_features(batch, batched=batched, upsample=upsample)
        if batched:
            f_q = {s: f.chunk(2)[0] for s, f in pyr.items()}
            f_s = {s: f.chunk(2)[1] for s, f in pyr.items()}
        else:
            f_q, f_s = pyr
        return self.decoder(f_q, f_s, upsample=upsample, scale_factor=scale_factor, wait_events=batch.get("wait_events"),
                            **(batch.get("corresps") or {}))

    def forward_symmetric(self, batch, batched=True, upsample=False, scale_factor=1):
        pyr = batch.get("pyramid") or self.extract_backbone_features(batch, batched=batched, upsample=upsample)   # matcher.py:516-528
        kw = {k: v for k, v in (batch.get("corresps") or {}).items() if k in ("flow", "certainty")}
        return self.decoder(pyr, None, upsample=upsample, scale_factor=scale_factor, swapped_pair=True,
                            wait_events=batch.get("wait_events"), **kw)

    # -- sampling --------------------------------------------------------------------------------
    def sample(self, matches, certainty, num=10000, seed=None):
        """matcher.py:468-495: certainty-thresholded draw of 4*num matches without replacement, KDE-balanced re-draw of num.

        Both of the reference's `torch.multinomial(w, k, replacement=False)` calls are exponential races (key = w / Exp(1),
        keep the k largest — also what ATen does internally); here the threshold, a counter-based uniform and the key are one
        kernel (ops.race_keys) and the selection is a top-k.  `seed` makes the draw a pure function of (inputs, seed) that the
        CPU oracle reproduces (oracle.sample_seeded); seed=None takes one from torch's CPU generator (torch.manual_seed governs
        it, no device sync).  Matches are returned in draw order."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        thresh = self.sample_thresh if "threshold" in self.sample_mode else -1.0
        matches, certainty = matches.reshape(-1, 4), certainty.reshape(-1)
        expansion = 4 if "balanced" in self.sample_mode else 1
        good = torch.topk(ops.race_keys(certainty, thresh, seed), min(expansion * num, len(certainty))).indices
        good_matches, good_certainty = matches[good], certainty[good]
        if thresh >= 0:
            good_certainty = torch.where(good_certainty > thresh, torch.ones_like(good_certainty), good_certainty)   # :474-477
        if "balanced" not in self.sample_mode:
            return good_matches, good_certainty
        density = ops.kde(good_matches, std=0.1)                                              # fp16, like the reference (:489)
        p = 1 / (density + 1)
        p[density < 10] = 1e-7
        bal = torch.topk(ops.race_keys(p, -1.0, seed, counter=good, stage=1), min(num, len(good_certainty))).indices
        return good_matches[bal], good_certainty[bal]

    # -- coordinates -----------------------------------------------------------------------------
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

    def to_normalized_coordinates(self, coords, H_A, W_A, H_B, W_B):
        if isinstance(coords, (list, tuple)):
            kA, kB = coords[0], coords[1]
        else:
            kA, kB = coords[..., :2], coords[..., 2:]
        kA = torch.stack((2 / W_A * kA[..., 0] - 1, 2 / H_A * kA[..., 1] - 1), axis=-1)
        kB = torch.stack((2 / W_B * kB[..., 0] - 1, 2 / H_B * kB[..., 1] - 1), axis=-1)
        return kA, kB

    def match_keypoints(self, x_A, x_B, warp, certainty, return_tuple=True, return_inds=False):
        """matcher.py:576-591: warp the A keypoints, mutual nearest neighbours among x_B, certainty gate.  The reference builds
        cdist(x_A_to_B, x_B) (|x_A| x |x_B| floats) and compares it with its row and column minima; here the two arg-min
        vectors come from one small kernel each (ops.nn_argmin) and nothing quadratic is stored.  Pairs come out ordered by
        the A index like torch.nonzero's; on EXACT distance ties (duplicate keypoints) the lowest index wins, where the
        reference would emit every tied pair."""
        x_A_to_B = F.grid_sample(warp[..., -2:].permute(2, 0, 1)[None], x_A[None, None], align_corners=False, mode="bilinear")[0, :, 0].mT
        cert = F.grid_sample(certainty[None, None, ...], x_A[None, None], align_corners=False, mode="bilinear")[0, 0, 0]
        nn_b = ops.nn_argmin(x_A_to_B, x_B)                     # nearest B keypoint of every warped A keypoint
        nn_a = ops.nn_argmin(x_B, x_A_to_B)                     # nearest warped A keypoint of every B keypoint
        iA = torch.nonzero((nn_a[nn_b] == torch.arange(len(nn_b), device=nn_b.device)) & (cert > self.sample_thresh), as_tuple=True)[0]
        iB = nn_b[iA]
        if return_tuple:
            return (iA, iB) if return_inds else (x_A[iA], x_B[iB])
        return torch.cat((iA, iB), dim=-1) if return_inds else torch.cat((x_A[iA], x_B[iB]), dim=-1)

    # -- match -----------------------------------------------------------------------------------
    @torch.inference_mode()
    def match_tensors(self, A_lo, B_lo, A_hi=None, B_hi=None):
        """P pairs at once.  A_lo,B_lo: (P,3,h,w) normalised coarse images; A_hi,B_hi: (P,3,H,W) for the upsample
        pass.  Returns the stack of per-pair results: warp (P,H,2W,4), certainty (P,H,2W) (symmetric) — the batched
        560->864 semantics the reference leaves undefined (its batched+upsample path raises)."""
        with ops.deferred_spd_checks():      # the GP solve's SPD check: no host sync in the middle, examined when the pipeline is queued
            return self._match_tensors(A_lo, B_lo, A_hi, B_hi)

    def _match_tensors(self, A_lo, B_lo, A_hi=None, B_hi=None):
        symmetric = self.symmetric
        # Stream plan.  Only the DINOv2 features feed the first decoder stage (GP + transformer + scale-16 refiner), and
        # that stage is a chain of small, latency-bound kernels.  So the ViT runs first on the main stream, and BOTH VGG
        # pyramids (560 for the coarse levels 8..1, 864 for the upsample pass) run on a second HIP stream behind it, filling
        # the CUs the GP solve leaves idle.  The decoder waits for the 560 pyramid before its scale-8 level and for the 864
        # one before the upsample pass.
        batch = {"im_A": A_lo, "im_B": B_lo}
        hi_pyr = None
        overlap = getattr(self, "overlap_upsample_encoder", True) and hasattr(self.encoder, "vit_features")
        if overlap:
            x_lo = torch.cat((A_lo, B_lo), dim=0)
            pyr = {16: self.encoder.vit_features(x_lo)}
            main = torch.cuda.current_stream()
            side = self._side_stream = getattr(self, "_side_stream", None) or torch.cuda.Stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pyr.update(self.encoder.cnn(x_lo, self.encoder.amp_dtype))
                lo_ready = torch.cuda.Event()
                lo_ready.record(side)
                if self.upsample_preds and A_hi is not None and B_hi is not None:
                    hi_pyr = self.extract_backbone_features({"im_A": A_hi, "im_B": B_hi}, batched=True, upsample=True)
            batch["pyramid"] = pyr
            batch["wait_events"] = {8: lo_ready}
        else:
            batch["pyramid"] = self.extract_backbone_features(batch, batched=True)
        corresps = self.forward_symmetric(batch) if symmetric else self.forward(batch, batched=True)
        hs, ws = A_lo.shape[-2:]
        cert16 = corresps[16]["certainty"] if self.attenuate_cert else None
        if self.upsample_preds:
            if A_hi is None or B_hi is None:
                raise ValueError("upsample_preds=True needs the high-resolution images (A_hi, B_hi)")
            hs, ws = A_hi.shape[-2:]
            scale_factor = math.sqrt(hs * ws / (A_lo.shape[-2] * A_lo.shape[-1]))            # matcher.py:677
            batch = {"im_A": A_hi, "im_B": B_hi, "corresps": corresps[1]}
            if hi_pyr is not None:
                torch.cuda.current_stream().wait_stream(self._side_stream)
                batch["pyramid"] = hi_pyr
            corresps = (self.forward_symmetric(batch, upsample=True, scale_factor=scale_factor) if symmetric
                        else self.forward(batch, batched=True, upsample=True, scale_factor=scale_factor))
        return ops.match_finalize(corresps[1]["flow"], corresps[1]["certainty"], cert16, symmetric=symmetric)

    def make_graphed(self, A_lo, B_lo, A_hi=None, B_hi=None, warmup=2):
        """match_tensors for ONE fixed batch shape as a captured hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm): the
        ~900 launches of a step — both streams of the plan in match_tensors, every library GEMM / convolution and every
        kernel of libroma_hip.so — are recorded once and replayed with one host call, which removes the host launch gaps
        between the small dependent kernels (GP chain, decoder transformer, narrow levels).  Returns run(A_lo, B_lo, A_hi,
        B_hi) -> (warp, certainty): inputs are copied into the graph's static buffers (same shapes as here), outputs are the
        graph's static tensors (clone them if they must survive the next call).  The SPD check of the GP solve is not part
        of the graph (it needs a host read): call match_tensors itself when inputs may be degenerate."""
        ins = [None if t is None else t.clone() for t in (A_lo, B_lo, A_hi, B_hi)]
        with torch.inference_mode():
            for _ in range(warmup):                               # builds every weight cache and library plan eagerly
                self.match_tensors(*ins)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.match_tensors(*ins)

        def run(*new):
            for dst, src in zip(ins, new):
                if dst is not None and src is not None and src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
            graph.replay()
            return out

        run.graph, run.static_inputs, run.static_outputs = graph, ins, out
        return run

    @torch.inference_mode()
    def encode(self, im_lo, im_hi=None):
        """Feature pyramids of ONE image (or a stack of images), to be reused by every pair the image takes part in (SfM-style
        matching of N images in M >> N pairs runs the encoders N times instead of 2M).  im_lo (P,3,h,w)
        normalised at the coarse resolution, im_hi (P,3,H,W) at the upsample resolution (needed when upsample_preds).
        Returns {"lo": {16,8,4,2,1: feats}, "hi": {8,4,2,1: feats} or None} — extract_backbone_features (matcher.py:458-466)
        on one side of the pair."""
        lo = self.encoder(im_lo, upsample=False)
        hi = self.encoder(im_hi, upsample=True) if (self.upsample_preds and im_hi is not None) else None
        return {"lo": lo, "hi": hi, "lo_size": tuple(im_lo.shape[-2:]), "hi_size": None if im_hi is None else tuple(im_hi.shape[-2:])}

    @torch.inference_mode()
    def match_encoded(self, enc_A, enc_B):
        """match_tensors on cached pyramids (see encode): the decoder passes and the post-processing only.  The two pyramids
        are concatenated along the batch ([A; B], the layout forward_symmetric works on): one copy of the features, a few
        hundred microseconds, against ~11 ms of encoder time per pair."""
        if not self.symmetric:
            raise NotImplementedError("match_encoded implements the symmetric mode the shipped models use")
        with ops.deferred_spd_checks():
            return self._match_encoded(enc_A, enc_B)

    def _match_encoded(self, enc_A, enc_B):
        pyr = {s: torch.cat((enc_A["lo"][s], enc_B["lo"][s]), dim=0) for s in enc_A["lo"]}
        corresps = self.forward_symmetric({"pyramid": pyr})
        cert16 = corresps[16]["certainty"] if self.attenuate_cert else None
        if self.upsample_preds:
            if enc_A["hi"] is None or enc_B["hi"] is None:
                raise ValueError("upsample_preds=True needs encode(im_lo, im_hi)")
            (h, w), (hs, ws) = enc_A["lo_size"], enc_A["hi_size"]
            pyr = {s: torch.cat((enc_A["hi"][s], enc_B["hi"][s]), dim=0) for s in enc_A["hi"]}
            corresps = self.forward_symmetric({"pyramid": pyr, "corresps": corresps[1]}, upsample=True,
                                              scale_factor=math.sqrt(hs * ws / (h * w)))             # matcher.py:677
        return ops.match_finalize(corresps[1]["flow"], corresps[1]["certainty"], cert16, symmetric=True)

    @torch.inference_mode()
    def match(self, im_A_input, im_B_input, *args, batched=False, device=None, im_A_hi=None, im_B_hi=None):
        """matcher.py:593-730.  Paths or RGB PIL images (one pair) -> warp (H,2W,4), certainty (H,2W); extra positional
        arguments are ignored like the reference does.  batched=True takes normalised tensors (plus im_A_hi/im_B_hi
        when upsample_preds) and returns the stacked per-pair results."""
        from PIL import Image
        if device is None:
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("roma_amd runs on an MI355X only; there is no CPU path (use the oracle for CPU numbers)")
        self.train(False)
        if batched:
            b, c, h, w = im_A_input.shape
            b2, c2, h2, w2 = im_B_input.shape
            assert w == w2 and h == h2, "For batched images we assume same size"
            if h != self.h_resized or self.w_resized != w:
                warn("Model resolution and batch resolution differ, may produce unexpected results")
            hi = (None, None) if im_A_hi is None else (im_A_hi.to(device), im_B_hi.to(device))
            return self.match_tensors(im_A_input.to(device), im_B_input.to(device), *hi)
        ims = []
        for im in (im_A_input, im_B_input):
            if isinstance(im, (str, os.PathLike)):
                if not getattr(self, "host_preprocess", False) and getattr(self, "device_jpeg", True):
                    # JPEG files: Huffman decoding on the host, everything after it on the device, bit-identical to PIL
                    # (preproc.decode_jpeg_device; None = a stream the kernels do not cover -> PIL below)
                    from .preproc import decode_jpeg_device
                    dec = decode_jpeg_device(im, device) if str(im).lower().endswith((".jpg", ".jpeg")) else None
                    if dec is not None:
                        ims.append(dec)
                        continue
                im = Image.open(im)
                _check_not_i16(im)
                im = im.convert("RGB")
            else:
                _check_rgb(im)
            ims.append(im)
        if getattr(self, "host_preprocess", False):
            # the reference's way: PIL resizes each image once per resolution on the host, fp32 tensors cross PCIe
            lo = [preprocess(im, (self.h_resized, self.w_resized))[None].to(device) for im in ims]
            hi = [preprocess(im, self.upsample_res)[None].to(device) for im in ims] if self.upsample_preds else [None, None]
        else:
            # one uint8 upload per photograph; both resolutions are resized + normalised on the device, bit-identical to PIL
            from .preproc import preprocess_device
            src = [im if isinstance(im, torch.Tensor) else torch.from_numpy(np.array(im, dtype=np.uint8)).to(device) for im in ims]
            lo = [preprocess_device(s, (self.h_resized, self.w_resized), device)[None] for s in src]
            hi = [preprocess_device(s, self.upsample_res, device)[None] for s in src] if self.upsample_preds else [None, None]
        warp, cert = self.match_tensors(lo[0], lo[1], hi[0], hi[1])
        return warp[0], cert[0]

    def visualize_warp(self, warp, certainty, im_A=None, im_B=None, im_A_path=None, im_B_path=None, device="cuda",
                       symmetric=True, save_path=None, unnormalize=False):
        """matcher.py:732-766 (host-side visualisation helper; not on the timed path)."""
        from PIL import Image
        H, W2, _ = warp.shape
        W = W2 // 2 if symmetric else W2
        if im_A is None:
            im_A, im_B = Image.open(im_A_path).convert("RGB"), Image.open(im_B_path).convert("RGB")
        if not isinstance(im_A, torch.Tensor):
            im_A, im_B = im_A.resize((W, H)), im_B.resize((W, H))
            x_B = (torch.tensor(np.array(im_B)) / 255).to(device).permute(2, 0, 1)
            x_A = (torch.tensor(np.array(im_A)) / 255).to(device).permute(2, 0, 1)
        else:
            x_A, x_B = im_A, im_B
        a2b = F.grid_sample(x_B[None], warp[:, :W, 2:][None], mode="bilinear", align_corners=False)[0]
        if symmetric:
            b2a = F.grid_sample(x_A[None], warp[:, W:, :2][None], mode="bilinear", align_corners=False)[0]
            warp_im = torch.cat((a2b, b2a), dim=2)
        else:
            warp_im = a2b
        white = torch.ones((H, W2 if symmetric else W), device=device)
        vis = certainty * warp_im + (1 - certainty) * white
        if save_path is not None:
            arr = (vis.permute(1, 2, 0).clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)
            Image.fromarray(arr).save(save_path)
        return vis
