#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden.py [--only NAME ...] [--full]

The reference (techshoww/RoMa, /root/reference, read-only) is imported unchanged.  Three third-party
packages it imports at module scope are absent from this image and are NOT on the arithmetic path of
inference (cv2, kornia) or are tiny and restated here from their published behaviour (torchvision:
``transforms.Resize`` on a PIL image = ``PIL.Image.resize(BICUBIC)``, ``transforms.Normalize`` =
``(x-mean)/std``, ``models.vgg19_bn`` = the cfg-"E" conv3x3+BN+ReLU / maxpool stack).  SURVEY.md §8(c).

Only the OUTPUT tensors (and a few small inputs) are written; inputs and weights are regenerated in
the tests from tests/golden/recipes.py.  The reference's source never leaves /root/reference.
This script is not run on the GPU box and is not imported by any test.
"""
from __future__ import annotations

import argparse
import enum
import math
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipes as R  # noqa: E402
from cases import *  # noqa: E402,F401,F403
from cases import RED, DEC_GAINS, E2E_GAINS, LOCAL_CORR_CASES, REFINER_CASES  # noqa: E402

REF = "/root/reference"


# ------------------------------------------------------------------------------------------------
# shims for absent third-party modules (see module docstring)
# ------------------------------------------------------------------------------------------------
def install_shims():
    from PIL import Image

    for name in ("cv2", "kornia", "kornia.augmentation", "kornia.geometry", "kornia.geometry.transform"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["kornia.geometry.transform"].warp_perspective = None

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvm = types.ModuleType("torchvision.models")

    class InterpolationMode(enum.Enum):
        NEAREST = "nearest"
        BILINEAR = "bilinear"
        BICUBIC = "bicubic"

    _pil_mode = {InterpolationMode.BICUBIC: Image.BICUBIC, InterpolationMode.BILINEAR: Image.BILINEAR,
                 InterpolationMode.NEAREST: Image.NEAREST}

    class Resize:
        def __init__(self, size, interpolation=InterpolationMode.BILINEAR):
            self.size, self.interpolation = size, interpolation

        def __call__(self, im):
            h, w = self.size
            return im.resize((w, h), _pil_mode[self.interpolation])

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype)[:, None, None]
            s = torch.tensor(self.std, dtype=x.dtype)[:, None, None]
            return (x - m) / s

    class ToTensor:
        def __call__(self, im):
            return torch.from_numpy(np.array(im, dtype=np.float32).transpose(2, 0, 1) / 255.0)

    def vgg19_bn(pretrained=False, **kw):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
        layers, c = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    tvt.Resize, tvt.Normalize, tvt.ToTensor = Resize, Normalize, ToTensor
    tvf.InterpolationMode = InterpolationMode
    tvt.functional = tvf
    tvm.vgg19_bn = vgg19_bn
    tvm.resnet50 = None
    tv.transforms, tv.models = tvt, tvm
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "torchvision.models": tvm})


def import_reference():
    install_shims()
    sys.path.insert(0, REF)
    import romatch  # noqa: F401
    return romatch


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def load_filled(module: nn.Module, prefix: str, seed=0, gains=None):
    """Overwrite every parameter/buffer of `module` with the recipe value for '<prefix><key>'."""
    sd = module.state_dict()
    new = {k: T(R.fill_tensor(prefix + k, v.shape, seed, gains)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


# ------------------------------------------------------------------------------------------------
# 1. local_correlation  (romatch/utils/local_correlation.py:4-48)
# ------------------------------------------------------------------------------------------------
def gen_local_corr(ref):
    from romatch.utils.local_correlation import local_correlation
    out = {}
    for name in LOCAL_CORR_CASES:
        f0, f1, flow, r = local_corr_inputs(name)
        o = local_correlation(T(f0), T(f1), local_radius=r, flow=None if flow is None else T(flow))
        out[name] = o.numpy()
    # full-size L16 call of the 560 coarse pass: inputs from the recipe, output pinned by statistics + a strided sample
    B, C, h, w, r = 2, 512, 40, 40, 7
    f0 = R.normal("lc.full16.f0", (B, C, h, w))
    f1 = R.normal("lc.full16.f1", (B, C, h, w))
    flow = R.coherent_flow("lc.full16.flow", B, h, w)
    o = local_correlation(T(f0), T(f1), local_radius=r, flow=T(flow)).numpy()
    out["full16_stats"] = R.checksum(o)
    out["full16_sample"] = o[:, ::7, ::3, ::3].copy()
    save("local_corr", **out)


# ------------------------------------------------------------------------------------------------
# 2. cls_to_flow_refine  (romatch/utils/utils.py:301-323)
# ------------------------------------------------------------------------------------------------
def gen_cls(ref):
    from romatch.utils.utils import cls_to_flow_refine
    x = cls_inputs()
    save("cls_to_flow_refine", flow=cls_to_flow_refine(T(x)).numpy(),
         flow_small=cls_to_flow_refine(T(cls_inputs(res=8, B=1, H=5, W=7))).numpy())


# ------------------------------------------------------------------------------------------------
# 3. CosKernel / GP.forward  (romatch/models/matcher.py:145-273)
# ------------------------------------------------------------------------------------------------
def gen_gp(ref):
    from romatch.models.matcher import GP, CosKernel
    out = {}
    for name, (b, c, h, w, d) in {"small": (1, 512, 10, 10, 512), "rect": (2, 64, 6, 9, 32)}.items():
        gp = GP(CosKernel, T=0.2, learn_temperature=False, only_attention=False, gp_dim=d, basis="fourier", no_cov=True)
        load_filled(gp, f"gp.{name}.", gains={"pos_conv": 4.0})
        x = R.normal(f"gp.{name}.x", (b, c, h, w))
        y = R.normal(f"gp.{name}.y", (b, c, h, w))
        with torch.no_grad():
            out[f"{name}_mu"] = gp(T(x), T(y)).numpy()
            k = CosKernel(T=0.2)
            xs = T(x).flatten(2).permute(0, 2, 1)
            ys = T(y).flatten(2).permute(0, 2, 1)
            out[f"{name}_Kxy"] = k(xs, ys).numpy()
    save("gp", **out)


# ------------------------------------------------------------------------------------------------
# 4. ConvRefiner.forward  (romatch/models/matcher.py:17-143)
# ------------------------------------------------------------------------------------------------
def build_refiner(cls, name):
    fd, ed, r, *_ = REFINER_CASES[name]
    D = 2 * fd + ed + ((2 * r + 1) ** 2 if r else 0)
    m = cls(D, D, 3, kernel_size=5, dw=True, hidden_blocks=2, displacement_emb="linear", displacement_emb_dim=ed,
            local_corr_radius=r, corr_in_other=True if r else None, amp=True, disable_local_corr_grad=True,
            bn_momentum=0.01)
    return m.eval()


def gen_refiner(ref):
    from romatch.models.matcher import ConvRefiner
    out = {}
    for name in REFINER_CASES:
        m = load_filled(build_refiner(ConvRefiner, name), f"ref.{name}.")
        x, y, flow, sf = refiner_inputs(name)
        with torch.no_grad():
            d, c = m(T(x), T(y), T(flow), scale_factor=sf)
        out[f"{name}_dflow"], out[f"{name}_dcert"] = d.numpy(), c.numpy()
    save("conv_refiner", **out)


# ------------------------------------------------------------------------------------------------
# 5. Decoder.forward, both modes, reduced pyramid  (romatch/models/matcher.py:275-422)
# ------------------------------------------------------------------------------------------------
def build_reduced_decoder(mods):
    ConvRefiner, GP, CosKernel, Decoder, TransformerDecoder, Block, MemEffAttention = mods
    dd = RED["gp"] + RED["feat16"]
    td = TransformerDecoder(nn.Sequential(*[Block(dd, RED["heads"], attn_class=MemEffAttention) for _ in range(RED["nblk"])]),
                            dd, RED["cls_res"] ** 2 + 1, is_classifier=True, amp=True, pos_enc=False)
    feat = dict(RED["feat"])
    feat[16] = RED["feat16"]
    refiners = {}
    for s in (16, 8, 4, 2, 1):
        r = RED["rad"][s]
        D = 2 * feat[s] + RED["emb"][s] + ((2 * r + 1) ** 2 if r else 0)
        refiners[str(s)] = ConvRefiner(D, D, 3, kernel_size=5, dw=True, hidden_blocks=2, displacement_emb="linear",
                                       displacement_emb_dim=RED["emb"][s], local_corr_radius=r,
                                       corr_in_other=True if r else None, amp=True, disable_local_corr_grad=True,
                                       bn_momentum=0.01)
    gps = nn.ModuleDict({"16": GP(CosKernel, T=0.2, learn_temperature=False, only_attention=False, gp_dim=RED["gp"],
                                  basis="fourier", no_cov=True)})
    vgg = RED["vgg"]
    proj = nn.ModuleDict({
        "16": nn.Sequential(nn.Conv2d(RED["dino"], feat[16], 1, 1), nn.BatchNorm2d(feat[16])),
        "8": nn.Sequential(nn.Conv2d(vgg[8], feat[8], 1, 1), nn.BatchNorm2d(feat[8])),
        "4": nn.Sequential(nn.Conv2d(vgg[4], feat[4], 1, 1), nn.BatchNorm2d(feat[4])),
        "2": nn.Sequential(nn.Conv2d(vgg[2], feat[2], 1, 1), nn.BatchNorm2d(feat[2])),
        "1": nn.Sequential(nn.Conv2d(vgg[1], feat[1], 1, 1), nn.BatchNorm2d(feat[1])),
    })
    dec = Decoder(td, gps, proj, nn.ModuleDict(refiners), detach=True, scales=["16", "8", "4", "2", "1"],
                  displacement_dropout_p=0.0, gm_warp_dropout_p=0.0)
    return dec.eval()


def gen_decoder(ref):
    from romatch.models.matcher import ConvRefiner, GP, CosKernel, Decoder
    from romatch.models.transformer import TransformerDecoder, Block, MemEffAttention
    dec = load_filled(build_reduced_decoder((ConvRefiner, GP, CosKernel, Decoder, TransformerDecoder, Block, MemEffAttention)),
                      "dec.", gains=DEC_GAINS)
    out = {}
    b = 2
    f1, f2 = decoder_pyramids("coarse", b, 112, upsample=False)
    with torch.no_grad():
        c = dec({s: T(v) for s, v in f1.items()}, {s: T(v) for s, v in f2.items()})
    for s in c:
        out[f"coarse_flow_{s}"] = c[s]["flow"].numpy()
        out[f"coarse_cert_{s}"] = c[s]["certainty"].numpy()
    g1, g2 = decoder_pyramids("up", b, 176, upsample=True)
    sf = math.sqrt(176 * 176 / (112 * 112))
    with torch.no_grad():
        u = dec({s: T(v) for s, v in g1.items()}, {s: T(v) for s, v in g2.items()}, upsample=True,
                flow=c[1]["flow"], certainty=c[1]["certainty"], scale_factor=sf)
    for s in u:
        out[f"up_flow_{s}"] = u[s]["flow"].numpy()
        out[f"up_cert_{s}"] = u[s]["certainty"].numpy()
    save("decoder", **out)


# ------------------------------------------------------------------------------------------------
# 6. match() post-processing  (romatch/models/matcher.py:653-730) on synthetic corresps
# ------------------------------------------------------------------------------------------------
def gen_post(ref):
    from romatch.models.matcher import RegressionMatcher
    from PIL import Image
    c16, flow, cert = post_inputs()

    class Dummy(nn.Module):
        def forward(self, *a, **k):
            raise RuntimeError

    m = RegressionMatcher(Dummy(), Dummy(), h=20, w=24, upsample_preds=True, symmetric=True, attenuate_cert=True,
                          sample_mode="threshold_balanced")
    m.upsample_res = (30, 36)
    calls = []

    def fake_forward_symmetric(batch, batched=True, upsample=False, scale_factor=1):
        calls.append((upsample, scale_factor, tuple(batch["im_A"].shape)))
        if not upsample:
            return {16: {"certainty": T(c16)}, 1: {"flow": T(flow) * 0 + 7, "certainty": T(cert) * 0 + 7}}
        return {1: {"flow": T(flow), "certainty": T(cert)}}

    m.forward_symmetric = fake_forward_symmetric
    im = Image.fromarray((R.uniform("post.im", (40, 50, 3), 0, 255)).astype(np.uint8))
    warp, certainty = m.match(im, im, device="cpu")
    save("match_post", warp=warp.numpy(), certainty=certainty.numpy(),
         scale_factor=np.float64(calls[1][1]))


# ------------------------------------------------------------------------------------------------
# 7. kde + deterministic part of sample()  (romatch/utils/kde.py:4-12, matcher.py:468-495)
# ------------------------------------------------------------------------------------------------
def gen_kde(ref):
    from romatch.utils.kde import kde
    x = T(kde_inputs())
    save("kde", fp32=kde(x, std=0.1, half=False).numpy(), fp16=kde(x, std=0.1, half=True).float().numpy(),
         down4=kde(x, std=0.1, half=False, down=4).numpy())


# ------------------------------------------------------------------------------------------------
# 8. pre-processing of the bundled JPEG pair (romatch/utils/utils.py:165-261)
# ------------------------------------------------------------------------------------------------
def gen_preproc(ref):
    from romatch.utils import get_tuple_transform_ops
    from PIL import Image
    out = {}
    ims = [Image.open(os.path.join(HERE, "assets", f"sacre_coeur_{n}.jpg")).convert("RGB") for n in "AB"]
    for res in (560, 864):
        a, b = get_tuple_transform_ops(resize=(res, res), normalize=True)(ims)
        for n, t in (("A", a), ("B", b)):
            out[f"{n}{res}_stats"] = R.checksum(t.numpy())
            out[f"{n}{res}_sample"] = t.numpy()[:, ::16, ::16].copy()
    save("preproc", **out)


# ------------------------------------------------------------------------------------------------
# 9. end-to-end match(): the shipped architecture (ViT-L/14 + VGG19-BN + full decoder) at reduced resolution
# ------------------------------------------------------------------------------------------------
def build_full_reference(res, up_res, seed=0):
    from romatch.models.model_zoo.roma_models import roma_model
    from romatch.models.transformer import vit_large
    from romatch.models.matcher import RegressionMatcher
    vit_shapes = {k: tuple(v.shape) for k, v in
                  vit_large(img_size=518, patch_size=14, init_values=1.0, ffn_layer="mlp", block_chunks=0).state_dict().items()}
    dinov2_weights = {k: T(v) for k, v in R.fill_state_dict({"dinov2." + k: s for k, s in vit_shapes.items()}, seed).items()}
    dinov2_weights = {k[len("dinov2."):]: v for k, v in dinov2_weights.items()}
    # roma_model() calls load_state_dict(weights) unconditionally: hand it the recipe weights keyed like its own state_dict
    orig = RegressionMatcher.load_state_dict
    holder = {}

    def capture(self, sd, *a, **k):
        holder["shapes"] = {kk: tuple(v.shape) for kk, v in self.state_dict().items()}
        return None

    RegressionMatcher.load_state_dict = capture
    try:
        model = roma_model(resolution=(res, res), upsample_preds=True, weights={}, dinov2_weights=dinov2_weights,
                           device="cpu", amp_dtype=torch.float32)
    finally:
        RegressionMatcher.load_state_dict = orig
    w = R.fill_state_dict(holder["shapes"], seed, E2E_GAINS)
    model.load_state_dict({k: T(v) for k, v in w.items()})
    model.upsample_res = (up_res, up_res)
    return model.eval()


def gen_e2e(ref, full=False):
    from PIL import Image
    out = {}
    model = build_full_reference(112, 168)
    pa, pb = (os.path.join(HERE, "assets", f"sacre_coeur_{n}.jpg") for n in "AB")
    t = time.time()
    warp, cert = model.match(pa, pb, device="cpu")
    print(f"  reduced 112->168 match: {time.time()-t:.1f}s")
    out["r112_warp"], out["r112_cert"] = warp.numpy(), cert.numpy()
    # PIL-image entry point must give the same result as the path entry point
    warp2, cert2 = model.match(Image.open(pa).convert("RGB"), Image.open(pb).convert("RGB"), device="cpu")
    assert torch.equal(warp, warp2) and torch.equal(cert, cert2)
    if full:
        # BASELINE.json configs[0]: 560x560 coarse-only on the sacre_coeur pair, CPU
        model.h_resized = model.w_resized = 560
        model.upsample_preds = False
        t = time.time()
        warp, cert = model.match(pa, pb, device="cpu")
        print(f"  560 coarse-only match: {time.time()-t:.1f}s")
        out["c560_warp_stats"], out["c560_cert_stats"] = R.checksum(warp.numpy()), R.checksum(cert.numpy())
        out["c560_warp_sample"] = warp.numpy()[::8, ::8].copy()
        out["c560_cert_sample"] = cert.numpy()[::8, ::8].copy()
        save("e2e_560", **{k: v for k, v in out.items() if k.startswith("c560")})
        # BASELINE.json configs[1] on the real pair: full 560 -> 864 coarse-to-fine (about a minute on 8 cores)
        model.upsample_preds = True
        model.upsample_res = (864, 864)
        t = time.time()
        warp, cert = model.match(pa, pb, device="cpu")
        print(f"  560->864 full match: {time.time()-t:.1f}s")
        save("e2e_864", warp_stats=R.checksum(warp.numpy()), cert_stats=R.checksum(cert.numpy()),
             warp_sample=warp.numpy()[::6, ::6].copy(), cert_sample=cert.numpy()[::6, ::6].copy())
    save("e2e_112", **{k: v for k, v in out.items() if k.startswith("r112")})


def gen_e2e_second_pair(ref):
    """A second photograph pair for the headline configuration (BASELINE.json configs[1], 560 -> 864, fp32, seeded weights): the
    reference's other bundled asset pair, toronto_A/B.jpg (a wide-baseline, low-overlap pair — a different regime from sacre_coeur).
    About a minute and a half on 8 cores; run with --only e2e2."""
    model = build_full_reference(560, 864)
    pa, pb = (os.path.join(HERE, "assets", f"toronto_{n}.jpg") for n in "AB")
    t = time.time()
    warp, cert = model.match(pa, pb, device="cpu")
    print(f"  toronto 560->864 full match: {time.time()-t:.1f}s")
    save("e2e_864_toronto", warp_stats=R.checksum(warp.numpy()), cert_stats=R.checksum(cert.numpy()),
         warp_sample=warp.numpy()[::6, ::6].copy(), cert_sample=cert.numpy()[::6, ::6].copy())


# ------------------------------------------------------------------------------------------------
# 10. TinyRoMa: corr_volume, pos_embed (fast path B=1, exact path B>1), forward/match on a stand-in backbone
# ------------------------------------------------------------------------------------------------
def gen_tiny(ref):
    from romatch.models.tiny import TinyRoMa
    out = {}
    xf = load_filled(StubXFeat(), "tiny.xfeat.")
    m = TinyRoMa(xfeat=xf, freeze_xfeat=True, exact_softmax=False)
    load_filled(m, "tiny.", gains={})
    m.eval()
    f0 = R.normal("tiny.f0", (2, 64, 8, 12))
    f1 = R.normal("tiny.f1", (2, 64, 8, 12))
    with torch.no_grad():
        cv = m.corr_volume(T(f0), T(f1))
        out["corr_volume"] = cv.numpy()
        out["pos_embed_fast_b1"] = m.pos_embed(cv[:1]).numpy()
        m.exact_softmax = True
        out["pos_embed_exact_b2"] = m.pos_embed(cv).numpy()
        m.exact_softmax = False
        im0 = R.uniform("tiny.im0", (1, 3, 100, 140), 0, 1)
        im1 = R.uniform("tiny.im1", (1, 3, 100, 140), 0, 1)
        c = m.forward({"im_A": T(im0), "im_B": T(im1)})
        for s in c:
            out[f"fwd_flow_{s}"], out[f"fwd_cert_{s}"] = c[s]["flow"].numpy(), c[s]["certainty"].numpy()
        warp, cert = m.match(T(im0), T(im1), batched=True)
        out["match_warp"], out["match_cert"] = warp.numpy(), cert.numpy()
    save("tiny", **out)


GENS = {"local_corr": gen_local_corr, "cls": gen_cls, "gp": gen_gp, "refiner": gen_refiner, "decoder": gen_decoder,
        "post": gen_post, "kde": gen_kde, "preproc": gen_preproc, "e2e": gen_e2e, "tiny": gen_tiny, "e2e2": gen_e2e_second_pair}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--full", action="store_true", help="also run the 560x560 coarse-only CPU case (≈1 min)")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    ref = import_reference()
    for name, fn in GENS.items():
        if (args.only and name not in args.only) or (name == "e2e2" and not (args.only and "e2e2" in args.only)):
            continue                                               # e2e2 (the second photograph pair, ~1.5 min) only on request
        print(f"[{name}]")
        t = time.time()
        if name == "e2e":
            fn(ref, full=args.full)
        else:
            fn(ref)
        print(f"  {time.time()-t:.1f}s")


if __name__ == "__main__":
    main()
