"""The oracle (oracle/roma_oracle.py) against the golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle: SURVEY.md §8(c) items 1-10."""
import math

import numpy as np
import pytest
import torch

from oracle import roma_oracle as O
from tests import helpers as H
from tests.golden import cases, recipes as R

torch.set_grad_enabled(False)


def close(a, b, atol, rtol=0.0):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    lim = atol + rtol * np.abs(b)
    assert (err <= lim).all(), f"max err {err.max():.3e} (atol {atol}, rtol {rtol})"


# 1 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.LOCAL_CORR_CASES))
def test_local_correlation(name):
    g = H.golden("local_corr")
    f0, f1, flow, r = cases.local_corr_inputs(name)
    out = O.local_correlation(H.T(f0), H.T(f1), r, flow=None if flow is None else H.T(flow))
    # blend form vs the reference's per-tap grid_sample: differs only by fp32 rounding of the tap coordinates
    close(out, g[name], atol=2e-5)


def test_local_correlation_full_size_l16():
    g = H.golden("local_corr")
    B, C, h, w, r = 2, 512, 40, 40, 7
    out = O.local_correlation(H.T(R.normal("lc.full16.f0", (B, C, h, w))), H.T(R.normal("lc.full16.f1", (B, C, h, w))), r,
                              flow=H.T(R.coherent_flow("lc.full16.flow", B, h, w))).numpy()
    close(out[:, ::7, ::3, ::3], g["full16_sample"], atol=3e-5)
    np.testing.assert_allclose(R.checksum(out)[:3], g["full16_stats"][:3], rtol=1e-5, atol=1e-2)


# 2 ---------------------------------------------------------------------------------------------
def test_cls_to_flow_refine():
    g = H.golden("cls_to_flow_refine")
    close(O.cls_to_flow_refine(H.T(cases.cls_inputs())), g["flow"], atol=1e-6)
    close(O.cls_to_flow_refine(H.T(cases.cls_inputs(res=8, B=1, H=5, W=7))), g["flow_small"], atol=1e-6)


# 3 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dims", [("small", (1, 512, 10, 10, 512)), ("rect", (2, 64, 6, 9, 32))])
def test_gp(name, dims):
    g = H.golden("gp")
    b, c, h, w, d = dims
    gp = H.load_recipe_weights(O.GP(d), f"gp.{name}.", gains={"pos_conv": 4.0})
    x, y = H.T(R.normal(f"gp.{name}.x", (b, c, h, w))), H.T(R.normal(f"gp.{name}.y", (b, c, h, w)))
    close(gp(x, y), g[f"{name}_mu"], atol=1e-5)
    close(O.cos_kernel(x.flatten(2).transpose(1, 2), y.flatten(2).transpose(1, 2)), g[f"{name}_Kxy"], atol=1e-6)


# 4 ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.REFINER_CASES))
def test_conv_refiner(name):
    g = H.golden("conv_refiner")
    m = H.load_recipe_weights(H.build_refiner(O, name), f"ref.{name}.")
    x, y, flow, sf = cases.refiner_inputs(name)
    d, c = m(H.T(x), H.T(y), H.T(flow), scale_factor=sf)
    close(d, g[f"{name}_dflow"], atol=2e-5)
    close(c, g[f"{name}_dcert"], atol=2e-5)


# 5 ---------------------------------------------------------------------------------------------
def test_decoder_both_modes():
    g = H.golden("decoder")
    dec = H.load_recipe_weights(H.build_reduced_decoder(O), "dec.", gains=cases.DEC_GAINS)
    f1, f2 = cases.decoder_pyramids("coarse", 2, 112, upsample=False)
    c = dec({s: H.T(v) for s, v in f1.items()}, {s: H.T(v) for s, v in f2.items()})
    for s in (16, 8, 4, 2, 1):
        close(c[s]["flow"], g[f"coarse_flow_{s}"], atol=1e-5)
        close(c[s]["certainty"], g[f"coarse_cert_{s}"], atol=1e-4)
    g1, g2 = cases.decoder_pyramids("up", 2, 176, upsample=True)
    u = dec({s: H.T(v) for s, v in g1.items()}, {s: H.T(v) for s, v in g2.items()}, upsample=True,
            flow=c[1]["flow"], certainty=c[1]["certainty"], scale_factor=math.sqrt(176 * 176 / (112 * 112)))
    for s in (8, 4, 2, 1):
        close(u[s]["flow"], g[f"up_flow_{s}"], atol=1e-5)
        close(u[s]["certainty"], g[f"up_cert_{s}"], atol=1e-4)


# 6 ---------------------------------------------------------------------------------------------
def test_match_postprocessing():
    g = H.golden("match_post")
    c16, flow, cert = cases.post_inputs()
    warp, certainty = O.match_finalize(H.T(flow), H.T(cert), H.T(c16), 30, 36, symmetric=True, attenuate_cert=True)
    close(warp[0], g["warp"], atol=1e-6)
    close(certainty[0], g["certainty"], atol=1e-6)
    assert abs(float(g["scale_factor"]) - math.sqrt(30 * 36 / (20 * 24))) < 1e-12
    assert (np.asarray(g["certainty"]) == 0).any(), "fixture must exercise the out-of-range zeroing"


# 7 ---------------------------------------------------------------------------------------------
def test_kde():
    g = H.golden("kde")
    x = H.T(cases.kde_inputs())
    close(O.kde(x, half=False), g["fp32"], atol=1e-4, rtol=1e-5)
    close(O.kde(x, half=True).float(), g["fp16"], atol=0.0)          # same ops, same dtype: bit-exact
    close(O.kde(x, half=False, down=4), g["down4"], atol=1e-4, rtol=1e-5)


# 8 ---------------------------------------------------------------------------------------------
def test_preprocessing():
    from PIL import Image
    g = H.golden("preproc")
    for n in "AB":
        im = Image.open(H.asset(f"sacre_coeur_{n}.jpg")).convert("RGB")
        for res in (560, 864):
            t = O.preprocess(im, (res, res)).numpy()
            close(t[:, ::16, ::16], g[f"{n}{res}_sample"], atol=1e-6)
            np.testing.assert_allclose(R.checksum(t), g[f"{n}{res}_stats"], rtol=1e-6, atol=1e-3)


# 9 ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_oracle():
    m = O.roma_model(resolution=(112, 112), upsample_res=(168, 168))
    vit = m.encoder.dinov2_vitl14[0]
    w, v = H.full_model_weights({k: tuple(t.shape) for k, t in m.state_dict().items()},
                                {k: tuple(t.shape) for k, t in vit.state_dict().items()})
    m.load_state_dict({k: H.T(a) for k, a in w.items()})
    vit.load_state_dict({k: H.T(a) for k, a in v.items()})
    vit.eval()
    return m


def test_state_dict_layout(full_oracle):
    """603 keys in the reference layout."""
    keys = list(full_oracle.state_dict().keys())
    assert len(keys) == 603
    assert "decoder.embedding_decoder.to_out.weight" in keys and "encoder.cnn.layers.0.weight" in keys
    assert tuple(full_oracle.state_dict()["decoder.conv_refiner.16.block1.0.weight"].shape) == (1377, 1, 5, 5)


def test_end_to_end_reduced(full_oracle):
    g = H.golden("e2e_112")
    warp, cert = full_oracle.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"))
    assert warp.shape == (168, 336, 4) and cert.shape == (168, 336)
    dw = (warp - H.T(g["r112_warp"])).abs()
    dc = (cert - H.T(g["r112_cert"])).abs()
    # the argmax over 4096 classes is a discontinuity: report max-abs AND fraction; here both are tiny
    assert float((dw > 1e-3).float().mean()) < 1e-4, float(dw.max())
    assert float(dw.max()) < 1e-3 and float(dc.max()) < 1e-3, (float(dw.max()), float(dc.max()))


def test_end_to_end_560_coarse_only(full_oracle):
    """BASELINE.json configs[0]: 560x560 coarse-only on the sacre_coeur pair, CPU."""
    g = H.golden("e2e_560")
    full_oracle.h_resized = full_oracle.w_resized = 560
    full_oracle.upsample_preds = False
    try:
        warp, cert = full_oracle.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"))
    finally:
        full_oracle.h_resized = full_oracle.w_resized = 112
        full_oracle.upsample_preds = True
    dw = (warp[::8, ::8] - H.T(g["c560_warp_sample"])).abs()
    dc = (cert[::8, ::8] - H.T(g["c560_cert_sample"])).abs()
    assert float((dw > 1e-3).float().mean()) < 1e-3, float(dw.max())
    assert float(dc.max()) < 1e-3
    np.testing.assert_allclose(R.checksum(cert.numpy())[:3], g["c560_cert_stats"][:3], rtol=1e-4)


# 10 --------------------------------------------------------------------------------------------
def test_tiny():
    g = H.golden("tiny")
    xf = H.load_recipe_weights(cases.StubXFeat(), "tiny.xfeat.")
    del xf.heatmap_head, xf.keypoint_head, xf.fine_matcher
    m = O.TinyRoMa(xf.eval())
    H.load_recipe_weights(m, "tiny.")
    m.eval()
    f0, f1 = H.T(R.normal("tiny.f0", (2, 64, 8, 12))), H.T(R.normal("tiny.f1", (2, 64, 8, 12)))
    cv = O.tiny_corr_volume(f0, f1)
    close(cv, g["corr_volume"], atol=1e-5)
    close(O.tiny_pos_embed(cv[:1], exact_softmax=False), g["pos_embed_fast_b1"], atol=1e-5)
    close(O.tiny_pos_embed(cv, exact_softmax=True), g["pos_embed_exact_b2"], atol=1e-5)
    im0, im1 = H.T(R.uniform("tiny.im0", (1, 3, 100, 140), 0, 1)), H.T(R.uniform("tiny.im1", (1, 3, 100, 140), 0, 1))
    c = m(im0, im1)
    for s in (8, 4):
        close(c[s]["flow"], g[f"fwd_flow_{s}"], atol=1e-4)
        close(c[s]["certainty"], g[f"fwd_cert_{s}"], atol=1e-4)
    warp, cert = m.match(im0, im1)
    close(warp, g["match_warp"], atol=1e-4)
    close(cert, g["match_cert"], atol=1e-4)
