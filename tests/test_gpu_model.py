"""Model-level parity on the GPU: the product modules (roma_amd.matcher, HIP kernels + ROCm library GEMMs) against
the reference's golden vectors and against the CPU oracle, with the SAME recipe weights loaded through the
reference's state-dict key layout.  fp32 mode is the parity mode (bar: 1e-3 max-abs on warp/certainty, BASELINE.json);
fp16 mode (the reference's GPU semantics) is checked for the bulk of the pixels, because the decoder's arg-max over 4096
classes is a hard discontinuity (hard parts)."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import cases, recipes as R

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


def maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def _M():
    import roma_amd.matcher as M
    return M


@pytest.mark.parametrize("name", list(cases.REFINER_CASES))
def test_conv_refiner_golden(name):
    g = H.golden("conv_refiner")
    m = H.load_recipe_weights(H.build_refiner(_M(), name), f"ref.{name}.").to(DEV)
    x, y, flow, sf = cases.refiner_inputs(name)
    d, c = m(H.T(x, DEV), H.T(y, DEV), H.T(flow, DEV), scale_factor=sf, dtype=torch.float32)
    assert maxerr(d, H.T(g[f"{name}_dflow"])) < 5e-5
    assert maxerr(c, H.T(g[f"{name}_dcert"])) < 5e-5


def test_gp_golden():
    g = H.golden("gp")
    for name, (b, c, h, w, d) in {"small": (1, 512, 10, 10, 512), "rect": (2, 64, 6, 9, 32)}.items():
        gp = H.load_recipe_weights(_M().GP(d), f"gp.{name}.", gains={"pos_conv": 4.0}).to(DEV)
        mu = gp(H.T(R.normal(f"gp.{name}.x", (b, c, h, w)), DEV), H.T(R.normal(f"gp.{name}.y", (b, c, h, w)), DEV))
        assert maxerr(mu, H.T(g[f"{name}_mu"])) < 2e-5


def test_gp_fast_path_vs_fp64_on_real_features(full_model):
    """The fp32 MFMA CosKernel + hand-blocked Cholesky (fast modes) against the fp64 evaluation (parity mode) on the
    DINOv2 features of a real photograph: the system has condition number ~1e4, so fp32 agrees to a few 1e-3 only."""
    from PIL import Image
    from roma_amd.matcher import preprocess
    _set_dtype(full_model, torch.float32)
    ims = [Image.open(H.asset(f"sacre_coeur_{n}.jpg")).convert("RGB") for n in "AB"]
    X = torch.cat([preprocess(im, (560, 560))[None] for im in ims]).to(DEV)
    f16 = full_model.encoder(X)[16]
    x = full_model.decoder.project("16", f16, torch.float32)
    y = torch.cat((x[1:], x[:1]))
    xs = x.permute(0, 2, 3, 1).reshape(2, 1600, -1).float().contiguous()
    ys = y.permute(0, 2, 3, 1).reshape(2, 1600, -1).float().contiguous()
    gp = full_model.decoder.gps["16"]
    fast = gp.posterior_rows(xs, ys, 40, 40, fp64=False)
    exact = gp.posterior_rows(xs, ys, 40, 40, fp64=True)
    err = float((fast - exact).abs().max())
    print(f"GP fast (fp32 MFMA + blocked Cholesky) vs fp64: max|d| = {err:.2e}, |mu| max = {float(exact.abs().max()):.2f}")
    assert err < 4e-3                                            # measured 1.3e-3 (x3) with the fp64-accumulated product (2.1e-3 without); the reference's own fp32 path: 9e-4
    # the swapped-pair form the Decoder uses (one self-kernel matrix + one solve for both directions, rows read in place)
    sw = gp.posterior_rows(xs, None, 40, 40, batch_shift=1)
    assert float((sw - fast).abs().max()) < 1e-4


def test_decoder_both_modes_golden():
    g = H.golden("decoder")
    dec = H.load_recipe_weights(H.build_reduced_decoder(_M()), "dec.", gains=cases.DEC_GAINS).to(DEV)
    dec.amp_dtype = torch.float32
    assert dec.gp_precision == "fp32"                             # the product GP kernels, not the fp64 diagnostic
    f1, f2 = cases.decoder_pyramids("coarse", 2, 112, upsample=False)
    c = dec({s: H.T(v, DEV) for s, v in f1.items()}, {s: H.T(v, DEV) for s, v in f2.items()})
    for s in (16, 8, 4, 2, 1):
        assert maxerr(c[s]["flow"], H.T(g[f"coarse_flow_{s}"])) < 2e-5, s
        assert maxerr(c[s]["certainty"], H.T(g[f"coarse_cert_{s}"])) < 2e-4, s
    g1, g2 = cases.decoder_pyramids("up", 2, 176, upsample=True)
    u = dec({s: H.T(v, DEV) for s, v in g1.items()}, {s: H.T(v, DEV) for s, v in g2.items()}, upsample=True,
            flow=c[1]["flow"], certainty=c[1]["certainty"], scale_factor=math.sqrt(176 * 176 / (112 * 112)))
    for s in (8, 4, 2, 1):
        assert maxerr(u[s]["flow"], H.T(g[f"up_flow_{s}"])) < 2e-5, s
        assert maxerr(u[s]["certainty"], H.T(g[f"up_cert_{s}"])) < 2e-4, s


@pytest.fixture(scope="module")
def full_model():
    from roma_amd.model_zoo import build_roma
    m = build_roma((112, 112), upsample_preds=True, amp_dtype=torch.float32)
    vit = m.encoder.dinov2_vitl14[0]
    w, v = H.full_model_weights({k: tuple(t.shape) for k, t in m.state_dict().items()},
                                {k: tuple(t.shape) for k, t in vit.state_dict().items()})
    m.load_state_dict({k: H.T(a) for k, a in w.items()})
    vit.load_state_dict({k: H.T(a) for k, a in v.items()})
    m.upsample_res = (168, 168)
    return m.to(DEV).eval()


def _set_dtype(m, dt, gp="fp32"):
    m.encoder.amp_dtype = dt
    m.decoder.amp_dtype = dt
    m.decoder.gp_precision = gp
    m.decoder.record = None
    for r in m.decoder.conv_refiner.values():
        r.amp_dtype = dt


# Parity bounds of the fp32 mode against the reference's own outputs, per GP arithmetic:
#   "fp32" = the product kernels (fp32-MFMA CosKernel + blocked Cholesky) — what every mode, including the timed fp16 one, runs;
#   "fp64" = GP in torch fp64 (diagnostic): K_yy + sigma I has condition number ~1e4 on real features, so two DIFFERENT fp32
#            evaluations of mu differ by ~1e-3 (the reference's own fp32 CPU path is 9e-4 from an fp64 run); fp64 on this side
#            leaves only the reference's share in the comparison.
# Bounds are (measured on MI355X) x 3, capped by the 1e-3 bar of BASELINE.json where the measurement allows it.
GP_MODES = ["fp32", "fp64"]


@pytest.mark.parametrize("gp", GP_MODES)
def test_end_to_end_reduced_fp32_vs_reference_golden(full_model, gp):
    """The shipped architecture (ViT-L/14 + VGG19-BN + full decoder), sacre_coeur pair, 112->168, fp32 mode."""
    g = H.golden("e2e_112")
    _set_dtype(full_model, torch.float32, gp)
    warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    assert warp.shape == (168, 336, 4) and cert.shape == (168, 336) and warp.dtype == torch.float32
    dw = (warp.cpu() - H.T(g["r112_warp"])).abs()
    dc = (cert.cpu() - H.T(g["r112_cert"])).abs()
    frac = float((dw > 1e-3).float().mean())
    print(f"fp32 e2e 112->168 gp={gp}: warp max {float(dw.max()):.2e} cert max {float(dc.max()):.2e} frac>1e-3 {frac:.2e}")
    assert float(dw.max()) < 5e-6 and float(dc.max()) < 2e-4        # measured 3.6e-7 / 4.5e-5 (fp32 GP), 3.9e-5 (fp64 GP)


@pytest.mark.parametrize("gp", GP_MODES)
def test_end_to_end_560_coarse_only_fp32(full_model, gp):
    g = H.golden("e2e_560")
    _set_dtype(full_model, torch.float32, gp)
    full_model.h_resized = full_model.w_resized = 560
    full_model.upsample_preds = False
    try:
        warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    finally:
        full_model.h_resized = full_model.w_resized = 112
        full_model.upsample_preds = True
    dw = (warp.cpu()[::8, ::8] - H.T(g["c560_warp_sample"])).abs()
    dc = (cert.cpu()[::8, ::8] - H.T(g["c560_cert_sample"])).abs()
    frac = float((dw > 1e-3).float().mean())
    print(f"560 coarse fp32 gp={gp}: warp max {float(dw.max()):.2e} cert max {float(dc.max()):.2e} frac>1e-3 {frac:.2e}")
    assert float(dw.max()) < E2E_BOUNDS[("c560", gp)][0] and float(dc.max()) < E2E_BOUNDS[("c560", gp)][1]


@pytest.mark.parametrize("gp", GP_MODES)
def test_end_to_end_full_560_to_864_fp32(full_model, gp, miopen_find):
    """BASELINE.json configs[1] on the real pair: full coarse-to-fine 560 -> 864, fp32 mode, against the reference's
    own output (tests/golden/e2e_864.npz, every 6th pixel + checksums).  Bar: 1e-3 max-abs (north_star)."""
    g = H.golden("e2e_864")
    _set_dtype(full_model, torch.float32, gp)
    full_model.h_resized = full_model.w_resized = 560
    full_model.upsample_res = (864, 864)
    try:
        warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    finally:
        full_model.h_resized = full_model.w_resized = 112
        full_model.upsample_res = (168, 168)
    assert warp.shape == (864, 1728, 4) and cert.shape == (864, 1728)
    dw = (warp.cpu()[::6, ::6] - H.T(g["warp_sample"])).abs()
    dc = (cert.cpu()[::6, ::6] - H.T(g["cert_sample"])).abs()
    frac_w, frac_c = float((dw > 1e-3).float().mean()), float((dc > 1e-3).float().mean())
    print(f"560->864 fp32 gp={gp}: warp max {float(dw.max()):.2e} frac>1e-3 {frac_w:.2e}; cert max {float(dc.max()):.2e} frac>1e-3 {frac_c:.2e}")
    assert float(dw.max()) < E2E_BOUNDS[("f864", gp)][0] and float(dc.max()) < E2E_BOUNDS[("f864", gp)][1]
    assert float(dw.median()) < 1e-6 and float(dc.median()) < 1e-4


@pytest.mark.parametrize("gp", GP_MODES)
def test_end_to_end_full_560_to_864_fp32_second_pair(full_model, gp, miopen_find):
    """The same configuration on the reference's OTHER bundled photograph pair (toronto_A/B.jpg: wide baseline, little overlap — a
    different regime from sacre_coeur, and a different draw of the GP's ill-conditioned directions), against the reference's own
    output (tests/golden/e2e_864_toronto.npz, make_golden.py --only e2e2).  Same 1e-3 max-abs bar."""
    g = H.golden("e2e_864_toronto")
    _set_dtype(full_model, torch.float32, gp)
    full_model.h_resized = full_model.w_resized = 560
    full_model.upsample_res = (864, 864)
    try:
        warp, cert = full_model.match(H.asset("toronto_A.jpg"), H.asset("toronto_B.jpg"), device=DEV)
    finally:
        full_model.h_resized = full_model.w_resized = 112
        full_model.upsample_res = (168, 168)
    dw = (warp.cpu()[::6, ::6] - H.T(g["warp_sample"])).abs()
    dc = (cert.cpu()[::6, ::6] - H.T(g["cert_sample"])).abs()
    print(f"toronto 560->864 fp32 gp={gp}: warp max {float(dw.max()):.2e} frac>1e-3 {float((dw > 1e-3).float().mean()):.2e}; "
          f"cert max {float(dc.max()):.2e} frac>1e-3 {float((dc > 1e-3).float().mean()):.2e}")
    assert float(dw.max()) < E2E_BOUNDS[("f864", gp)][0] and float(dc.max()) < E2E_BOUNDS[("f864", gp)][1]
    assert float(dw.median()) < 1e-6 and float(dc.median()) < 1e-4
    np.testing.assert_allclose(R.checksum(warp.cpu().numpy())[:3], g["warp_stats"][:3], rtol=1e-4, atol=1e-2)


# (warp max-abs, certainty max-abs) bounds; measured values in the comments (MI355X, round 2)
E2E_BOUNDS = {
    ("c560", "fp64"): (5e-6, 1e-3),      # measured 2.4e-7 / 5.2e-4
    ("c560", "fp32"): (5e-6, 1e-3),      # measured 2.4e-7 / 4.4e-4  (product GP kernels)
    ("f864", "fp64"): (5e-6, 1e-3),      # measured 3.6e-7 / 4.6e-4
    ("f864", "fp32"): (5e-6, 1e-3),      # measured 3.6e-7 / 5.2e-4  (product GP kernels): inside the 1e-3 bar of north_star
}


def _match_560_864(model, dtype, gp="fp32", record=False):
    _set_dtype(model, dtype, gp)
    model.h_resized = model.w_resized = 560
    model.upsample_res = (864, 864)
    if record:
        model.decoder.record = {}
    try:
        out = model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
        rec = model.decoder.record
    finally:
        model.h_resized = model.w_resized = 112
        model.upsample_res = (168, 168)
        _set_dtype(model, torch.float32)
    return out, rec


def test_fp16_mode_560_to_864_vs_reference_and_argmax_flips(full_model, miopen_find):
    """The mode bench.py TIMES (fp16 = the reference's GPU autocast semantics) at the headline configuration, against the
    reference's own (fp32 CPU) output, plus the count of scale-16 arg-max flips against the fp32 mode (the 4096-way arg-max of
    cls_to_flow_refine, utils.py:316, is the path's one discontinuity; SURVEY §7 measured that rounding the features to fp16
    moves single pixels by O(1) while p99 stays ~2.6e-4)."""
    g = H.golden("e2e_864")
    (w32, c32), r32 = _match_560_864(full_model, torch.float32, record=True)
    (w16, c16), r16 = _match_560_864(full_model, torch.float16, record=True)
    assert torch.isfinite(w16).all() and torch.isfinite(c16).all()
    flips = int((r32["argmax16"] != r16["argmax16"]).sum())
    dmu = float((r32["mu16"] - r16["mu16"]).abs().max())
    dw = (w16.cpu()[::6, ::6] - H.T(g["warp_sample"])).abs()
    dc = (c16.cpu()[::6, ::6] - H.T(g["cert_sample"])).abs()
    q = lambda t, f: float(t.flatten().kthvalue(max(1, int(t.numel() * f))).values)  # noqa: E731
    print(f"fp16 560->864 vs reference: warp median {float(dw.median()):.2e} p99 {q(dw, 0.99):.2e} max {float(dw.max()):.2e} "
          f"frac>1e-3 {float((dw > 1e-3).float().mean()):.2e}; cert median {float(dc.median()):.2e} p99 {q(dc, 0.99):.2e} max {float(dc.max()):.2e}; "
          f"scale-16 arg-max flips vs fp32 mode {flips}/3200; GP mu max|d| {dmu:.2e}")
    assert flips <= FP16_BOUNDS["flips"]
    assert float(dw.median()) < FP16_BOUNDS["warp_median"] and q(dw, 0.99) < FP16_BOUNDS["warp_p99"]
    assert float(dc.median()) < FP16_BOUNDS["cert_median"] and q(dc, 0.99) < FP16_BOUNDS["cert_p99"]


# measured on MI355X: 17 of 3200 scale-16 arg-max flips vs the fp32 mode; vs the reference's fp32 output: warp median 0, p99 4.3e-2,
# 2.3 % of entries beyond 1e-3 (the flipped neighbourhoods); certainty median 1.8e-4, p99 2.3e-2.  Bounds = measured x 3.
FP16_BOUNDS = {"flips": 60, "warp_median": 1e-5, "warp_p99": 0.13, "cert_median": 6e-4, "cert_p99": 7e-2}


def test_non_square_resolution_vs_oracle(full_model):
    """Non-square coarse / upsample resolutions (the reference's demos use upsample_res=(864,1152)): the product against
    the CPU oracle with the same weights, fp32 mode, synthetic pair."""
    from oracle import roma_oracle as O
    from roma_amd.synthetic import synthetic_pair
    _set_dtype(full_model, torch.float32)
    lo, hi = (112, 168), (160, 240)
    o = O.roma_model(lo, hi)
    o.load_state_dict({k: v.cpu() for k, v in full_model.state_dict().items()})
    o.encoder.dinov2_vitl14[0].load_state_dict({k: v.float().cpu() for k, v in full_model.encoder.dinov2_vitl14[0].state_dict().items()})
    o.encoder.dinov2_vitl14[0].eval()
    pair = synthetic_pair(7, lo, hi)
    rw, rc = o.match_tensors(*pair)
    full_model.h_resized, full_model.w_resized = lo
    full_model.upsample_res = hi
    try:
        w, c = full_model.match_tensors(*(t.to(DEV) for t in pair))
    finally:
        full_model.h_resized = full_model.w_resized = 112
        full_model.upsample_res = (168, 168)
    assert w.shape == (1, 160, 480, 4) and c.shape == (1, 160, 480)
    dw, dc = (w.cpu() - rw).abs(), (c.cpu() - rc).abs()
    print(f"non-square fp32: warp max {float(dw.max()):.2e} frac>1e-3 {float((dw > 1e-3).float().mean()):.2e}; cert max {float(dc.max()):.2e}")
    assert float((dw > 1e-3).float().mean()) < 5e-3 and float(dw.median()) < 1e-5 and float(dc.median()) < 1e-4


def test_bf16_mode_runs_and_agrees_in_bulk(full_model):
    g = H.golden("e2e_112")
    _set_dtype(full_model, torch.bfloat16)
    try:
        warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    finally:
        _set_dtype(full_model, torch.float32)
    assert torch.isfinite(warp).all() and torch.isfinite(cert).all()
    assert float((cert.cpu() - H.T(g["r112_cert"])).abs().median()) < 2e-2


def test_end_to_end_fp16_mode_bulk_agreement(full_model):
    g = H.golden("e2e_112")
    _set_dtype(full_model, torch.float16)
    try:
        warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    finally:
        _set_dtype(full_model, torch.float32)
    assert torch.isfinite(warp).all() and torch.isfinite(cert).all()
    dw = (warp.cpu() - H.T(g["r112_warp"])).abs()
    dc = (cert.cpu() - H.T(g["r112_cert"])).abs()
    p99 = float(dw.flatten().kthvalue(int(dw.numel() * 0.99)).values)
    print(f"fp16 e2e: warp median {float(dw.median()):.2e} p99 {p99:.2e} max {float(dw.max()):.2e}; cert max {float(dc.max()):.2e}")
    # measured: median 0, p99 8.5e-6, max 1.2e-5, certainty max 5.5e-3 (x3); max-abs is left out because one flipped coarse
    # arg-max (64 coarse pixels here) would move a neighbourhood
    assert float(dw.median()) < 1e-5 and p99 < 3e-5 and float(dc.max()) < 2e-2


def test_fp16_mode_with_the_hand_written_attention_kernel(full_model):
    """The opt-in DINOv2 attention kernel (ROMA_ATTENTION=hip -> roma_attention_fwd) inside the whole fp16 pipeline, 112 -> 168: the
    DINOv2 features agree with the SDPA path at the level of the 16-bit noise, and the match agrees in bulk (a flipped coarse
    arg-max moves 1/64 of this small map: bound on the bulk, as in test_end_to_end_fp16_mode_bulk_agreement)."""
    import roma_amd.transformer as TR
    _set_dtype(full_model, torch.float16)
    try:
        x = torch.randn(2, 3, 112, 112, generator=torch.Generator().manual_seed(11)).to(DEV)
        TR.ATTENTION_KERNEL = False
        f_sdpa = full_model.encoder.vit_features(x).float()
        w_sdpa, c_sdpa = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
        TR.ATTENTION_KERNEL = True
        f_hip = full_model.encoder.vit_features(x).float()
        w_hip, c_hip = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    finally:
        TR.ATTENTION_KERNEL = "sdpa"
        _set_dtype(full_model, torch.float32)
    rel = float((f_hip - f_sdpa).pow(2).mean().sqrt() / f_sdpa.pow(2).mean().sqrt())
    dw = (w_hip - w_sdpa).abs()
    moved = float((dw > 1e-3).float().mean())
    print(f"hip vs sdpa attention: feature rms difference {rel:.2e} of the feature rms; warp entries moved > 1e-3: {moved:.3f}; "
          f"certainty max diff {float((c_hip - c_sdpa).abs().max()):.2e}")
    assert torch.isfinite(w_hip).all() and torch.isfinite(c_hip).all()
    assert rel < 5e-3                                              # measured 1.4e-3 (the fp16 noise itself is 1.8e-3 of the fp32 features)
    assert float(dw.median()) < 1e-5 and moved <= 6 / 64 and float((c_hip - c_sdpa).abs().max()) < 3e-2


def test_batched_pairs_equal_per_pair_results(full_model):
    """Stack-of-per-pair semantics: P=2 pairs in one call == each pair alone (bit for bit in fp32)."""
    _set_dtype(full_model, torch.float32)
    gen = torch.Generator().manual_seed(5)
    lo = torch.randn(2, 2, 3, 112, 112, generator=gen).to(DEV)
    hi = torch.randn(2, 2, 3, 168, 168, generator=gen).to(DEV)
    w2, c2 = full_model.match_tensors(lo[0], lo[1], hi[0], hi[1])
    for i in range(2):
        w1, c1 = full_model.match_tensors(lo[0, i:i + 1], lo[1, i:i + 1], hi[0, i:i + 1], hi[1, i:i + 1])
        assert maxerr(w2[i], w1[0]) < 1e-4 and maxerr(c2[i], c1[0]) < 1e-4


def test_sample_and_coordinates(full_model):
    _set_dtype(full_model, torch.float32)
    warp, cert = full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device=DEV)
    torch.manual_seed(0)
    m, c = full_model.sample(warp, cert, num=500)
    assert m.shape == (500, 4) and c.shape == (500,)
    assert float(m.abs().max()) <= 1.0
    kA, kB = full_model.to_pixel_coordinates(m, 480, 640, 640, 618)
    assert kA.shape == (500, 2) and float(kA[:, 0].max()) <= 640 and float(kB[:, 1].max()) <= 640
    nA, nB = full_model.to_normalized_coordinates((kA, kB), 480, 640, 640, 618)
    assert maxerr(nA, m[:, :2]) < 1e-5 and maxerr(nB, m[:, 2:]) < 1e-5
    assert full_model.get_output_resolution() == (168, 168)


def test_match_keypoints_equals_the_reference_formula(full_model):
    """match_keypoints (matcher.py:576-591) with two arg-min kernels instead of the |x_A| x |x_B| cdist matrix: same mutual
    nearest neighbours as the literal formula (oracle.match_keypoints) on random keypoints and a smooth synthetic warp."""
    from oracle import roma_oracle as O
    g = torch.Generator().manual_seed(3)
    Hh, Ww = 96, 128
    ys, xs = torch.linspace(-1 + 1 / Hh, 1 - 1 / Hh, Hh), torch.linspace(-1 + 1 / Ww, 1 - 1 / Ww, Ww)
    grid = torch.stack((xs[None].expand(Hh, Ww), ys[:, None].expand(Hh, Ww)), dim=-1)
    tgt = 0.9 * grid.flip(-1) * torch.tensor([1.0, -1.0]) + 0.05                                  # a rotation-like smooth map
    warp = torch.cat((torch.cat((grid, tgt), dim=-1), torch.cat((tgt, grid), dim=-1)), dim=1)      # (H, 2W, 4) symmetric layout
    cert = torch.rand(Hh, 2 * Ww, generator=g)
    x_A = torch.rand(3000, 2, generator=g) * 1.9 - 0.95
    x_B = torch.rand(2500, 2, generator=g) * 1.9 - 0.95
    rA, rB = O.match_keypoints(x_A, x_B, warp, cert)
    iA, iB = full_model.match_keypoints(x_A.to(DEV), x_B.to(DEV), warp.to(DEV), cert.to(DEV), return_inds=True)
    assert iA.numel() > 50
    got, ref = set(zip(iA.tolist(), iB.tolist())), set(zip(rA.tolist(), rB.tolist()))
    print(f"match_keypoints: {len(got)} mutual matches, {len(got ^ ref)} differ from the cdist formulation")
    assert len(got ^ ref) <= 2                       # cdist takes its matmul route: near-ties may round the other way
    kA, kB = full_model.match_keypoints(x_A.to(DEV), x_B.to(DEV), warp.to(DEV), cert.to(DEV))
    assert torch.equal(kA, x_A.to(DEV)[iA]) and torch.equal(kB, x_B.to(DEV)[iB])


def test_encode_once_match_many(full_model):
    """pyramids cached per image (model.encode) and reused across pairs (model.match_encoded) give the
    result of match_tensors on the same pair (the encoders see batch 1 instead of 2: library kernels may pick other tilings,
    so agreement is to fp32 rounding, not bitwise)."""
    from PIL import Image
    from roma_amd.matcher import preprocess
    _set_dtype(full_model, torch.float32)
    ims = [Image.open(H.asset(f"sacre_coeur_{n}.jpg")).convert("RGB") for n in "AB"]
    lo = [preprocess(im, (112, 112))[None].to(DEV) for im in ims]
    hi = [preprocess(im, (168, 168))[None].to(DEV) for im in ims]
    w_ref, c_ref = full_model.match_tensors(lo[0], lo[1], hi[0], hi[1])
    enc = [full_model.encode(lo[i], hi[i]) for i in range(2)]
    w, c = full_model.match_encoded(enc[0], enc[1])
    w_ba, c_ba = full_model.match_encoded(enc[1], enc[0])                 # the cached pyramids serve the reversed pair too
    print(f"encode/match_encoded vs match_tensors: warp max {maxerr(w, w_ref):.2e}, cert max {maxerr(c, c_ref):.2e}")
    assert maxerr(w, w_ref) < 1e-4 and maxerr(c, c_ref) < 1e-4
    W2 = w.shape[2] // 2
    assert maxerr(w_ba[:, :, :W2, 2:], w[:, :, W2:, :2]) < 1e-4           # B->A half of (A,B) == A->B half of (B,A)


def test_graphed_step_equals_eager(full_model):
    """RegressionMatcher.make_graphed: one captured hipGraph (both streams of the step, library and hand-written kernels) replays
    to what the eager launches produce, also after the static inputs were overwritten with another pair.  Not bitwise: the
    library's stream-K GEMMs reduce with atomics, so two EAGER runs differ in the last bits too (printed beside it)."""
    from roma_amd.synthetic import synthetic_pair
    _set_dtype(full_model, torch.float16)
    try:
        p0 = [t.to(DEV) for t in synthetic_pair(0, (112, 112), (168, 168))]
        p1 = [t.to(DEV) for t in synthetic_pair(1, (112, 112), (168, 168))]
        run = full_model.make_graphed(*p0)
        for pair in (p0, p1, p0):
            w, c = run(*pair)
            w, c = w.clone(), c.clone()
            we, ce = full_model.match_tensors(*pair)
            we2, ce2 = full_model.match_tensors(*pair)
            torch.cuda.synchronize()
            dw, dc = (w - we).abs(), (c - ce).abs()
            p99 = float(dw.flatten().kthvalue(int(dw.numel() * 0.99)).values)
            print(f"graph vs eager: warp max {float(dw.max()):.2e} p99 {p99:.2e}, cert max {float(dc.max()):.2e}; "
                  f"eager vs eager: warp max {maxerr(we, we2):.2e}, cert max {maxerr(ce, ce2):.2e}")
            assert float(dw.median()) < 1e-6 and p99 < 1e-4 and float(dc.median()) < 1e-6
    finally:
        _set_dtype(full_model, torch.float32)


def test_match_rejects_bad_images(full_model):
    from PIL import Image
    with pytest.raises(NotImplementedError):
        full_model.match(Image.new("L", (32, 32)), Image.new("L", (32, 32)), device=DEV)
    with pytest.raises(RuntimeError):
        full_model.match(H.asset("sacre_coeur_A.jpg"), H.asset("sacre_coeur_B.jpg"), device="cpu")


def test_tiny_roma_forward_and_match_golden():
    from roma_amd.tiny import TinyRoMa
    g = H.golden("tiny")
    xf = H.load_recipe_weights(cases.StubXFeat(), "tiny.xfeat.")
    m = TinyRoMa(xf.eval(), freeze_xfeat=True, exact_softmax=False)
    H.load_recipe_weights(m, "tiny.")
    m = m.to(DEV).eval()
    im0, im1 = H.T(R.uniform("tiny.im0", (1, 3, 100, 140), 0, 1), DEV), H.T(R.uniform("tiny.im1", (1, 3, 100, 140), 0, 1), DEV)
    c = m({"im_A": im0, "im_B": im1})
    for s in (8, 4):
        assert maxerr(c[s]["flow"], H.T(g[f"fwd_flow_{s}"])) < 2e-4
        assert maxerr(c[s]["certainty"], H.T(g[f"fwd_cert_{s}"])) < 2e-4
    warp, cert = m.match(im0, im1)
    assert maxerr(warp, H.T(g["match_warp"])) < 2e-4 and maxerr(cert, H.T(g["match_cert"])) < 2e-4


def test_device_preprocessing_is_bit_identical_to_pil():
    """PIL bicubic resize + /255 + normalise (utils.py:165-261) done on the device from one uint8 upload
    must equal the host path bit for bit — so match() from files is unchanged by it."""
    import os
    from PIL import Image
    from roma_amd.matcher import preprocess
    from roma_amd.preproc import preprocess_device, resize_device
    root = os.path.join(os.path.dirname(__file__), "golden", "assets")
    for name in ("sacre_coeur_A.jpg", "sacre_coeur_B.jpg"):
        im = Image.open(os.path.join(root, name)).convert("RGB")
        for size in [(560, 560), (864, 864), (37, 53), (1000, 700)]:
            dev = preprocess_device(im, size, DEV)
            host = preprocess(im, size)
            assert dev.shape == host.shape and torch.equal(dev.cpu(), host), (name, size)
        u8 = resize_device(torch.from_numpy(np.array(im, dtype=np.uint8)).to(DEV), (123, 77)).cpu().numpy()
        assert np.array_equal(u8, np.asarray(im.resize((77, 123), Image.BICUBIC)))
