"""Host-side logic that needs no GPU: state-dict layout / strict loading, preprocessing vs the reference fixture,
BatchNorm folding algebra, shard arithmetic, synthetic data determinism, loud failure without a GPU."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import recipes as R

torch.set_grad_enabled(False)


def test_state_dict_matches_reference_layout():
    from roma_amd.model_zoo import build_roma
    m = build_roma((560, 560))
    sd = m.state_dict()
    assert len(sd) == 603
    assert tuple(sd["decoder.conv_refiner.16.block1.0.weight"].shape) == (1377, 1, 5, 5)
    assert tuple(sd["decoder.conv_refiner.8.out_conv.weight"].shape) == (3, 1137, 1, 1)
    assert tuple(sd["decoder.conv_refiner.4.disp_emb.weight"].shape) == (32, 2, 1, 1)
    assert tuple(sd["decoder.embedding_decoder.to_out.weight"].shape) == (4097, 1024)
    assert tuple(sd["decoder.gps.16.pos_conv.weight"].shape) == (512, 2, 1, 1)
    assert "decoder.embedding_decoder.blocks.0.attn.qkv.bias" not in sd        # decoder blocks: no qkv bias
    assert tuple(sd["encoder.cnn.layers.36.weight"].shape) == (512, 512, 3, 3)      # conv of the last captured VGG stage
    vit = m.encoder.dinov2_vitl14[0].state_dict()
    assert tuple(vit["pos_embed"].shape) == (1, 1370, 1024) and "blocks.23.ls2.gamma" in vit and len(vit) == 343
    assert not any(k.startswith("encoder.dinov2") for k in sd)                 # held outside the module tree


def test_strict_load_and_oracle_key_equality():
    from roma_amd.model_zoo import build_roma
    from oracle import roma_oracle as O
    m = build_roma((112, 112))
    o = O.roma_model((112, 112), (168, 168))
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    bad = dict(m.state_dict())
    bad.pop("decoder.proj.16.0.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)


def test_preprocess_matches_reference_fixture():
    from PIL import Image
    from roma_amd.matcher import preprocess
    g = H.golden("preproc")
    im = Image.open(H.asset("sacre_coeur_A.jpg")).convert("RGB")
    t = preprocess(im, (560, 560)).numpy()
    assert np.abs(t[:, ::16, ::16] - g["A560_sample"]).max() < 1e-6
    np.testing.assert_allclose(R.checksum(t), g["A560_stats"], rtol=1e-6, atol=1e-3)


def test_match_input_validation_without_gpu():
    from PIL import Image
    from roma_amd.model_zoo import build_roma
    m = build_roma((112, 112))
    with pytest.raises(RuntimeError, match="MI355X"):
        m.match(Image.new("RGB", (32, 32)), Image.new("RGB", (32, 32)), device="cpu")
    assert m.get_output_resolution() == (14 * 16 * 6, 14 * 16 * 6)
    m.upsample_res = (864, 1152)
    assert m.get_output_resolution() == (864, 1152)
    m.upsample_preds = False
    assert m.get_output_resolution() == (112, 112)


def test_refiner_bn_folding_algebra():
    """scale/shift produced by ConvRefiner.prepare reproduce BatchNorm(eval)(dwconv(x)+bias) exactly (fp32, CPU math)."""
    import torch.nn.functional as F
    from roma_amd.matcher import ConvRefiner
    r = ConvRefiner(24, 24, 3, hidden_blocks=1, displacement_emb_dim=6, local_corr_radius=None).eval()
    H.load_recipe_weights(r, "fold.")
    P = r.prepare(torch.float32)
    x = torch.randn(1, 24, 9, 11)
    dw, bn, _, pw = r.block1
    ref = F.relu(bn(dw(x)))
    w25, scale, shift, wt, b = P["blocks"][0]
    conv = F.conv2d(x, w25.t().reshape(24, 1, 5, 5), None, 1, 2, groups=24)
    mine = F.relu(conv * scale[None, :, None, None] + shift[None, :, None, None])
    assert float((mine - ref).abs().max()) < 1e-5
    ref2 = pw(ref)
    mine2 = (mine.permute(0, 2, 3, 1).reshape(-1, 24) @ wt + b).reshape(1, 9, 11, 24).permute(0, 3, 1, 2)
    assert float((mine2 - ref2).abs().max()) < 1e-5


def test_proj_folding_algebra():
    from roma_amd.model_zoo import build_roma
    m = build_roma((112, 112)).eval()
    H.load_recipe_weights(m.decoder.proj, "projfold.")
    wt, b = m.decoder.folded_proj(torch.float32)["4"]
    x = torch.randn(2, 256, 5, 7)
    ref = m.decoder.proj["4"](x)
    mine = (x.permute(0, 2, 3, 1).reshape(-1, 256) @ wt + b).reshape(2, 5, 7, 256).permute(0, 3, 1, 2)
    assert float((mine - ref).abs().max()) < 1e-4


def test_vgg_folding_algebra():
    import torch.nn.functional as F
    from roma_amd.encoders import VGG19
    v = VGG19().eval()
    H.load_recipe_weights(v, "vggfold.")
    plan = v.fold(torch.float32)
    x = torch.randn(1, 3, 16, 16)
    ref = v.layers[2](v.layers[1](v.layers[0](x)))
    w, b = plan[0]
    assert float((F.relu(F.conv2d(x, w, b, padding=1)) - ref).abs().max()) < 1e-5
    assert sum(1 for s in plan if s is None) == 4 and sum(1 for s in plan if s is not None) == 12   # features[:40]


def test_synthetic_inputs_are_deterministic_and_normalised():
    from roma_amd.synthetic import synthetic_pair, synthetic_state_dict
    a = synthetic_pair(3, (56, 56), (84, 84))
    b = synthetic_pair(3, (56, 56), (84, 84))
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a[0].shape == (1, 3, 56, 56) and a[2].shape == (1, 3, 84, 84)
    assert abs(float(a[2].mean())) < 0.05 and 0.8 < float(a[2].std()) < 1.2
    assert not torch.equal(synthetic_pair(4, (56, 56), (84, 84))[0], a[0])
    s1 = synthetic_state_dict({"x.weight": (4, 3, 3, 3), "x.running_var": (4,)}, 0)
    s2 = synthetic_state_dict({"x.running_var": (4,), "x.weight": (4, 3, 3, 3)}, 0)
    assert torch.equal(s1["x.weight"], s2["x.weight"]) and float(s1["x.running_var"].min()) >= 0.5


def test_shard_ranges_cover_all_pairs_in_order():
    from roma_amd.dist import shard_range
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_pil_bicubic_integer_restatement_is_bit_exact():
    """roma_amd/preproc.py: the fixed-point tables + two integer passes reproduce PIL.Image.resize(BICUBIC) bit for bit
    (down-scaling with antialias support, up-scaling, non-square, one axis untouched) — the reference's TupleResize,
    utils.py:234-239.  This is the host-side reference of the roma_resample_u8 kernel."""
    import os
    import numpy as np
    from PIL import Image
    from roma_amd.preproc import resample_numpy, resample_tables
    root = os.path.join(os.path.dirname(__file__), "golden", "assets")
    for name in ("sacre_coeur_A.jpg", "sacre_coeur_B.jpg"):
        im = Image.open(os.path.join(root, name)).convert("RGB")
        a = np.asarray(im)
        for size in [(560, 560), (864, 864), (37, 53), (a.shape[0] * 2, a.shape[1] + 7), (a.shape[0], 300)]:
            ref = np.asarray(im.resize((size[1], size[0]), Image.BICUBIC))
            assert np.array_equal(resample_numpy(a, size), ref), (name, size)
    b, k, ks = resample_tables(640, 560)
    assert b.shape == (560, 2) and k.shape == (560, ks) and ks == 7
    assert (np.abs(k.sum(axis=1) - (1 << 22)) <= ks).all()          # every row of 22-bit fixed-point weights sums to 1
