"""HIP kernels (through the C ABI / ctypes binding) against the CPU oracle and the reference's golden vectors.
Run with -m gpu on an MI355X.  Tolerances: fp32 storage 2e-5 abs on O(1) values (different summation order only);
fp16/bf16 storage is compared with the oracle fed the SAME rounded inputs, tolerance = output rounding."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import cases, recipes as R

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


def _ops():
    from roma_amd import ops
    return ops


def _O():
    from oracle import roma_oracle as O
    return O


def maxerr(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def solve64(K, F):
    """fp64 reference solve on ONE host thread: multithreaded oneMKL getrf hangs on the GPU boxes' host CPUs (see
    oracle._inv_single_thread)."""
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return torch.linalg.solve(K.double(), F.double())
    finally:
        torch.set_num_threads(nt)


def test_library_is_loaded_and_on_gpu():
    from roma_amd import _lib
    assert _lib.load().roma_abi_version() == 5
    assert torch.cuda.is_available()
    with pytest.raises(RuntimeError):
        _ops().local_correlation(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4), 2)     # CPU tensors: loud failure


# ---- local_correlation -------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.LOCAL_CORR_CASES))
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_local_corr_golden_fp32(name, layout):
    g = H.golden("local_corr")
    f0, f1, flow, r = cases.local_corr_inputs(name)
    a, b = H.T(f0, DEV), H.T(f1, DEV)
    if layout == "nhwc":
        a, b = a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)
    out = _ops().local_correlation(a, b, r, flow=None if flow is None else H.T(flow, DEV))
    assert out.shape == g[name].shape
    assert maxerr(out, H.T(g[name])) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["r7_coh", "r3_adv", "r2_coh"])
def test_local_corr_low_precision_storage(name, dtype):
    f0, f1, flow, r = cases.local_corr_inputs(name)
    a, b = H.T(f0).to(dtype), H.T(f1).to(dtype)
    ref = _O().local_correlation(a.float(), b.float(), r, flow=H.T(flow))
    out = _ops().local_correlation(a.to(DEV).contiguous(memory_format=torch.channels_last),
                                   b.to(DEV).contiguous(memory_format=torch.channels_last), r, flow=H.T(flow, DEV))
    assert out.dtype == dtype
    tol = (2 ** -10 if dtype == torch.float16 else 2 ** -7) * max(1.0, float(ref.abs().max()))
    assert maxerr(out, ref) <= tol


@pytest.mark.parametrize("shape", [(2, 512, 40, 40, 7), (2, 512, 70, 70, 3), (2, 256, 140, 140, 2), (1, 256, 37, 53, 2), (1, 64, 9, 200, 5)])
def test_local_corr_full_sizes_vs_oracle(shape):
    B, C, h, w, r = shape
    f0 = R.normal(f"lcfull.{shape}.f0", (B, C, h, w))
    f1 = R.normal(f"lcfull.{shape}.f1", (B, C, h, w))
    flow = R.coherent_flow(f"lcfull.{shape}.flow", B, h, w)
    ref = _O().local_correlation(H.T(f0), H.T(f1), r, flow=H.T(flow))
    out = _ops().local_correlation(H.T(f0, DEV).contiguous(memory_format=torch.channels_last),
                                   H.T(f1, DEV).contiguous(memory_format=torch.channels_last), r, flow=H.T(flow, DEV))
    assert maxerr(out, ref) < 5e-5


_ORACLE_LC = {}


def _oracle_lc_rounded(shape, dtype):
    """oracle.local_correlation on inputs rounded to `dtype` (what the 16-bit kernels are fed), computed once per (shape, dtype)."""
    key = (shape, dtype)
    if key not in _ORACLE_LC:
        B, C, h, w, r = shape
        f0 = H.T(R.normal(f"lc16.{shape}.f0", (B, C, h, w))).to(dtype)
        f1 = H.T(R.normal(f"lc16.{shape}.f1", (B, C, h, w))).to(dtype)
        flow = H.T(R.coherent_flow(f"lc16.{shape}.flow", B, h, w))
        _ORACLE_LC[key] = (f0, f1, flow, _O().local_correlation(f0.float(), f1.float(), r, flow=flow))
    return _ORACLE_LC[key]


@pytest.mark.parametrize("variant", ["auto", "tile8x4", "tile8x8", "rows8"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 512, 40, 40, 7), (2, 512, 70, 70, 3), (2, 256, 140, 140, 2), (2, 512, 108, 108, 3), (2, 256, 216, 216, 2)])
def test_local_corr_16bit_kernels_vs_oracle_at_the_five_call_shapes(shape, dtype, variant):
    """What bench.py times (16-bit channels-last, AUTO) and every other kernel variant, DIRECTLY against the CPU oracle on the same
    rounded inputs, at all five 560->864 call shapes of BASELINE.md §3 (L16, L8, L4, U8, U4; B = 2 = one symmetric pair).
    Tolerance = the output's own rounding."""
    B, C, h, w, r = shape
    if r > 3 and variant != "auto":
        pytest.skip("r = 7 has one 16-bit kernel")
    f0, f1, flow, ref = _oracle_lc_rounded(shape, dtype)
    out = _ops().local_correlation(f0.to(DEV).contiguous(memory_format=torch.channels_last),
                                   f1.to(DEV).contiguous(memory_format=torch.channels_last), r, flow=flow.to(DEV), variant=variant)
    assert out.dtype == dtype
    tol = (2 ** -10 if dtype == torch.float16 else 2 ** -7) * max(1.0, float(ref.abs().max()))
    assert maxerr(out, ref) <= tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 512, 40, 40, 7), (1, 512, 70, 70, 3), (1, 256, 140, 140, 2), (1, 64, 37, 53, 2), (2, 32, 21, 18, 3),
                                   (1, 96, 30, 44, 7), (1, 256, 216, 216, 2)])
@pytest.mark.parametrize("kind", ["coherent", "adversarial", "mixed"])
def test_local_corr_matrix_core_path_vs_fp32_kernel(shape, kind, dtype):
    """16-bit channels-last inputs take the MFMA / per-pixel-patch kernel; the fp32 kernel (checked against the oracle
    above) on the SAME rounded inputs is the reference.  'mixed' = coherent flow with a block of adversarial targets and
    NaN / far-out-of-range entries, so one launch exercises every per-tile mode and the image borders."""
    B, C, h, w, r = shape
    ops = _ops()
    f0 = H.T(R.normal(f"lcm.{shape}.f0", (B, C, h, w))).to(dtype)
    f1 = H.T(R.normal(f"lcm.{shape}.f1", (B, C, h, w))).to(dtype)
    if kind == "coherent":
        flow = R.coherent_flow(f"lcm.{shape}.flow", B, h, w)
    elif kind == "adversarial":
        flow = R.adversarial_flow(f"lcm.{shape}.flow", B, h, w)
    else:
        flow = R.coherent_flow(f"lcm.{shape}.flow", B, h, w)
        adv = R.adversarial_flow(f"lcm.{shape}.adv", B, h, w)
        flow[:, :, h // 3: 2 * h // 3, w // 4: w // 2] = adv[:, :, h // 3: 2 * h // 3, w // 4: w // 2]
        flow[:, 0, 0, 0] = float("nan")
        flow[:, 1, -1, -1] = 1e30
        flow[:, :, 1, 2] = -7.0
    flow = H.T(flow, DEV)
    a16 = f0.to(DEV).contiguous(memory_format=torch.channels_last)
    b16 = f1.to(DEV).contiguous(memory_format=torch.channels_last)
    out = ops.local_correlation(a16, b16, r, flow=flow)
    ref = ops.local_correlation(a16.float().contiguous(memory_format=torch.channels_last),
                                b16.float().contiguous(memory_format=torch.channels_last), r, flow=flow)
    assert out.dtype == dtype and torch.isfinite(out.float()).all()
    tol = (2 ** -10 if dtype == torch.float16 else 2 ** -7) * max(1.0, float(ref.abs().max()))
    assert maxerr(out, ref) <= tol


@pytest.mark.parametrize("variant", ["tile8x4", "tile8x8", "rows8"])
@pytest.mark.parametrize("shape", [(1, 512, 70, 70, 3), (1, 256, 140, 140, 2), (2, 256, 37, 53, 1), (1, 64, 37, 53, 2), (2, 32, 21, 18, 3),
                                   (2, 256, 19, 45, 3), (1, 512, 21, 9, 2)])
@pytest.mark.parametrize("kind", ["coherent", "adversarial", "mixed"])
def test_local_corr_kernel_variants_agree(shape, kind, variant):
    """The kernels for 16-bit channels-last inputs with r <= 3 (roma_hip.h: ROMA_LC_*) against the fp32 kernel on the same
    rounded inputs, partial border tiles, NaN / far-out-of-range flow entries included.  (A variant that does not apply to a shape
    — the row-streaming kernel needs C = 256 or 512 — falls back inside the library; the call must still be correct.)"""
    B, C, h, w, r = shape
    ops = _ops()
    f0 = H.T(R.normal(f"lcv.{shape}.f0", (B, C, h, w))).half()
    f1 = H.T(R.normal(f"lcv.{shape}.f1", (B, C, h, w))).half()
    flow = R.coherent_flow(f"lcv.{shape}.flow", B, h, w)
    if kind == "adversarial":
        flow = R.adversarial_flow(f"lcv.{shape}.flow", B, h, w)
    elif kind == "mixed":
        adv = R.adversarial_flow(f"lcv.{shape}.adv", B, h, w)
        flow[:, :, h // 3: 2 * h // 3, w // 4: w // 2] = adv[:, :, h // 3: 2 * h // 3, w // 4: w // 2]
        flow[:, 0, 0, 0] = float("nan")
        flow[:, 1, -1, -1] = 1e30
        flow[:, :, 1, 2] = -7.0
    flow = H.T(flow, DEV)
    a16 = f0.to(DEV).contiguous(memory_format=torch.channels_last)
    b16 = f1.to(DEV).contiguous(memory_format=torch.channels_last)
    out = ops.local_correlation(a16, b16, r, flow=flow, variant=variant)
    ref = ops.local_correlation(a16.float().contiguous(memory_format=torch.channels_last),
                                b16.float().contiguous(memory_format=torch.channels_last), r, flow=flow)
    assert torch.isfinite(out.float()).all()
    assert maxerr(out, ref) <= 2 ** -10 * max(1.0, float(ref.abs().max()))
    planar = ops.local_correlation(a16, b16, r, flow=flow, variant=variant, out=torch.empty(out.shape, dtype=out.dtype, device=DEV))
    assert maxerr(planar, ref) <= 2 ** -10 * max(1.0, float(ref.abs().max()))          # NCHW output path


def test_local_corr_full_size_golden_l16():
    g = H.golden("local_corr")
    B, C, h, w, r = 2, 512, 40, 40, 7
    out = _ops().local_correlation(H.T(R.normal("lc.full16.f0", (B, C, h, w)), DEV), H.T(R.normal("lc.full16.f1", (B, C, h, w)), DEV), r,
                                   flow=H.T(R.coherent_flow("lc.full16.flow", B, h, w), DEV)).cpu().numpy()
    assert np.abs(out[:, ::7, ::3, ::3] - g["full16_sample"]).max() < 5e-5
    np.testing.assert_allclose(R.checksum(out)[:3], g["full16_stats"][:3], rtol=1e-5, atol=1e-2)


def test_local_corr_properties_at_bench_size():
    """Size-independent properties at the largest 560->864 call (U4: 256ch, 216x216, r=2), fp16 channels-last:
    linearity in f0, identity flow centre tap == <f0,f1>/sqrt(C), zero outside the image."""
    ops = _ops()
    B, C, h, w, r = 2, 256, 216, 216, 2
    gen = torch.Generator(device="cpu").manual_seed(0)
    f0 = torch.randn(B, C, h, w, generator=gen).half().to(DEV).contiguous(memory_format=torch.channels_last)
    f1 = torch.randn(B, C, h, w, generator=gen).half().to(DEV).contiguous(memory_format=torch.channels_last)
    flow = H.T(R.coherent_flow("lcprop.flow", B, h, w), DEV)
    a = ops.local_correlation(f0, f1, r, flow=flow).float()
    b2 = ops.local_correlation(f0 * 2, f1, r, flow=flow).float()
    assert maxerr(b2, 2 * a) <= 2e-3 * float(a.abs().max())
    ident = ops.local_correlation(f0, f1, r, flow=None).float()
    centre = (f0.float() * f1.float()).sum(1) / math.sqrt(C)
    assert maxerr(ident[:, (2 * r + 1) ** 2 // 2], centre) <= 2e-3 * float(centre.abs().max())
    far = torch.full((B, 2, h, w), 3.0, device=DEV)
    assert float(ops.local_correlation(f0, f1, r, flow=far).abs().max()) == 0.0


def test_local_corr_writes_into_concat_slice():
    ops = _ops()
    B, C, h, w, r = 1, 32, 10, 12, 2
    K = 25
    f0, f1 = torch.randn(B, C, h, w, device=DEV), torch.randn(B, C, h, w, device=DEV)
    flow = H.T(R.coherent_flow("lcslice.flow", B, h, w), DEV)
    buf = torch.zeros(B, h, w, 2 * C + K + 7, device=DEV)
    d = buf.permute(0, 3, 1, 2)
    d[:, :C] = f0
    d[:, C:2 * C] = f1
    ops.local_correlation(d[:, :C], d[:, C:2 * C], r, flow=flow, out=d[:, 2 * C:2 * C + K])
    ref = ops.local_correlation(f0, f1, r, flow=flow)
    assert maxerr(d[:, 2 * C:2 * C + K], ref) < 1e-6
    assert float(buf[..., 2 * C + K:].abs().max()) == 0.0


def test_local_corr_argument_errors():
    ops = _ops()
    f = torch.zeros(1, 8, 4, 4, device=DEV)
    with pytest.raises(ValueError):
        ops.local_correlation(f, f, 9)
    with pytest.raises(NotImplementedError):
        ops.local_correlation(f, f, 2, padding_mode="border")
    with pytest.raises(ValueError):
        ops.local_correlation(f, f, 2, flow=torch.zeros(1, 2, 5, 4, device=DEV))


# ---- warp / interp / disp_emb ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,layout", [(64, "nhwc"), (9, "nhwc"), (12, "nchw")])
def test_warp_bilinear(dtype, C, layout):
    import torch.nn.functional as F
    B, h, w = 2, 17, 23
    src = H.T(R.normal(f"warp.src.{C}", (B, C, h, w))).to(dtype)
    flow = H.T(R.adversarial_flow("warp.flow", B, h + 3, w - 2, lim=1.1))
    ref = F.grid_sample(src.float(), flow.permute(0, 2, 3, 1), mode="bilinear", align_corners=False)
    s = src.to(DEV)
    if layout == "nhwc":
        s = s.contiguous(memory_format=torch.channels_last)
    out = _ops().warp_bilinear(s, flow.to(DEV))
    assert maxerr(out, ref) <= (1e-5 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,pitch,off", [(9, 24, 9), (9, 24, 10), (3, 8, 3), (12, 32, 14), (16, 40, 17), (5, 16, 8)])
def test_warp_bilinear_few_channels_inside_a_wider_pixel(dtype, C, pitch, off):
    """The scale-1 refiner's call: source = channels [0, C) of a channels-last concat buffer, destination = channels [off, off + C)
    of the SAME buffer (any 2-byte alignment), the source batch-shifted by B / 2 — the one-thread-per-pixel kernel of sampling.hip
    (C <= 16, not a whole number of packets).  Reference: F.grid_sample on the rounded source; the other channels must stay untouched."""
    import torch.nn.functional as F
    B, h, w = 4, 19, 26
    buf = torch.full((B, h, w, pitch), 7.0, device=DEV, dtype=dtype)
    d = buf.permute(0, 3, 1, 2)
    d[:, :C] = H.T(R.normal(f"warps.src.{C}", (B, C, h, w))).to(DEV).to(dtype)
    src = d[:, :C]
    flow = H.T(R.adversarial_flow("warps.flow", B, h, w, lim=1.1), DEV)
    rolled = torch.roll(src.float(), shifts=-(B // 2), dims=0)                                 # y[b] = x[(b + B/2) % B]
    ref = F.grid_sample(rolled, flow.permute(0, 2, 3, 1), mode="bilinear", align_corners=False)
    out = _ops().warp_bilinear(src, flow, out=d[:, off:off + C], batch_shift=B // 2)
    assert out.data_ptr() == buf.data_ptr() + off * buf.element_size()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert maxerr(out, ref) <= tol
    keep = torch.ones(pitch, dtype=torch.bool)
    keep[:C] = False
    keep[off:off + C] = False
    assert bool((buf[..., keep.to(DEV)] == 7.0).all())


@pytest.mark.parametrize("sizes", [((40, 40), (70, 70)), ((70, 70), (140, 140)), ((560, 560), (108, 108)), ((5, 6), (30, 36)), ((7, 9), (7, 9))])
def test_interp_bilinear(sizes):
    import torch.nn.functional as F
    (hi, wi), (ho, wo) = sizes
    x = H.T(R.normal(f"interp.{sizes}", (2, 3, hi, wi)))
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False)
    assert maxerr(_ops().interp_bilinear(x.to(DEV), (ho, wo)), ref) < 2e-6


def test_disp_emb_and_flow_update():
    O = _O()
    B, E, h, w = 2, 6, 11, 13
    flow = H.T(R.coherent_flow("de.flow", B, h, w))
    wgt, bias = H.T(R.normal("de.w", (E, 2, 1, 1))), H.T(R.normal("de.b", (E,)))
    ref = torch.nn.functional.conv2d(1.25 * 1.5 * (flow - O.pixel_grid(B, h, w)), wgt, bias)
    out = _ops().disp_emb(flow.to(DEV), wgt.to(DEV), bias.to(DEV), 1.25 * 1.5)
    assert maxerr(out, ref) < 1e-5
    delta = H.T(R.normal("de.delta", (B, 3, h, w)))
    cert = H.T(R.normal("de.cert", (B, 1, h, w)))
    f2, c2 = _ops().flow_update(flow.clone().to(DEV), cert.to(DEV), delta.to(DEV), 0.01, 0.02)
    assert maxerr(f2, flow + torch.stack((0.01 * delta[:, 0], 0.02 * delta[:, 1]), 1)) < 1e-6
    assert maxerr(c2, cert + delta[:, 2:]) < 1e-6


# ---- cls_to_flow_refine ------------------------------------------------------------------------
def test_cls_to_flow_refine_golden_both_layouts():
    g = H.golden("cls_to_flow_refine")
    x = H.T(cases.cls_inputs(), DEV)
    assert maxerr(_ops().cls_to_flow_refine(x), H.T(g["flow"])) < 2e-6                         # planar (reference layout)
    rows = x.permute(0, 2, 3, 1).reshape(2, 64, 4096)
    rows = torch.cat((rows, torch.full((2, 64, 1), 0.25, device=DEV)), dim=2).contiguous()      # token-major + certainty logit
    flow, cert = _ops().cls_rows_to_flow(rows, 2, 8, 8)
    assert maxerr(flow.permute(0, 2, 3, 1), H.T(g["flow"])) < 2e-6
    assert float((cert - 0.25).abs().max()) == 0.0
    xs = H.T(cases.cls_inputs(res=8, B=1, H=5, W=7), DEV)
    assert maxerr(_ops().cls_to_flow_refine(xs), H.T(g["flow_small"])) < 2e-6


def test_cls_to_flow_refine_full_size_fp16():
    O = _O()
    x = (H.T(R.normal("clsfull", (2, 4096, 40, 40))) * 3).half()
    ref = O.cls_to_flow_refine(x.float())
    clog = H.T(R.normal("clsfull.cert", (2, 1600, 1))).half()
    rows = torch.cat((x.permute(0, 2, 3, 1).reshape(2, 1600, 4096), clog), dim=2).to(DEV)   # 8194-byte rows: every 16-byte phase occurs
    flow, cert = _ops().cls_rows_to_flow(rows, 2, 40, 40)
    assert maxerr(flow.permute(0, 2, 3, 1), ref) < 1e-5
    assert torch.equal(cert.flatten().cpu(), clog.float().flatten())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("res,pad", [(8, 0), (8, 3), (24, 0), (67, 5)])
def test_cls_rows_16bit_vector_path_shapes(dtype, res, pad):
    """The one-pass 16-byte-load kernel on rows whose pitch / start are not 16-byte multiples (boundary packets are read under a
    guard), small and odd class counts (res = 67: 4489 classes, 9 packets on some lanes), arg-max at the first / last class."""
    O = _O()
    C, hw = res * res, 35
    x = (H.T(R.normal(f"clsvec.{res}", (1, C, 5, 7))) * 3).to(dtype)
    x[0, 0, 0, 0] = 30.0                                        # mode = class 0 (mode-1 clamps)
    x[0, C - 1, 0, 1] = 30.0                                    # mode = last class (mode+1 clamps)
    ref = O.cls_to_flow_refine(x.float())
    buf = torch.zeros(hw * (C + 1 + pad) + 8, dtype=dtype, device=DEV)
    for off in (0, 1, 5):                                       # row 0 starts 0 / 2 / 10 bytes past a 16-byte boundary
        rows = buf[off:off + hw * (C + 1 + pad)].view(1, hw, C + 1 + pad)[:, :, :C + 1]
        rows[..., :C] = x.permute(0, 2, 3, 1).reshape(1, hw, C).to(DEV)
        rows[..., C] = 0.5
        flow, cert = _ops().cls_rows_to_flow(rows, 1, 5, 7)
        assert maxerr(flow.permute(0, 2, 3, 1), ref) < 1e-5
        assert float((cert - 0.5).abs().max()) == 0.0


# ---- CosKernel (MFMA) --------------------------------------------------------------------------
@pytest.mark.parametrize("name,dims", [("small", (1, 512, 10, 10)), ("rect", (2, 64, 6, 9))])
def test_cos_kernel_golden(name, dims):
    g = H.golden("gp")
    b, c, h, w = dims
    x = H.T(R.normal(f"gp.{name}.x", (b, c, h, w)), DEV).flatten(2).transpose(1, 2)
    y = H.T(R.normal(f"gp.{name}.y", (b, c, h, w)), DEV).flatten(2).transpose(1, 2)
    assert maxerr(_ops().cos_kernel(x, y), H.T(g[f"{name}_Kxy"])) < 2e-6


def test_cos_kernel_full_size_asymmetric_and_diag():
    O = _O()
    x = H.T(R.normal("cosfull.x", (2, 1600, 512)))
    y = H.T(R.normal("cosfull.y", (2, 1600, 512))) * H.T(R.uniform("cosfull.s", (2, 1600, 1), 0.5, 2.0))
    assert maxerr(_ops().cos_kernel(x.to(DEV), y.to(DEV)), O.cos_kernel(x, y)) < 2e-6
    kyy = _ops().cos_kernel(y.to(DEV), y.to(DEV), diag_add=0.1)
    # on the diagonal c ~ 1 and exp((c-1)/T) amplifies the fp32 rounding of c by 1/T = 5
    assert maxerr(kyy, O.cos_kernel(y, y) + 0.1 * torch.eye(1600)[None]) < 6e-6


def test_spd_solve_against_fp64():
    torch.manual_seed(0)
    for n, m in ((1600, 512), (100, 16), (54, 32), (65, 8)):
        x = torch.nn.functional.normalize(torch.randn(2, n, 64), dim=-1)
        K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n)
        F = torch.randn(2, n, m)
        ref = solve64(K, F)
        X = _ops().spd_solve(K.to(DEV).contiguous(), F.to(DEV))
        e0 = float((X.cpu().double() - ref).abs().max())
        assert e0 < 2e-5 * float(ref.abs().max())
        # one refinement step through the finished factor (forward + back substitution kernels, fp64 residual)
        X1 = _ops().spd_solve(K.to(DEV).contiguous(), F.to(DEV), refine=1)
        e1 = float((X1.cpu().double() - ref).abs().max())
        assert e1 < 2e-6 * float(ref.abs().max()) and e1 <= e0


# ---- match_finalize / kde ----------------------------------------------------------------------
def test_match_finalize_golden():
    g = H.golden("match_post")
    c16, flow, cert = cases.post_inputs()
    warp, certainty = _ops().match_finalize(H.T(flow, DEV), H.T(cert, DEV), H.T(c16, DEV), symmetric=True)
    assert maxerr(warp[0], H.T(g["warp"])) < 1e-6
    assert maxerr(certainty[0], H.T(g["certainty"])) < 1e-6


def test_kde_golden_and_full_size():
    O = _O()
    g = H.golden("kde")
    x = H.T(cases.kde_inputs(), DEV)
    assert maxerr(_ops().kde(x, half=False), H.T(g["fp32"])) < 1e-3
    assert maxerr(_ops().kde(x, half=False, down=4), H.T(g["down4"])) < 1e-3
    # half=True: the reference runs cdist's matmul route (x^2 + y^2 - 2xy) in fp16, which is itself ~15 % off its own fp32
    # value; the kernel goes through the same rounding points, so the reference's fp16 fixture is met to one fp16 ulp of
    # the sum (the only freedom left is the fp32 summation order of 512 terms).
    d16 = _ops().kde(x, half=True)
    assert d16.dtype == torch.float16
    ref16 = H.T(g["fp16"])
    assert float(((d16.float().cpu() - ref16).abs() / ref16).max()) < 1.5e-3
    assert float((d16.float().cpu() == ref16).float().mean()) > 0.95
    xf = H.T(R.uniform("kdefull", (10000, 4), -1, 1))
    ref = O.kde(xf, half=False)
    out = _ops().kde(xf.to(DEV), half=False)
    assert float(((out.cpu() - ref).abs() / ref).max()) < 1e-4


def test_kde_half_full_size_vs_oracle():
    """N = 40 000 (sample()'s size): fp16-arithmetic KDE against the oracle's literal fp16 evaluation (same torch ops as the
    reference, kde.py:6-12); sums of 40 000 fp16 terms agree to fp16 resolution."""
    O = _O()
    x = H.T(R.uniform("kdehalf", (40000, 4), -1, 1))
    ref = O.kde(x, half=True).float()
    out = _ops().kde(x.to(DEV), half=True).float().cpu()
    assert float(((out - ref).abs() / ref).max()) < 2e-3
    assert bool(((out < 10) == (ref < 10)).float().mean() > 0.999)          # the cut sample() applies (matcher.py:491)


# ---- sample(): exponential-race keys -------------------------------------------------------------
def test_race_keys_match_the_oracle_and_select_the_same_set():
    O = _O()
    c = H.T(R.uniform("race.c", (300000,), 0, 0.2))
    c[::7] = 0.0
    for thresh, seed in ((0.05, 3), (-1.0, 11)):
        k_ref = O.race_keys(c, thresh, seed)
        k = _ops().race_keys(c.to(DEV), thresh, seed).cpu()
        assert float(((k - k_ref).abs() / k_ref.clamp_min(1e-30)).max()) < 1e-5        # -log(u): libm vs device logf
        a, b = set(torch.topk(k, 4000).indices.tolist()), set(torch.topk(k_ref, 4000).indices.tolist())
        assert len(a ^ b) <= 2
    assert float(_ops().race_keys(torch.zeros(100, device=DEV), 0.05, 1).max()) == 0.0
    # counting by item identity: a permuted population gives the permuted keys (the second draw of sample() relies on it)
    perm = torch.randperm(c.numel())
    k_perm = _ops().race_keys(c[perm].to(DEV), 0.05, 3, counter=perm.to(DEV)).cpu()
    assert torch.equal(k_perm, _ops().race_keys(c.to(DEV), 0.05, 3).cpu()[perm])
    assert torch.equal(O.race_keys(c[perm], 0.05, 3, counter=perm), O.race_keys(c, 0.05, 3)[perm])
    # the two draws of one sample() call are separate streams, and neither is the first draw of the next seed
    k10 = _ops().race_keys(c.to(DEV), -1.0, 7, stage=0).cpu()
    k11 = _ops().race_keys(c.to(DEV), -1.0, 7, stage=1).cpu()
    k20 = _ops().race_keys(c.to(DEV), -1.0, 8, stage=0).cpu()
    assert not torch.equal(k11, k20) and not torch.equal(k10, k11)
    ref11 = O.race_keys(c, -1.0, 7, stage=1)
    assert float(((k11 - ref11).abs() / ref11.clamp_min(1e-30)).max()) < 1e-5
    pos = c > 0
    e10, e11, e20 = (torch.log(c[pos] / k[pos]) for k in (k10, k11, k20))       # log of the Exp(1) variates themselves

    def corr(a, b):
        return float(torch.corrcoef(torch.stack((a, b)))[0, 1])
    assert abs(corr(e11, e20)) < 0.02 and abs(corr(e10, e20)) < 0.02 and abs(corr(e10, e11)) < 0.02


def test_race_keys_are_a_draw_without_replacement_proportional_to_weight():
    w = torch.cat((torch.full((2000,), 1.0), torch.full((200000,), 0.002))).to(DEV)          # 2000 vs 400 of total mass
    heavy = 0
    for seed in range(20):
        idx = torch.topk(_ops().race_keys(w, -1.0, seed), 1000).indices
        assert idx.unique().numel() == 1000
        heavy += int((idx < 2000).sum())
    # sequential draws without replacement from (2000 x 1.0, 200000 x 0.002): the heavy share of the first 1000 draws is
    # about 0.80 (mass ratio 2000:400 at the start, 1200:399 at the end)
    assert 0.76 < heavy / 20000 < 0.85


# ---- fused residual add + LayerNorm --------------------------------------------------------------
@pytest.mark.parametrize("xdt,ydt,odt", [(torch.float32, torch.float16, torch.float16), (torch.float16, torch.float16, torch.float16),
                                         (torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32, torch.float32)])
@pytest.mark.parametrize("C", [1024, 96, 1536])
def test_add_layernorm_vs_torch(xdt, ydt, odt, C):
    import torch.nn.functional as F
    rows = 777
    x = H.T(R.normal(f"aln.x.{C}", (3, rows // 3, C))).to(xdt).to(DEV)
    y = H.T(R.normal(f"aln.y.{C}", (3, rows // 3, C))).to(ydt).to(DEV)
    ls = H.T(R.uniform(f"aln.ls.{C}", (C,), 0.5, 1.5)).to(DEV)
    g, b = H.T(R.uniform(f"aln.g.{C}", (C,), 0.5, 1.5)).to(DEV), H.T(R.normal(f"aln.b.{C}", (C,), scale=0.1)).to(DEV)
    for use_ls in (False, True):
        xs = x.clone()
        out = _ops().add_layernorm(xs, y, g, b, 1e-6, odt, ls=ls if use_ls else None)
        x_ref = (x.float() + (ls * y.float() if use_ls else y.float())).to(xdt)
        ref = F.layer_norm(x_ref.float(), (C,), g, b, 1e-6).to(odt)
        assert torch.equal(xs, x_ref) or maxerr(xs, x_ref) <= float(x_ref.float().abs().max()) * (2 ** -10 if xdt != torch.float32 else 1e-7)
        tol = {torch.float32: 2e-6, torch.float16: 2 ** -9, torch.bfloat16: 2 ** -6}[odt] * max(1.0, float(ref.float().abs().max()))
        assert out.dtype == odt and maxerr(out, ref) <= tol
    # no add (first LayerNorm of a stack) and no LayerNorm (the cast in front of to_out)
    xs = x.clone()
    assert maxerr(_ops().add_layernorm(xs, None, g, b, 1e-5, odt), F.layer_norm(x.float(), (C,), g, b, 1e-5)) <= \
        {torch.float32: 2e-6, torch.float16: 2 ** -9, torch.bfloat16: 2 ** -6}[odt] * 8
    assert torch.equal(xs, x)
    xs = x.clone()
    cast = _ops().add_layernorm(xs, y, None, None, 0.0, odt)
    assert torch.equal(cast, (x.float() + y.float()).to(xdt).to(odt)) or maxerr(cast, (x.float() + y.float()).to(xdt).to(odt)) <= 2 ** -9 * 8


# ---- batch-shifted second operand (forward_symmetric without the half-swap copy) -----------------
def test_batch_shift_equals_explicit_swap():
    ops = _ops()
    B, C, h, w, r = 4, 64, 21, 18, 3
    f = H.T(R.normal("bs.f", (B, C, h, w))).half().to(DEV).contiguous(memory_format=torch.channels_last)
    flow = H.T(R.coherent_flow("bs.flow", B, h, w), DEV)
    swapped = torch.cat((f[B // 2:], f[:B // 2])).contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.local_correlation(f, f, r, flow=flow, batch_shift=B // 2), ops.local_correlation(f, swapped, r, flow=flow))
    assert torch.equal(ops.warp_bilinear(f, flow, batch_shift=B // 2), ops.warp_bilinear(swapped, flow))
    f32, s32 = f.float().contiguous(), swapped.float().contiguous()                            # planar fp32 paths
    assert torch.equal(ops.local_correlation(f32, f32, r, flow=flow, batch_shift=B // 2), ops.local_correlation(f32, s32, r, flow=flow))
    assert torch.equal(ops.warp_bilinear(f32, flow, batch_shift=B // 2), ops.warp_bilinear(s32, flow))
    rows = f.permute(0, 2, 3, 1).reshape(B, h * w, C)
    srows = swapped.permute(0, 2, 3, 1).reshape(B, h * w, C)
    assert torch.equal(ops.cos_kernel(rows, rows, batch_shift=B // 2), ops.cos_kernel(rows.float().contiguous(), srows.float().contiguous()))


@pytest.mark.parametrize("variant", ["rows8"])
def test_local_corr_rows_kernel_on_concat_slices_with_batch_shift_and_identity_flow(variant):
    """The row-streaming kernel the way the decoder calls it: f0 / f1 are channel slices of wider channels-last buffers (pitch > C,
    slice not at channel 0), the second operand is the first with its batch halves swapped (batch_shift), the output goes into a
    slice of the concat buffer; and flow = None (identity grid).  Reference: the fp32 kernel on the same rounded values."""
    ops = _ops()
    B, C, h, w, r = 4, 256, 27, 22, 2
    K = (2 * r + 1) ** 2
    buf = torch.zeros(B, h, w, 2 * C + K + 15, device=DEV, dtype=torch.float16)
    d = buf.permute(0, 3, 1, 2)
    d[:, 8:8 + C] = H.T(R.normal("rows.slice.f", (B, C, h, w)), DEV).half()
    f = d[:, 8:8 + C]
    dense = f.float().contiguous(memory_format=torch.channels_last)
    swapped = torch.cat((dense[B // 2:], dense[:B // 2])).contiguous(memory_format=torch.channels_last)
    for flow in (H.T(R.coherent_flow("rows.slice.flow", B, h, w), DEV), None):
        ref = ops.local_correlation(dense, swapped, r, flow=flow)
        out = ops.local_correlation(f, f, r, flow=flow, batch_shift=B // 2, variant=variant, out=d[:, 8 + C:8 + C + K])
        assert maxerr(out, ref) <= 2 ** -10 * max(1.0, float(ref.abs().max()))
        assert float(buf[..., 8 + C + K:].abs().max()) == 0.0 and float(buf[..., :8].abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["coherent", "adversarial", "mixed"])
@pytest.mark.parametrize("shape", [(2, 512, 40, 40, 7), (2, 512, 70, 70, 3), (1, 256, 37, 53, 2), (2, 256, 19, 45, 3), (1, 512, 21, 9, 1)])
def test_local_corr_rows_kernel_bands_partition_the_output(shape, kind, monkeypatch):
    """The row-streaming kernel gives a tile several workgroups when a launch is small (local_corr_rows.hip, "bands": each streams a
    band of the tile's box rows / takes a share of the patch pixels and stores only the output rows it owns).  Whatever the band
    count — forced here through ROMA_LC_BANDS, which the launcher reads per call — every output element must be written exactly
    once with the same arithmetic: results are BIT-identical to the one-workgroup-per-tile launch, in both output layouts, into a
    buffer pre-filled with NaN (an element nobody owns would stay NaN), borders / NaN / far-out-of-range flow included."""
    B, C, h, w, r = shape
    ops = _ops()
    f0 = H.T(R.normal(f"lcb.{shape}.f0", (B, C, h, w))).half().to(DEV).contiguous(memory_format=torch.channels_last)
    f1 = H.T(R.normal(f"lcb.{shape}.f1", (B, C, h, w))).half().to(DEV).contiguous(memory_format=torch.channels_last)
    flow = R.coherent_flow(f"lcb.{shape}.flow", B, h, w)
    if kind == "adversarial":
        flow = R.adversarial_flow(f"lcb.{shape}.flow", B, h, w)
    elif kind == "mixed":
        adv = R.adversarial_flow(f"lcb.{shape}.adv", B, h, w)
        flow[:, :, h // 3: 2 * h // 3, w // 4: w // 2] = adv[:, :, h // 3: 2 * h // 3, w // 4: w // 2]
        flow[:, 0, 0, 0] = float("nan")
        flow[:, 1, -1, -1] = 1e30
        flow[:, :, 1, 2] = -7.0
        flow[:, 1, : h // 4] -= 1.5                                                            # windows hanging over the top edge
        flow[:, 1, -(h // 4):] += 1.5                                                          # ... and the bottom edge (virtual row)
    flow = H.T(flow, DEV)
    K = (2 * r + 1) ** 2

    def run(bands, nhwc):
        monkeypatch.setenv("ROMA_LC_BANDS", str(bands))
        out = torch.full((B, K, h, w), float("nan"), dtype=torch.float16, device=DEV)
        if nhwc:
            out = out.contiguous(memory_format=torch.channels_last)
        return ops.local_correlation(f0, f1, r, flow=flow, variant="rows8", out=out)

    for nhwc in (True, False):
        one = run(1, nhwc)
        assert torch.isfinite(one.float()).all()
        for bands in (2, 3, 5, 8):
            assert torch.equal(run(bands, nhwc), one), (bands, nhwc)


def test_cos_kernel_reads_channels_last_slices_in_place():
    ops = _ops()
    buf = torch.randn(2, 10, 12, 200, device=DEV).half()                                       # a wider channels-last buffer
    x = buf[..., :64].reshape(2, 120, 64)
    assert x.data_ptr() == buf.data_ptr() and x.stride(1) == 200
    assert torch.equal(ops.cos_kernel(x, x, diag_add=0.1), ops.cos_kernel(x.float().contiguous(), x.float().contiguous(), diag_add=0.1))


def test_chol_step_and_subst_step_against_torch():
    """One fused forward step (roma_chol_step) and one substitution step in each direction (roma_chol_subst_step) against their
    definitions in include/roma_hip.h, computed with torch in fp64: n = 200 (three full blocks + one of 8 rows), m = 24."""
    from roma_amd import _lib
    from roma_amd._lib import check
    lib = _lib.load()
    torch.manual_seed(3)
    B, n, m, nb = 2, 200, 24, 64
    x = torch.nn.functional.normalize(torch.randn(B, n, 32), dim=-1)
    K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n)
    F = torch.randn(B, n, m)
    A = torch.cat((K, F), dim=2).to(DEV).contiguous()
    A0 = A.double().cpu()
    st = torch.cuda.current_stream().cuda_stream
    W = torch.zeros(B, 4, nb, nb, device=DEV)
    R = torch.zeros(B, 4, nb, n + m, device=DEV)
    info = torch.zeros(B, dtype=torch.int32, device=DEV)
    check(lib.roma_chol_diag_block(A.data_ptr(), n + m, A.stride(0), W.data_ptr(), nb, W.stride(0), nb, B, info.data_ptr(), 0, st), "diag")
    check(lib.roma_chol_step(A.data_ptr(), n + m, A.stride(0), n, n + m, 0, nb, W[:, 0].data_ptr(), nb, W.stride(0), R[:, 0].data_ptr(), n + m,
                             R.stride(0), W[:, 1].data_ptr(), nb, W.stride(0), info.data_ptr(), nb, B, st), "step")
    torch.cuda.synchronize()
    L00 = torch.linalg.cholesky(A0[:, :nb, :nb])
    W0 = torch.linalg.inv(L00)
    assert float((W[:, 0].double().cpu() - W0).abs().max()) < 1e-4
    Rref = W0 @ A0[:, :nb, nb:]                                           # (B, 64, n - 64 + m)
    assert float((R[:, 0, :, :n + m - nb].double().cpu() - Rref).abs().max()) < 1e-4
    T = A0[:, nb:, nb:] - Rref[:, :, :n - nb].transpose(1, 2) @ Rref      # trailing update, full (the kernel keeps tiles J >= I)
    got = A.double().cpu()[:, nb:, nb:]
    for i0 in range(0, n - nb, 64):                                       # tiles on or right of the diagonal
        i1 = min(i0 + 64, n - nb)
        if i0 == 0:                                                       # next diagonal block: factored in place (lower), W_1 = L^-1
            L11 = torch.linalg.cholesky(T[:, :64, :64])
            assert float((torch.tril(got[:, :64, :64]) - L11).abs().max()) < 1e-4
            assert float((W[:, 1].double().cpu() - torch.linalg.inv(L11)).abs().max()) < 2e-4
            assert float((got[:, i0:i1, 64:] - T[:, i0:i1, 64:]).abs().max()) < 1e-4
        else:
            assert float((got[:, i0:i1, i0:] - T[:, i0:i1, i0:]).abs().max()) < 1e-4
    assert int(info.abs().max()) == 0
    # substitution steps on a synthetic factor: panels P_k = L[e_k:, block k]^T, W_k = inv(L_kk)
    Lf = torch.linalg.cholesky(A0[:, :n, :n])
    S = 4
    Wall = torch.zeros(B, S, nb, nb, dtype=torch.float64)
    Pall = torch.zeros(B, S, nb, n + m, dtype=torch.float64)
    for k in range(S):
        j, e = k * nb, min((k + 1) * nb, n)
        Wall[:, k, :e - j, :e - j] = torch.linalg.inv(Lf[:, j:e, j:e])
        Pall[:, k, :e - j, :n - e] = Lf[:, e:, j:e].transpose(1, 2)
    Wd, Pd = Wall.float().to(DEV), Pall.float().to(DEV)
    G = torch.randn(B, n, m)
    for direction, s_blk in ((+1, 1), (-1, 2)):
        Td = G.clone().to(DEV).contiguous()
        Xd = torch.zeros(B, n, m, device=DEV)
        check(lib.roma_chol_subst_step(direction, Wd[:, s_blk].data_ptr(), nb, Wd.stride(0), Pd.data_ptr(), Pd.stride(0), Pd.stride(1), n + m,
                                       Td.data_ptr(), Td.stride(0), nb * m, m, 0, Xd.data_ptr(), m, Xd.stride(0), n, m, nb, s_blk, B, st), "subst")
        torch.cuda.synchronize()
        js, es = s_blk * nb, min((s_blk + 1) * nb, n)
        Ws = Wall[:, s_blk, :es - js, :es - js]
        Gs = G.double()[:, js:es]
        Xs = (Ws @ Gs) if direction > 0 else (Ws.transpose(1, 2) @ Gs)
        assert float((Xd.double().cpu()[:, js:es] - Xs).abs().max()) < 1e-4
        Tref = G.double().clone()
        if direction > 0:
            Tref[:, es:] -= Lf[:, es:, js:es] @ Xs                        # T_i -= L[i, s] Y_s for i > s
        else:
            Tref[:, :js] -= Lf[:, js:es, :js].transpose(1, 2) @ Xs        # T_i -= L[s, i]^T X_s for i < s
        assert float((Td.double().cpu() - Tref).abs().max()) < 1e-4


# ---- spd_solve robustness ------------------------------------------------------------------------
def test_spd_solve_raises_on_a_matrix_that_is_not_positive_definite():
    from roma_amd._lib import RomaHipError
    ops = _ops()
    n = 200
    x = torch.nn.functional.normalize(torch.randn(2, n, 32), dim=-1)
    K = (torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n)).to(DEV)
    F = torch.randn(2, n, 8, device=DEV)
    ops.spd_solve(K.clone(), F)                                                                 # fine
    bad = K.clone()
    bad[1, 130, 130] = -5.0
    with pytest.raises(RomaHipError, match="matrix 1 .*block step 2"):
        ops.spd_solve(bad, F)
    nan = K.clone()
    nan[0, 3, 3] = float("nan")
    with pytest.raises(RomaHipError, match="matrix 0"):
        ops.spd_solve(nan, F)
    ops.spd_solve(bad, F, check="defer")                                                        # deferred: raises at raise_pending()
    with pytest.raises(RomaHipError):
        ops.raise_pending()
    ops.raise_pending()                                                                          # queue is empty again
    # the scoped form match_tensors / match_encoded use: examined when the block ends, dropped when it ends by an exception
    with pytest.raises(RomaHipError, match="matrix 1"):
        with ops.deferred_spd_checks():
            assert ops.deferring()
            ops.spd_solve(bad, F, check="defer")
    assert not ops.deferring()
    with pytest.raises(KeyError):
        with ops.deferred_spd_checks():
            ops.spd_solve(bad, F, check="defer")
            raise KeyError("something else went wrong in the pipeline")
    ops.raise_pending()                                                                          # nothing stale is left behind
    with ops.deferred_spd_checks():
        ops.spd_solve(K.clone(), F, check="defer")                                               # a good solve passes through


def test_chol_step_rejects_a_narrow_block_that_is_not_the_last():
    """roma_chol_step factors the NEXT diagonal block at min(64, n - e) rows: a caller block size other than 64 is only legal for the
    last block (nb = 32 used to overrun the W slots)."""
    from roma_amd import _lib
    lib = _lib.load()
    n, m, nb = 100, 4, 32
    A = torch.zeros(1, n, n + m, device=DEV)
    W = torch.zeros(1, 4, 64, 64, device=DEV)
    R = torch.zeros(1, 4, 64, n + m, device=DEV)
    info = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.roma_chol_step(A.data_ptr(), A.stride(1), A.stride(0), n, n + m, 0, nb, W.data_ptr(), 64, W.stride(0), R.data_ptr(), n + m,
                            R.stride(0), W[:, 1].data_ptr(), 64, W.stride(0), info.data_ptr(), nb, 1, st)
    assert rc == -5 and b"block size 32" in lib.roma_last_error()


def test_spd_solve_under_tunableop():
    """Round 1 recorded `HIP error: invalid argument` from the in-place strided trailing update while PYTORCH_TUNABLEOP was
    enabled (TunableOp's candidate sweep); spd_solve now runs its GEMMs with TunableOp off and restores the setting."""
    import torch.cuda.tunable as tun
    was = tun.is_enabled()
    tun.enable(True)
    try:
        tun.set_filename("/tmp/roma_tunableop_test.csv")           # keep its result file out of the repository
        x = torch.nn.functional.normalize(torch.randn(2, 300, 32), dim=-1)
        K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(300)
        F = torch.randn(2, 300, 16)
        X = _ops().spd_solve(K.to(DEV), F.to(DEV))
        assert tun.is_enabled()
        ref = solve64(K, F)
        assert float((X.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    finally:
        tun.enable(was)


# ---- depthwise 5x5 + BN + ReLU -----------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,h,w", [(24, 19, 21), (576, 12, 10), (144, 7, 33)])
def test_dwconv5x5_bn_relu(dtype, C, h, w):
    import torch.nn.functional as F
    B = 2
    x = H.T(R.normal(f"dw.x.{C}", (B, C, h, w))).to(dtype)
    wt = H.T(R.normal(f"dw.w.{C}", (C, 1, 5, 5), scale=0.2))
    scale, shift = H.T(R.uniform("dw.sc", (C,), 0.5, 1.5)), H.T(R.normal("dw.sh", (C,), scale=0.1))
    ref = F.relu(F.conv2d(x.float(), wt, None, 1, 2, groups=C) * scale[None, :, None, None] + shift[None, :, None, None])
    out = _ops().dwconv5x5_bn_relu(x.to(DEV).contiguous(memory_format=torch.channels_last),
                                   wt.reshape(C, 25).t().contiguous().to(DEV), scale.to(DEV), shift.to(DEV))
    assert maxerr(out, ref) <= (2e-5 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype,C,h,w,pitch", [(torch.float16, 576, 150, 147, 576), (torch.float16, 1152, 108, 108, 1152),
                                               (torch.bfloat16, 1408, 90, 93, 1408), (torch.float16, 144, 300, 301, 160)])
def test_dwconv5x5_persistent_variant_shapes(dtype, C, h, w, pitch):
    """Sizes at which roma_dwconv5x5_bn_relu takes the persistent kernel (taps in LDS, software-pipelined rows; >= 2 rounds of
    work per workgroup): both workgroup sizes (taps <= / > 76 KB), widths that are not multiples of the 8-pixel strip, border
    strips on all four sides, a channel pitch wider than C."""
    import torch.nn.functional as F
    B = 2
    x = H.T(R.normal(f"dwp.x.{C}", (B, C, h, w))).to(dtype)
    wt = H.T(R.normal(f"dwp.w.{C}", (C, 1, 5, 5), scale=0.2))
    scale, shift = H.T(R.uniform("dwp.sc", (C,), 0.5, 1.5)), H.T(R.normal("dwp.sh", (C,), scale=0.1))
    ref = F.relu(F.conv2d(x.float(), wt, None, 1, 2, groups=C) * scale[None, :, None, None] + shift[None, :, None, None])
    ops = _ops()
    xin = ops.nhwc_empty(B, C, h, w, dtype, DEV, pitch=pitch)
    xin.copy_(x.to(DEV))
    out = ops.nhwc_empty(B, C, h, w, dtype, DEV, pitch=pitch)
    ops.dwconv5x5_bn_relu(xin, wt.reshape(C, 25).t().contiguous().to(DEV), scale.to(DEV), shift.to(DEV), out=out)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert maxerr(out, ref) <= tol * max(1.0, float(ref.abs().max()) / 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C", [8, 24, 32])
def test_pointwise_small(dtype, C):
    M = 5003
    x = H.T(R.normal(f"pw.x.{C}", (M, C))).to(dtype)
    wt = H.T(R.normal(f"pw.w.{C}", (C, C), scale=0.3))
    b = H.T(R.normal(f"pw.b.{C}", (C,)))
    ref = x.float() @ wt + b
    out = _ops().pointwise_small(x.to(DEV), wt.to(DEV), b.to(DEV))
    tol = {torch.float32: 1e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]
    assert out.dtype == dtype and maxerr(out, ref) <= tol * max(1.0, float(ref.abs().max()))
    # strided rows (a channel slice of a wider channels-last buffer)
    wide = torch.zeros(M, C + 16, dtype=dtype, device=DEV)
    wide[:, :C] = x.to(DEV)
    assert maxerr(_ops().pointwise_small(wide[:, :C], wt.to(DEV), b.to(DEV)), ref) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,pitch", [(24, 24), (144, 144), (569, 576), (1377, 1384)])
def test_refiner_head(dtype, C, pitch):
    B, h, w = 2, 9, 11
    x = torch.zeros(B, h, w, pitch)
    x[..., :C] = H.T(R.normal(f"rh.x.{C}", (B, h, w, C)))
    x = x.to(dtype)
    wo = torch.zeros(pitch, 3)
    wo[:C] = H.T(R.normal(f"rh.w.{C}", (C, 3), scale=1.0 / math.sqrt(C)))
    bo = H.T(R.normal("rh.b", (3,)))
    flow = H.T(R.coherent_flow("rh.flow", B, h, w))
    cert = H.T(R.normal("rh.cert", (B, 1, h, w)))
    d = (x.float().reshape(-1, pitch) @ wo + bo).reshape(B, h, w, 3).permute(0, 3, 1, 2)
    ref_flow = flow + torch.stack((0.01 * d[:, 0], 0.02 * d[:, 1]), 1)
    f2, c2, delta = _ops().refiner_head(x.to(DEV), wo.to(DEV), bo.to(DEV), flow.clone().to(DEV), cert.to(DEV), 0.01, 0.02, want_delta=True)
    tol = 1e-5 if dtype == torch.float32 else 1e-4
    assert maxerr(delta, d) < tol * 10 and maxerr(f2, ref_flow) < tol and maxerr(c2, cert + d[:, 2:]) < tol * 10
    f3, c3 = _ops().refiner_head(x.to(DEV), wo.to(DEV), bo.to(DEV), flow.clone().to(DEV), None, 0.01, 0.02)
    assert maxerr(c3, d[:, 2:]) < tol * 10


# ---- TinyRoMa fused corr + pos_embed -----------------------------------------------------------
def test_tiny_corr_posembed_golden():
    g = H.golden("tiny")
    f0, f1 = H.T(R.normal("tiny.f0", (2, 64, 8, 12)), DEV), H.T(R.normal("tiny.f1", (2, 64, 8, 12)), DEV)
    assert maxerr(_ops().tiny_corr_posembed(f0, f1, exact=True), H.T(g["pos_embed_exact_b2"])) < 1e-5
    assert maxerr(_ops().tiny_corr_posembed(f0[:1], f1[:1], exact=False), H.T(g["pos_embed_fast_b1"])) < 1e-5


def test_tiny_corr_posembed_full_size():
    O = _O()
    f0, f1 = H.T(R.normal("tinyfull.f0", (1, 64, 60, 80))) * 2, H.T(R.normal("tinyfull.f1", (1, 64, 60, 80))) * 2
    cv = O.tiny_corr_volume(f0, f1)
    assert maxerr(_ops().tiny_corr_posembed(f0.to(DEV), f1.to(DEV), exact=True), O.tiny_pos_embed(cv, True)) < 2e-5
    assert maxerr(_ops().tiny_corr_posembed(f0.to(DEV), f1.to(DEV), exact=False), O.tiny_pos_embed(cv, False)) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,kpad,B,H,W", [(144, 160, 1, 37, 45), (24, 32, 2, 19, 50), (144, 160, 2, 8, 16), (32, 32, 1, 64, 64), (160, 160, 1, 23, 17),
                                         (24, 32, 1, 16, 16), (24, 32, 3, 33, 17), (24, 32, 1, 400, 376), (24, 32, 1, 5, 3)])
def test_refiner_block_vs_torch(dtype, C, kpad, B, H, W):
    """Fused ConvRefiner block vs conv2d(groups)+BN+ReLU+1x1 in fp32 on the same (rounded) operands — matcher.py:77-103.
    C = 24 runs the kernel with the depthwise on the matrix cores (16 x 16-pixel tiles, persistent workgroups: 400 x 376 is 600 tiles for
    512 workgroups, so the two-tile-deep halo pipeline wraps; 5 x 3 is one partial tile)."""
    import torch.nn.functional as F
    ops, dev = _ops(), DEV
    g = torch.Generator().manual_seed(C * 1000 + H)
    x = torch.randn(B, H, W, C, generator=g).to(dev).to(dtype)
    w25 = torch.zeros(25, kpad)
    w25[:, :C] = torch.randn(25, C, generator=g) * 0.2
    scale = torch.zeros(kpad)
    shift = torch.zeros(kpad)
    bias = torch.zeros(kpad)
    scale[:C] = torch.rand(C, generator=g) + 0.5
    shift[:C] = torch.randn(C, generator=g) * 0.3
    bias[:C] = torch.randn(C, generator=g) * 0.1
    wt = torch.zeros(kpad, kpad)
    wt[:C, :C] = torch.randn(C, C, generator=g) / C ** 0.5
    w25d, wtd = w25.to(dev).to(dtype), wt.to(dev).to(dtype)
    out = ops.refiner_block(x, w25d, scale.to(dev), shift.to(dev), wtd, bias.to(dev), C)
    # fp32 reference on the rounded operands; the intermediate is rounded to `dtype` as in the kernel (and the reference's autocast)
    xf = x.float().permute(0, 3, 1, 2)
    t = F.conv2d(xf, w25d.float()[:, :C].t().reshape(C, 1, 5, 5), padding=2, groups=C)
    t = torch.relu(t * scale[:C].to(dev).view(1, C, 1, 1) + shift[:C].to(dev).view(1, C, 1, 1)).to(dtype).float()
    ref = torch.einsum("bkhw,nk->bhwn", t, wtd.float()[:C, :C]) + bias[:C].to(dev)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    err = (out.float() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 56, 40), (1, 512, 70, 70), (3, 5, 7, 9)])
def test_bias_relu_inplace(shape, dtype):
    x = H.T(R.normal(f"br.{shape}", shape), DEV).to(dtype)
    b = H.T(R.normal(f"br.b.{shape}", (shape[1],)), DEV).to(dtype)
    ref = torch.relu(x.float() + b.float().view(1, -1, 1, 1)).to(dtype)
    out = _ops().bias_relu_(x.clone(), b)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 56, 40), (1, 128, 140, 140), (3, 5, 6, 10), (1, 3, 4, 2)])
def test_bias_relu_pool2_inplace(shape, dtype):
    """bias + ReLU in place and the 2x2 max-pool of the result in one pass == the two library passes (encoders.py:68-78)."""
    x = H.T(R.normal(f"brp.{shape}", shape), DEV).to(dtype)
    b = H.T(R.normal(f"brp.b.{shape}", (shape[1],)), DEV).to(dtype)
    ref = torch.relu(x.float() + b.float().view(1, -1, 1, 1)).to(dtype)
    xs = x.clone()
    pooled = _ops().bias_relu_pool2_(xs, b)
    assert torch.equal(xs, ref)
    assert torch.equal(pooled, torch.nn.functional.max_pool2d(ref.float(), 2, 2).to(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 64, 56, 40), (1, 512, 70, 70), (2, 128, 6, 10), (1, 256, 35, 35)])
def test_bias_relu_and_pool_channels_last(shape, dtype):
    """The channels-last variants (the layout VGG19 runs in with MIOpen's solver search on): same results as the library passes,
    the map updated in place and still channels-last, the pooled map channels-last too."""
    x = H.T(R.normal(f"brn.{shape}", shape), DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    b = H.T(R.normal(f"brn.b.{shape}", (shape[1],)), DEV).to(dtype)
    ref = torch.relu(x.float() + b.float().view(1, -1, 1, 1)).to(dtype)
    xs = x.clone(memory_format=torch.preserve_format)
    assert _ops().bias_relu_(xs, b) is xs and xs.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(xs, ref)
    if shape[2] % 2 == 0 and shape[3] % 2 == 0:
        xs = x.clone(memory_format=torch.preserve_format)
        pooled = _ops().bias_relu_pool2_(xs, b)
        assert torch.equal(xs, ref) and pooled.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(pooled, torch.nn.functional.max_pool2d(ref.float(), 2, 2).to(dtype))


@pytest.mark.parametrize("shape", [(2, 37, 45), (1, 8, 16), (1, 140, 140), (3, 5, 3)])
def test_refiner_block_wide_vs_the_two_kernel_path(shape):
    """The fused D = 576 block (depthwise 5x5 + BN + ReLU + 1x1 in one kernel, matcher.py:77-103) against the path it replaces —
    ops.dwconv5x5_bn_relu (checked against torch elsewhere in this file) followed by an fp32 matmul on the fp16-rounded
    intermediate — on maps with partial tiles on both borders, and against torch's own depthwise conv."""
    ops = _ops()
    B, h, w = shape
    D = 576
    g = torch.Generator().manual_seed(h * 1000 + w)
    x = torch.randn(B, h, w, D, generator=g).half().to(DEV)
    w25 = (torch.randn(25, D, generator=g) * 0.2).to(DEV)
    scale = (torch.rand(D, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(D, generator=g) * 0.1).to(DEV)
    wt = (torch.randn(D, D, generator=g) / D ** 0.5).half().to(DEV)          # [out][in]
    bias = (torch.randn(D, generator=g) * 0.1).to(DEV)
    w25 = w25.half().float()                                                  # the fused kernel holds its taps in fp16 (autocast semantics)
    out = ops.refiner_block_wide(x, ops.refiner_wide_taps(w25), scale, shift, ops.refiner_wide_pack(wt), bias)
    t = ops.dwconv5x5_bn_relu(x.permute(0, 3, 1, 2), w25, scale, shift).permute(0, 2, 3, 1)
    ref = (t.float().reshape(-1, D) @ wt.float().t() + bias).reshape(B, h, w, D)
    tol = 2 ** -9 * max(1.0, float(ref.abs().max()))
    assert maxerr(out, ref) <= tol
    # and the depthwise half against torch (fp32 conv on the same rounded input)
    tt = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w25.t().reshape(D, 1, 5, 5), padding=2, groups=D)
    tt = torch.relu(tt * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).half().float().permute(0, 2, 3, 1)
    ref2 = (tt.reshape(-1, D) @ wt.float().t() + bias).reshape(B, h, w, D)
    assert maxerr(out, ref2) <= 2 * tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,kpad,M", [(144, 160, 4999), (144, 160, 64), (160, 160, 1000), (48, 160, 333), (24, 32, 777)])
def test_pointwise_mfma_vs_torch(C, kpad, M, dtype):
    """Streaming MFMA 1x1 conv (matcher.py:102) vs an fp32 matmul on the same rounded operands."""
    g = torch.Generator().manual_seed(C + M)
    x = torch.randn(M, C, generator=g).to(DEV).to(dtype)
    wt = torch.zeros(kpad, kpad)
    wt[:C, :C] = torch.randn(C, C, generator=g) / C ** 0.5
    bias = torch.zeros(kpad)
    bias[:C] = torch.randn(C, generator=g) * 0.1
    wtd = wt.to(DEV).to(dtype)
    out = _ops().pointwise_mfma(x, wtd, bias.to(DEV), C)
    ref = x.float() @ wtd.float()[:C, :C].t() + bias[:C].to(DEV)
    tol = (2 ** -10 if dtype == torch.float16 else 2 ** -7) * max(1.0, float(ref.abs().max()))
    assert maxerr(out, ref) <= tol


# ---- attention ------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,N,nv,Hh", [(2, 1664, 1601, 16), (1, 128, None, 2), (2, 200, 137, 3), (1, 77, 77, 1), (1, 64, 3, 1)])
def test_attention_vs_torch_sdpa(dtype, B, N, nv, Hh):
    """ops.attention (flash-style HIP kernel reading q/k/v in place from the qkv projection's output) against fp32 softmax attention on the
    same rounded operands: the DINOv2 shape (1 601 real tokens row-padded to 1 664), ragged sizes, fewer keys than one tile."""
    g = torch.Generator().manual_seed(N * 7 + Hh)
    qkv = (torch.randn(B, N, 3, Hh, 64, generator=g) * 1.5).to(dtype).to(DEV)
    out = _ops().attention(qkv, nv)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))                    # (B, H, N, 64)
    n = N if nv is None else nv
    p = torch.softmax(q @ k[:, :, :n].transpose(-1, -2) * 64 ** -0.5, dim=-1)
    ref = (p @ v[:, :, :n]).permute(0, 2, 1, 3).reshape(B, N, Hh * 64)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert out.shape == ref.shape and out.dtype == dtype
    assert maxerr(out, ref) < tol * max(1.0, float(ref.abs().max()))
