"""Operator-level seams of the reference, backed by the gfx950 kernels in libroma_hip.so.

Same names, arguments and error behaviour as the reference's module-level functions
(`local_correlation` romatch/utils/local_correlation.py:4, `cls_to_flow_refine` romatch/utils/utils.py:301,
`kde` romatch/utils/kde.py:4), plus the lower-level helpers the matcher uses.  Tensors must live on a ROCm
device; anything else raises — there is no CPU or eager-PyTorch fallback here by design.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib
from ._lib import ROMA_BF16, ROMA_F16, ROMA_F32, ROMA_NCHW, ROMA_NHWC, check

_DT = {torch.float32: ROMA_F32, torch.float16: ROMA_F16, torch.bfloat16: ROMA_BF16}


class KernelTimer:
    """Optional HIP-event timing of individual kernel launches ON THE LAUNCH STREAM (torch's current stream).
    bench.py enables it over the timed region to report the roofline of the dominant kernel; off by default."""

    def __init__(self):
        self.enabled = False
        self.bracket_overhead_s = 0.0
        self.records = []          # (name, algorithmic_bytes, start_event, end_event, tag)

    def start(self):
        self.records.clear()
        # calibrate the cost of an empty start/end bracket on this stream (a few microseconds of command-processor time
        # that rocprofv3's kernel timestamps do not contain); subtracted per launch in summary()
        torch.cuda.synchronize()
        pairs = []
        for _ in range(32):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        gaps = sorted(s.elapsed_time(e) * 1e-3 for s, e in pairs)
        self.bracket_overhead_s = gaps[len(gaps) // 2]
        self.enabled = True

    def stop(self):
        self.enabled = False

    def wrap(self, name, nbytes, tag, launch):
        if not self.enabled or torch.cuda.is_current_stream_capturing():
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = launch()
        e.record()
        self.records.append((name, nbytes, s, e, tag))
        return out

    def summary(self):
        """{name: {"launches", "bytes", "seconds", "by_tag": {tag: [launches, bytes, seconds]}}} (call after a sync)."""
        out = {}
        for name, nbytes, s, e, tag in self.records:
            d = out.setdefault(name, {"launches": 0, "bytes": 0, "seconds": 0.0, "by_tag": {}})
            t = max(s.elapsed_time(e) * 1e-3 - self.bracket_overhead_s, 1e-7)
            d["launches"] += 1
            d["bytes"] += nbytes
            d["seconds"] += t
            bt = d["by_tag"].setdefault(tag, [0, 0, 0.0])
            bt[0] += 1
            bt[1] += nbytes
            bt[2] += t
        return out


TIMER = KernelTimer()


def _need_gpu(*ts):
    """Every tensor lives on ONE ROCm device, and that device is the current one (the launch goes to torch's current
    stream, which belongs to the current device)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("roma_amd kernels run on an MI355X (ROCm device tensors) only; got a "
                               f"{t.device} tensor and there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"roma_amd kernel arguments live on different devices ({dev} and {t.device})")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError(f"roma_amd kernel arguments live on {dev} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in `with torch.cuda.device(t.device):`")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {t.dtype}") from None


def _p(t):
    return None if t is None else t.data_ptr()


def feat_layout(t: torch.Tensor):
    """(layout, pitch, tensor) of a (B,C,H,W) feature map: channels-last storage (possibly a channel slice of a wider
    channels-last buffer) -> NHWC, contiguous -> NCHW, anything else is made contiguous."""
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if sc == 1 and sh == W * sw and sb == H * W * sw and sw >= C:
        return ROMA_NHWC, sw, t
    if t.is_contiguous():
        return ROMA_NCHW, C, t
    t = t.contiguous()
    return ROMA_NCHW, C, t


def nhwc_empty(B, C, H, W, dtype, device, pitch=None):
    """A (B,C,H,W)-shaped view of fresh channels-last storage with `pitch` >= C channels per pixel."""
    pitch = pitch or C
    buf = torch.empty((B, H, W, pitch), dtype=dtype, device=device)
    return buf[..., :C].permute(0, 3, 1, 2)


def local_correlation(feature0, feature1, local_radius, padding_mode="zeros", flow=None, sample_mode="bilinear", out=None,
                      batch_shift=0, variant="auto"):
    """romatch/utils/local_correlation.py:4-48.  Returns (B,(2r+1)^2,h,w) in feature0's dtype and memory format
    (or fills `out`, e.g. a channel slice of the refiner's channels-last concat buffer).  batch_shift: feature0[b] meets
    feature1[(b + batch_shift) % B] (forward_symmetric passes the same map twice with batch_shift = B/2).  variant: "auto" or
    one of "tile8x4" / "tile8x8" / "rows8" — the kernels for 16-bit channels-last inputs (roma_hip.h)."""
    if padding_mode != "zeros" or sample_mode != "bilinear":
        raise NotImplementedError("only padding_mode='zeros', sample_mode='bilinear' (the modes RoMa uses)")
    _need_gpu(feature0, feature1, flow, out)
    if feature0.shape != feature1.shape or feature0.dtype != feature1.dtype:
        raise ValueError("feature0/feature1 must have the same shape and dtype")
    B, C, H, W = feature0.shape
    r = int(local_radius)
    K = (2 * r + 1) ** 2
    l0, p0, f0 = feat_layout(feature0)
    l1, p1, f1 = feat_layout(feature1)
    if l0 != l1:                                   # one layout per call: bring f1 to f0's
        f1 = f1.contiguous(memory_format=torch.channels_last if l0 == ROMA_NHWC else torch.contiguous_format)
        l1, p1, f1 = feat_layout(f1)
    if flow is not None:
        if tuple(flow.shape) != (B, 2, H, W):
            raise ValueError(f"flow must be (B,2,h,w), got {tuple(flow.shape)}")
        flow = flow.float().contiguous()
    if out is None:
        out = nhwc_empty(B, K, H, W, feature0.dtype, feature0.device) if l0 == ROMA_NHWC else \
            torch.empty((B, K, H, W), dtype=feature0.dtype, device=feature0.device)
    lo, po, o = feat_layout(out)
    if o is not out or tuple(out.shape) != (B, K, H, W) or out.dtype != feature0.dtype:
        raise ValueError("out must be a (B,K,h,w) tensor of the input dtype, contiguous or channels-last")
    es = f0.element_size()
    # algorithmic bytes of SURVEY §8(d): f0 + f1 + flow(fp32) + out
    nbytes = 2 * B * C * H * W * es + B * 2 * H * W * 4 + B * K * H * W * es
    TIMER.wrap("local_corr", nbytes, f"C{C}_h{H}x{W}_r{r}",
               lambda: check(_lib.load().roma_local_corr(_p(f0), _p(f1), _p(flow), _p(out), B, C, H, W, r, _dt(f0), l0, p0, p1,
                                                         lo, po, int(batch_shift) % B, _lib.LC_VARIANTS[variant], _stream()), "roma_local_corr"))
    return out


def warp_bilinear(src, flow, out=None, batch_shift=0):
    """F.grid_sample(src, flow.permute(0,2,3,1), mode='bilinear', align_corners=False) — matcher.py:109.
    batch_shift: output item b samples src[(b + batch_shift) % B]."""
    _need_gpu(src, flow, out)
    B, C, Hs, Ws = src.shape
    _, _, H, W = flow.shape
    ls, ps, s = feat_layout(src)
    flow = flow.float().contiguous()
    if out is None:
        out = nhwc_empty(B, C, H, W, src.dtype, src.device) if ls == ROMA_NHWC else torch.empty((B, C, H, W), dtype=src.dtype, device=src.device)
    lo, po, o = feat_layout(out)
    if o is not out:
        raise ValueError("out must be contiguous or channels-last")
    check(_lib.load().roma_warp_bilinear(_p(s), _p(flow), _p(out), B, C, Hs, Ws, H, W, _dt(s), ls, ps, lo, po, int(batch_shift) % B, _stream()),
          "roma_warp_bilinear")
    return out


def disp_emb(flow, weight, bias, gain, out=None, dtype=torch.float32):
    """Conv1x1(2->E)(gain*(flow - identity_grid)) — matcher.py:111-120.  weight (E,2[,1,1]), bias (E)."""
    _need_gpu(flow, weight, bias, out)
    B, _, H, W = flow.shape
    E = weight.shape[0]
    flow = flow.float().contiguous()
    w = weight.reshape(E, 2).float().contiguous()
    bvec = bias.float().contiguous()
    if out is None:
        out = torch.empty((B, E, H, W), dtype=dtype, device=flow.device)
    lo, po, o = feat_layout(out)
    if o is not out:
        raise ValueError("out must be contiguous or channels-last")
    check(_lib.load().roma_disp_emb(_p(flow), _p(w), _p(bvec), _p(out), B, E, H, W, float(gain), _dt(out), lo, po, _stream()),
          "roma_disp_emb")
    return out


def interp_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for fp32 (B,C,H,W) maps — matcher.py:349-360,408-417."""
    _need_gpu(x)
    B, C, Hi, Wi = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    x = x.float().contiguous()
    y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_lib.load().roma_interp_bilinear(_p(x), _p(y), B * C, Hi, Wi, Ho, Wo, _stream()), "roma_interp_bilinear")
    return y


def flow_update(flow, certainty, delta, sx, sy):
    """In place: flow += (sx*delta[:,0], sy*delta[:,1]); returns (flow, certainty + delta[:,2:3]) — matcher.py:397-402.
    `certainty` may be None (the reference's 0.0 at the coarsest scale)."""
    _need_gpu(flow, certainty, delta)
    B, _, H, W = flow.shape
    assert flow.dtype == torch.float32 and flow.is_contiguous() and delta.dtype == torch.float32 and delta.is_contiguous()
    cert_in = None if certainty is None else certainty.float().contiguous()
    cert = torch.empty((B, 1, H, W), dtype=torch.float32, device=flow.device)
    check(_lib.load().roma_flow_update(_p(flow), _p(cert), _p(cert_in), _p(delta), B, H, W, float(sx), float(sy), _stream()),
          "roma_flow_update")
    return flow, cert


def cls_to_flow_refine(cls):
    """romatch/utils/utils.py:301-323.  cls (B,C,H,W) logits (any strides) -> (B,H,W,2) fp32."""
    _need_gpu(cls)
    B, C, H, W = cls.shape
    sb, sc, sh, sw = cls.stride()
    if sh != W * sw:
        cls = cls.contiguous()
        sb, sc, sh, sw = cls.stride()
    flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=cls.device)
    check(_lib.load().roma_cls_to_flow_refine(_p(cls), _p(flow), None, B, C, H * W, sb, sc, sw, _dt(cls), _stream()),
          "roma_cls_to_flow_refine")
    return flow.permute(0, 2, 3, 1)


def cls_rows_to_flow(rows, B, H, W):
    """Token-major logits straight from `to_out`: rows (B, H*W, C+1) -> flow (B,2,H,W), certainty (B,1,H,W), fp32.
    transformer/__init__.py:42-45 + utils.py:301-323 + matcher.py:383-385 in one kernel."""
    _need_gpu(rows)
    assert rows.dim() == 3 and rows.shape[0] == B and rows.shape[1] == H * W and rows.stride(2) == 1
    C = rows.shape[2] - 1
    flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=rows.device)
    cert = torch.empty((B, 1, H, W), dtype=torch.float32, device=rows.device)
    check(_lib.load().roma_cls_to_flow_refine(_p(rows), _p(flow), _p(cert), B, C, H * W, rows.stride(0), 1, rows.stride(1),
                                              _dt(rows), _stream()), "roma_cls_to_flow_refine")
    return flow, cert


def _rows(t):
    """(B,N,D) rows with unit channel stride and a common row pitch (a channels-last feature slice qualifies): returns
    (tensor, pitch) without copying when possible."""
    B, N, D = t.shape
    es = t.element_size()
    if t.stride(2) == 1 and t.stride(0) == N * t.stride(1) and t.stride(1) >= D and (t.stride(1) * es) % 16 == 0 \
            and t.data_ptr() % 16 == 0 and t.dtype in _DT:
        return t, t.stride(1)
    t = (t if t.dtype in _DT else t.float()).contiguous()
    return t, D


def cos_kernel(x, y, T=0.2, eps=1e-6, diag_add=0.0, batch_shift=0):
    """CosKernel.__call__ — matcher.py:154-163.  x (B,N,D), y (B,M,D) -> (B,N,M) fp32, on the fp32 MFMA.  16-bit rows are
    widened exactly in the kernel (= the reference's `.float()`, matcher.py:254); x[b] meets y[(b + batch_shift) % B]."""
    _need_gpu(x, y)
    if x.dtype != y.dtype:
        x, y = x.float(), y.float()
    x, xp = _rows(x)
    y, yp = _rows(y)
    B, N, D = x.shape
    M = y.shape[1]
    K = torch.empty((B, N, M), dtype=torch.float32, device=x.device)
    check(_lib.load().roma_cos_kernel(_p(x), _p(y), _p(K), B, N, M, D, _dt(x), xp, yp, int(batch_shift) % B, float(T), float(eps),
                                      float(diag_add), _stream()), "roma_cos_kernel")
    return K


class _PendingSpdCheck:
    """`info` of one spd_solve on its way to pinned host memory; examined by raise_pending() once the copy has landed."""

    def __init__(self, info):
        self.host = torch.empty(info.shape, dtype=info.dtype, pin_memory=True)
        self.host.copy_(info, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def result(self):
        self.event.synchronize()
        return self.host.tolist()


_PENDING = []


def _raise_if_not_spd(bad):
    if any(bad):
        b = next(i for i, v in enumerate(bad) if v)
        raise _lib.RomaHipError(f"spd_solve: matrix {b} of the batch is not positive definite (block step {(bad[b] - 1) // 64}, pivot "
                                f"{(bad[b] - 1) % 64} is <= 0 or NaN) — non-finite or degenerate features?  The reference's "
                                f"torch.linalg.inv raises here too (matcher.py:261).")


_DEFER_DEPTH = [0]


def deferring():
    """True inside `with deferred_spd_checks()`: spd_solve(check="defer") results will be examined when the block ends."""
    return _DEFER_DEPTH[0] > 0


class deferred_spd_checks:
    """RegressionMatcher.match_tensors / match_encoded wrap their pipeline in this: inside, the GP solve's not-SPD check is an
    asynchronous copy (no host sync in the middle of the pipeline); when the block ends normally the queued checks are examined
    (the factorisation finished long before, so the wait is free) and a failure raises there — on the inputs that caused it; when the
    block ends by an exception the queue is dropped, so nothing stale survives into a later, unrelated call.  Outside such a block
    the Decoder asks for check="now".  (Nothing is queued while a hipGraph is being captured.)"""

    def __enter__(self):
        _DEFER_DEPTH[0] += 1
        return self

    def __exit__(self, et, ev, tb):
        _DEFER_DEPTH[0] -= 1
        if _DEFER_DEPTH[0] > 0:
            return False
        todo, _PENDING[:] = list(_PENDING), []
        if et is None and not torch.cuda.is_current_stream_capturing():
            for chk in todo:
                _raise_if_not_spd(chk.result())
        return False


def raise_pending():
    """Examine whatever spd_solve(check="defer") has queued (for callers that defer by hand; no-op while a hipGraph is captured)."""
    if torch.cuda.is_current_stream_capturing():
        return
    todo, _PENDING[:] = list(_PENDING), []
    for chk in todo:
        _raise_if_not_spd(chk.result())


def spd_solve(K, F, check="now", refine=0):
    """Solve K X = F for a batch of SPD matrices by a hand-blocked Cholesky: K (B,n,n) fp32, F (B,n,m) fp32.  Replaces
    `inv(K) @ F` of GP.forward (matcher.py:259-263).

    Works on the augmented matrix A = [K | F].  Per 64-wide block step s ONE launch (roma_chol_step) computes the panel
    R = W_s @ A[rows of the block, columns right of it] = [L_panel^T | Y_s] (the transposed Cholesky panel and the block's share of
    the forward substitution at once), applies the trailing update A[below, right] -= R_left^T @ R to K and to the right-hand
    sides together (tiles on or right of the diagonal only), and factors the NEXT diagonal block in the workgroup that just
    produced it (W_{s+1} = L^-1 of that block).  Round 1 / early round 2 used a diagonal-block kernel + two library GEMMs per block
    for this (75 launches) and two GEMMs per block for the backward substitution (50 launches), which is now one
    roma_chol_subst_step launch per block row: 51 launches of hand-written kernels for n = 1600.

    refine > 0: that many steps of iterative refinement follow — the residual F - K X in fp64 (one fp64 bmm) and the correction
    through the finished factor (forward + back substitution, 2 x 25 launches, no second factorisation).  cond(K_yy) is ~1e4 on real
    features, so a plain fp32 solve is good to ~1e-3 in Z; one step brings Z to 2e-4 of the fp64 solution OF THE SAME K.  Off by
    default: once roma_cos_kernel delivers K entries good to 1e-6 the GP posterior is 2.1e-4 from the fp64 answer without it and
    1.9e-4 with it (tools/gp_error_budget.py) — the K entries, not the solve, carry what error is left — and it costs 1 ms per match.

    A non-positive or NaN pivot (K not SPD: non-finite / degenerate features) is recorded by the kernel, which clamps it
    and carries on; check="now" reads that record back and raises like the reference's torch.linalg.inv does;
    check="defer" queues an asynchronous copy of it and raise_pending() raises later (no host sync inside the pipeline);
    check=None skips it.

    PYTORCH_TUNABLEOP_ENABLED=1: TunableOp's candidate sweep returned hipErrorInvalidValue for the in-place strided
    trailing update of an earlier build (gpurun_out/bench_tune.err, round 1); the one library call left in this routine (the fp64
    residual product of `refine`) therefore always runs with TunableOp switched off (restored afterwards).

    The block size is fixed at 64: roma_chol_step factors the NEXT diagonal block at min(64, n - e) rows whatever the caller's block
    size, so any other value would overrun the W slots (round 2's `nb` argument was never exercised with anything else and is gone)."""
    _need_gpu(K, F)
    assert K.dtype == torch.float32 and F.dtype == torch.float32
    tun = torch.cuda.tunable.is_enabled()
    if tun:
        torch.cuda.tunable.enable(False)
    try:
        X, info = _spd_solve(K, F, refine)
    finally:
        if tun:
            torch.cuda.tunable.enable(True)
    if check == "now":
        _raise_if_not_spd(info.tolist())
    elif check == "defer":
        _PENDING.append(_PendingSpdCheck(info))
    return X


def _spd_solve(K, F, refine=0):
    nb = 64
    B, n, _ = K.shape
    m = F.shape[2]
    lib = _lib.load()
    A = torch.cat((K, F.expand(B, -1, -1)), dim=2)                # (B, n, n+m)
    steps = [(j, min(j + nb, n)) for j in range(0, n, nb)]
    W = torch.empty((B, len(steps), nb, nb), dtype=torch.float32, device=K.device)
    info = torch.zeros((B,), dtype=torch.int32, device=K.device)
    # forward: one launch per block (roma_chol_step: panel r = W_s A[j:e, e:], trailing update, NEXT diagonal block factored in place)
    Rall = torch.empty((B, len(steps), nb, n + m), dtype=torch.float32, device=K.device)
    check(lib.roma_chol_diag_block(A.data_ptr(), A.stride(1), A.stride(0), W.data_ptr(), nb, W.stride(0), steps[0][1], B,
                                   info.data_ptr(), 0, _stream()), "roma_chol_diag_block")
    for s, (j, e) in enumerate(steps):
        last = s + 1 == len(steps)
        check(lib.roma_chol_step(A.data_ptr(), A.stride(1), A.stride(0), n, n + m, j, e - j, W[:, s].data_ptr(), nb, W.stride(0),
                                 Rall[:, s].data_ptr(), n + m, Rall.stride(0), None if last else W[:, s + 1].data_ptr(), nb, W.stride(0),
                                 info.data_ptr(), nb * (s + 1), B, _stream()), "roma_chol_step")
    X = torch.empty((B, n, m), dtype=torch.float32, device=K.device)
    S = len(steps)

    def subst(direction, T, sTb, sTs, ldt, in_panel, out):
        for s in (range(S - 1, -1, -1) if direction < 0 else range(S)):
            check(lib.roma_chol_subst_step(direction, W[:, s].data_ptr(), nb, W.stride(0), Rall.data_ptr(), Rall.stride(0), Rall.stride(1),
                                           n + m, T.data_ptr(), sTb, sTs, ldt, in_panel, out.data_ptr(), m, out.stride(0), n, m, nb, s, B,
                                           _stream()), "roma_chol_subst_step")

    subst(-1, Rall, Rall.stride(0), Rall.stride(1), n + m, 1, X)     # L^T X = Y: one launch per block row, from the bottom up
    for _ in range(int(refine)):
        # iterative refinement: residual in fp64 against the fp32-stored K, correction through the finished factor
        # (forward + back substitution, 2 x S launches; no second factorisation)
        g = (F.double() - torch.bmm(K.double(), X.double())).float().contiguous()
        y = torch.empty_like(g)
        subst(+1, g, g.stride(0), nb * m, m, 0, y)
        d = torch.empty_like(g)
        subst(-1, y, y.stride(0), nb * m, m, 0, d)
        X = X + d
    return X, info


def match_finalize(flow, certainty, cert16, symmetric=True):
    """match() post-processing — matcher.py:656-662, 684-718.  Returns warp (P,H,2W|W,4), certainty (P,H,2W|W)."""
    _need_gpu(flow, certainty, cert16)
    B, _, H, W = flow.shape
    P = B // 2 if symmetric else B
    flow = flow.float().contiguous()
    certainty = certainty.float().contiguous()
    H16 = W16 = 0
    if cert16 is not None:
        cert16 = cert16.float().contiguous()
        H16, W16 = cert16.shape[-2:]
    WW = 2 * W if symmetric else W
    warp = torch.empty((P, H, WW, 4), dtype=torch.float32, device=flow.device)
    cert = torch.empty((P, H, WW), dtype=torch.float32, device=flow.device)
    check(_lib.load().roma_match_finalize(_p(flow), _p(certainty), _p(cert16), _p(warp), _p(cert), P, H, W, H16, W16,
                                          1 if symmetric else 0, _stream()), "roma_match_finalize")
    return warp, cert


def kde(x, std=0.1, half=True, down=None):
    """romatch/utils/kde.py:4-12.  half=True (the reference's default): the coordinates are rounded to fp16 and every term
    goes through the rounding points of the reference's fp16 evaluation (torch.cdist's matmul route in fp16, then fp16
    `** 2`, `/`, `exp`), summed in fp32, result fp16 — the reference's own fp16 noise (up to ~20 % per term) is part of
    what sample()'s `density < 10` cut sees, so it is reproduced rather than removed.  half=False: fp32 throughout."""
    _need_gpu(x)
    if x.dim() != 2 or x.shape[1] != 4:
        raise ValueError("kde expects (N,4) matches")
    xs = (x.half() if half else x).float().contiguous()
    N = xs.shape[0]
    dens = torch.empty((N,), dtype=torch.float32, device=x.device)
    check(_lib.load().roma_kde_density(_p(xs), _p(dens), N, int(down or 1), float(std), 1 if half else 0, _stream()), "roma_kde_density")
    return dens.half() if half else dens


def race_keys(p, thresh=-1.0, seed=0, counter=None, stage=0):
    """Exponential-race keys of sampling without replacement (matcher.py:474-493): key_i = w_i / E_i, w_i = 1 where
    p_i > thresh >= 0 else p_i, E_i ~ Exp(1) from a counter hash of (seed, stage, counter[i] or i).  topk(keys, k) is a draw of k
    items without replacement with probabilities proportional to w.  stage: 0 / 1 = first / second draw of one sample() call."""
    _need_gpu(p, counter)
    p = p.reshape(-1).float().contiguous()
    if counter is not None:
        counter = counter.reshape(-1).long().contiguous()
        assert counter.numel() == p.numel()
    keys = torch.empty_like(p)
    check(_lib.load().roma_race_keys(_p(p), _p(counter), _p(keys), p.numel(), float(thresh), int(seed) & 0xFFFFFFFF,
                                     int(stage) & 0xFFFFFFFF, _stream()), "roma_race_keys")
    return keys


def nn_argmin(q, r):
    """idx[i] = arg min_j |q_i - r_j|^2 for 2-D point sets q (NQ,2), r (NR,2) -> (NQ,) int64; the two arg-min vectors that the
    mutual-nearest-neighbour test of match_keypoints (matcher.py:585-588) needs, without the |q| x |r| cdist matrix."""
    _need_gpu(q, r)
    if q.dim() != 2 or q.shape[1] != 2 or r.dim() != 2 or r.shape[1] != 2:
        raise ValueError("nn_argmin expects (N,2) point sets")
    q, r = q.float().contiguous(), r.float().contiguous()
    idx = torch.empty((q.shape[0],), dtype=torch.int32, device=q.device)
    check(_lib.load().roma_nn_argmin(_p(q), _p(r), _p(idx), q.shape[0], r.shape[0], _stream()), "roma_nn_argmin")
    return idx.long()


def dwconv5x5_bn_relu(x, w25, scale, shift, out=None):
    """relu(BN(depthwise5x5(x))) with BN folded into (scale, shift) — matcher.py:77-103.  x, out: channels-last
    (B,C,H,W) views; w25 (25,C) fp32 tap-major."""
    _need_gpu(x, w25, scale, shift, out)
    B, C, H, W = x.shape
    lx, px, xx = feat_layout(x)
    if lx != ROMA_NHWC:
        raise ValueError("dwconv5x5_bn_relu needs channels-last input")
    if out is None:
        out = nhwc_empty(B, C, H, W, x.dtype, x.device, pitch=px)
    lo, po, o = feat_layout(out)
    if lo != ROMA_NHWC or o is not out:
        raise ValueError("dwconv5x5_bn_relu needs channels-last output")
    check(_lib.load().roma_dwconv5x5_bn_relu(_p(xx), _p(w25), _p(scale), _p(shift), _p(out), B, C, H, W, _dt(xx), px, po, _stream()),
          "roma_dwconv5x5_bn_relu")
    return out


def resample_u8(img, out_size, axis, bounds, coef, ksize):
    """One pass of PIL's 8-bit resampling (see roma_amd/preproc.py): img uint8 (H,W,C) CUDA, axis 1 = horizontal."""
    _need_gpu(img, bounds, coef)
    H, W, C = img.shape
    assert img.dtype == torch.uint8 and img.is_contiguous() and bounds.dtype == torch.int32 and coef.dtype == torch.int32
    assert bounds.shape == (out_size, 2) and coef.shape == (out_size, ksize) and bounds.is_contiguous() and coef.is_contiguous()
    out = torch.empty((H, out_size, C) if axis else (out_size, W, C), dtype=torch.uint8, device=img.device)
    check(_lib.load().roma_resample_u8(_p(img), _p(out), H, W, C, out_size, axis, _p(bounds), _p(coef), ksize, _stream()), "roma_resample_u8")
    return out


def normalize_u8(img, mean, std):
    """uint8 (H,W,3) CUDA -> fp32 (3,H,W): ((v/255) - mean)/std, the reference's operation order (utils.py:176-185, 251-261)."""
    import ctypes
    _need_gpu(img)
    H, W, C = img.shape
    assert C == 3 and img.dtype == torch.uint8 and img.is_contiguous()
    out = torch.empty((3, H, W), dtype=torch.float32, device=img.device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    check(_lib.load().roma_normalize_u8(_p(img), _p(out), H, W, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), _stream()),
          "roma_normalize_u8")
    return out


def _is_nhwc(x):
    return x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)


def bias_relu_(x, bias):
    """In place max(x + bias[c], 0) on a dense (B,C,H,W) map, planar or channels-last — the folded conv+BN+ReLU epilogue of VGG19-BN
    (encoders.py:68-78)."""
    _need_gpu(x, bias)
    B, C, H, W = x.shape
    assert bias.dtype == x.dtype and bias.numel() == C and bias.is_contiguous()
    if _is_nhwc(x):
        check(_lib.load().roma_bias_relu_nhwc(_p(x), _p(bias), B * H * W, C, _dt(x), _stream()), "roma_bias_relu_nhwc")
        return x
    assert x.is_contiguous()
    step = max(1, 65535 // C)
    for b0 in range(0, B, step):
        xb = x[b0:b0 + step]
        check(_lib.load().roma_bias_relu_nchw(_p(xb), _p(bias), xb.shape[0], C, H * W, _dt(x), _stream()), "roma_bias_relu_nchw")
    return x


def bias_relu_pool2_(x, bias):
    """bias_relu_ fused with the following MaxPool2d(2, 2): x (B,C,H,W), planar or channels-last, is updated in place (the pyramid feature captured
    before the pool, encoders.py:68-78) and the pooled (B,C,H/2,W/2) map is returned."""
    _need_gpu(x, bias)
    B, C, H, W = x.shape
    assert bias.dtype == x.dtype and bias.numel() == C and bias.is_contiguous() and H % 2 == 0 and W % 2 == 0
    if _is_nhwc(x):
        out = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
        check(_lib.load().roma_bias_relu_pool2_nhwc(_p(x), _p(bias), _p(out), B, C, H, W, _dt(x), _stream()), "roma_bias_relu_pool2_nhwc")
        return out
    assert x.is_contiguous()
    out = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    step = max(1, 65535 // C)
    for b0 in range(0, B, step):
        xb = x[b0:b0 + step]
        check(_lib.load().roma_bias_relu_pool2_nchw(_p(xb), _p(bias), _p(out[b0:b0 + step]), xb.shape[0], C, H, W, _dt(x), _stream()),
              "roma_bias_relu_pool2_nchw")
    return out


def pointwise_mfma(rows, wt, bias, C, out=None):
    """rows (M, pitch) @ wt[:C,:C]^T + bias for 32 < C <= 160 on the matrix cores (matcher.py:102).  wt (kpad,kpad) [out][in] in
    rows' dtype, bias (kpad) fp32; out may be `rows` itself only for a full in-place update of the same rows."""
    _need_gpu(rows, wt, bias, out)
    M, pitch = rows.shape
    kpad = wt.shape[0]
    assert rows.is_contiguous() and wt.is_contiguous() and wt.dtype == rows.dtype and bias.numel() == kpad and bias.dtype == torch.float32
    if out is None:
        out = torch.empty_like(rows)
    assert out.shape == rows.shape and out.is_contiguous()
    check(_lib.load().roma_pointwise_mfma(_p(rows), _p(wt), _p(bias), _p(out), M, C, kpad, _dt(rows), pitch, pitch, _stream()),
          "roma_pointwise_mfma")
    return out


def refiner_wide_pack(wt_out_in):
    """Re-tile a (D, D) [out][in] 16-bit 1x1-conv weight into the panel-major layout roma_refiner_block_wide streams (host-side
    helper of the library; returns a tensor on wt's device)."""
    w = wt_out_in.detach().to("cpu").contiguous()
    assert w.dim() == 2 and w.shape[0] == w.shape[1] and w.element_size() == 2
    out = torch.empty_like(w)
    check(_lib.load().roma_refiner_wide_pack(w.data_ptr(), out.data_ptr(), w.shape[0]), "roma_refiner_wide_pack")
    return out.to(wt_out_in.device)


def refiner_wide_taps(w25, dtype=torch.float16):
    """(25, D) tap-major depthwise weights -> the 16-bit layout roma_refiner_block_wide stages (host-side helper of the library,
    include/roma_hip.h): [D/32 panels][5 tap rows][4 packets][2 half-packets][6 pair sets][4 channels][2].  The kernel multiplies
    pixel PAIRS (even, odd column) with tap pairs (v_dot2_f32_f16): for a tap row (w0..w4) the even output takes (w0,w1) (w2,w3)
    (w4,0) and the odd one (0,w0) (w1,w2) (w3,w4) against the same three pair words."""
    D = w25.shape[1]
    w = w25.detach().to("cpu").to(dtype).contiguous()
    assert w.shape == (25, D) and w.element_size() == 2
    out = torch.empty((D // 32, 5, 4, 2, 6, 4, 2), dtype=dtype)
    check(_lib.load().roma_refiner_wide_taps(w.data_ptr(), out.data_ptr(), D), "roma_refiner_wide_taps")
    return out.to(w25.device)


def refiner_block_wide(x_nhwc, w25p, scale, shift, wp, bias, out=None):
    """One fused ConvRefiner block (matcher.py:77-103) at D = 576, fp16: x_nhwc (B,h,w,576) contiguous, w25p from refiner_wide_taps,
    scale / shift / bias (576) fp32, wp from refiner_wide_pack.  out must not alias x."""
    _need_gpu(x_nhwc, w25p, scale, shift, wp, bias, out)
    B, H, W, D = x_nhwc.shape
    assert x_nhwc.is_contiguous() and w25p.shape == (D // 32, 5, 4, 2, 6, 4, 2) and w25p.dtype == x_nhwc.dtype and w25p.is_contiguous()
    assert all(t.dtype == torch.float32 and t.numel() == D and t.is_contiguous() for t in (scale, shift, bias))
    if out is None:
        out = torch.empty_like(x_nhwc)
    assert out.shape == x_nhwc.shape and out.is_contiguous() and out.data_ptr() != x_nhwc.data_ptr()
    check(_lib.load().roma_refiner_block_wide(_p(x_nhwc), _p(w25p), _p(scale), _p(shift), _p(wp), _p(bias), _p(out), B, H, W, D, D, D,
                                              _dt(x_nhwc), _stream()), "roma_refiner_block_wide")
    return out


def refiner_block(x_nhwc, w25, scale, shift, wt, bias, C, out=None):
    """One fused ConvRefiner block (depthwise 5x5 + BN + ReLU + 1x1 conv, matcher.py:77-103) for C <= 160, fp16/bf16.
    x_nhwc: (B,h,w,pitch) contiguous; w25 (25,kpad), wt (kpad,kpad) [out][in] in x's dtype; scale/shift/bias (kpad) fp32."""
    _need_gpu(x_nhwc, w25, scale, shift, wt, bias, out)
    B, h, w, pitch = x_nhwc.shape
    kpad = wt.shape[0]
    assert x_nhwc.is_contiguous() and wt.is_contiguous() and w25.is_contiguous() and wt.dtype == x_nhwc.dtype == w25.dtype
    assert w25.shape == (25, kpad) and wt.shape == (kpad, kpad) and scale.numel() == shift.numel() == bias.numel() == kpad
    if out is None:
        out = torch.empty_like(x_nhwc)
    assert out.is_contiguous() and out.shape == x_nhwc.shape and out.data_ptr() != x_nhwc.data_ptr()
    check(_lib.load().roma_refiner_block(_p(x_nhwc), _p(w25), _p(scale), _p(shift), _p(wt), _p(bias), _p(out), B, C, h, w, kpad,
                                         _dt(x_nhwc), pitch, pitch, _stream()), "roma_refiner_block")
    return out


def refiner_head(x_nhwc, wo, bo, flow, certainty, sx, sy, want_delta=False):
    """out_conv (D -> 3, fp32) of the last refiner block fused with the flow / certainty update (matcher.py:141, 397-402).
    x_nhwc: (B,h,w,pitch) channels-last activation; wo (C,3) fp32; bo (3) fp32; flow (B,2,h,w) fp32 is updated IN PLACE;
    certainty (B,1,h,w) fp32 or None.  Returns (flow, new_certainty[, delta (B,3,h,w)])."""
    _need_gpu(x_nhwc, wo, bo, flow, certainty)
    B, h, w, pitch = x_nhwc.shape
    C = wo.shape[0]
    assert x_nhwc.is_contiguous() and flow.is_contiguous() and flow.dtype == torch.float32 and wo.is_contiguous()
    cert_in = None if certainty is None else certainty.float().contiguous()
    cert = torch.empty((B, 1, h, w), dtype=torch.float32, device=flow.device)
    delta = torch.empty((B, 3, h, w), dtype=torch.float32, device=flow.device) if want_delta else None
    check(_lib.load().roma_refiner_head(_p(x_nhwc), _p(wo), _p(bo), _p(flow), _p(cert_in), _p(cert), _p(delta), B, h, w, C, pitch,
                                        _dt(x_nhwc), float(sx), float(sy), _stream()), "roma_refiner_head")
    return (flow, cert, delta) if want_delta else (flow, cert)


def pointwise_small(x_rows, wt, bias, out=None):
    """1x1 convolution on narrow channels-last rows: x_rows (M, C) view with row stride >= C, wt (C, C) fp32 (in, out),
    bias (C) fp32 — matcher.py:102 for the D = 24 refiner.  Returns (M, C) rows of x's dtype."""
    _need_gpu(x_rows, wt, bias, out)
    M, C = x_rows.shape
    assert x_rows.stride(1) == 1
    if out is None:
        out = torch.empty((M, C), dtype=x_rows.dtype, device=x_rows.device)
    check(_lib.load().roma_pointwise_small(_p(x_rows), _p(wt), _p(bias), _p(out), M, C, _dt(x_rows), x_rows.stride(0), out.stride(0),
                                           _stream()), "roma_pointwise_small")
    return out


def attention(qkv, n_valid=None, out=None):
    """softmax(q k^T / sqrt(d)) v for every head — attention.py:48-60 — straight from the qkv projection's output.
    qkv: (B, N, 3, H, 64) fp16 / bf16, contiguous; keys / values are the first n_valid tokens (row padding beyond them queries but is
    never attended to).  Returns (B, N, H*64), the input layout of the output projection."""
    _need_gpu(qkv, out)
    B, N, three, H, d = qkv.shape
    assert three == 3 and qkv.is_contiguous()
    if out is None:
        out = torch.empty((B, N, H * d), dtype=qkv.dtype, device=qkv.device)
    assert out.is_contiguous() and out.shape == (B, N, H * d) and out.dtype == qkv.dtype
    es = qkv.element_size()
    sb, sn, sh = qkv.stride(0), qkv.stride(1), qkv.stride(3)
    base = qkv.data_ptr()
    step = qkv.stride(2) * es
    check(_lib.load().roma_attention_fwd(base, base + step, base + 2 * step, out.data_ptr(), B, H, N, N if n_valid is None else int(n_valid), d,
                                         sb, sn, sh, sb, sn, sh, sb, sn, sh, out.stride(0), out.stride(1), d, float(d) ** -0.5, _dt(qkv),
                                         _stream()), "roma_attention_fwd")
    return out


def add_layernorm(x, y, ln_weight, ln_bias, eps, out_dtype, ls=None):
    """x <- x + ls*y (in place, x's dtype; y None: no add); returns LayerNorm(x)*w+b cast to out_dtype (ln_weight None: just
    the cast).  x (..., C) contiguous rows; the seam between two transformer half-blocks (transformer/layers/block.py:87-107)."""
    _need_gpu(x, y, ln_weight, ln_bias, ls)
    C = x.shape[-1]
    assert x.is_contiguous() and (y is None or (y.is_contiguous() and y.shape == x.shape))
    rows = x.numel() // C
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    for t in (ln_weight, ln_bias, ls):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == C)
    check(_lib.load().roma_add_layernorm(_p(x), _dt(x), C, _p(y), _dt(y) if y is not None else 0, C, _p(ls), _p(ln_weight), _p(ln_bias),
                                         _p(out), _DT[out_dtype], C, rows, C, float(eps), _stream()), "roma_add_layernorm")
    return out


def tiny_corr_posembed(f0, f1, exact=False):
    """TinyRoMa.corr_volume + pos_embed fused — tiny.py:241-254,178-203.  f0 (B,C,H0,W0), f1 (B,C,H1,W1) -> (B,2,H0,W0)."""
    _need_gpu(f0, f1)
    B, C, H0, W0 = f0.shape
    _, _, H1, W1 = f1.shape
    a = f0.float().permute(0, 2, 3, 1).contiguous()
    b = f1.float().permute(0, 2, 3, 1).contiguous()
    out = torch.empty((B, 2, H0, W0), dtype=torch.float32, device=f0.device)
    # MFMA-bound: the "bytes" slot of the timer carries the algorithmic FLOPs 2 * N0 * N1 * C per item
    TIMER.wrap("tiny_corr", 2.0 * B * H0 * W0 * H1 * W1 * C, f"C{C}_{H0}x{W0}",
               lambda: check(_lib.load().roma_tiny_corr_posembed(_p(a), _p(b), _p(out), B, C, H0, W0, H1, W1, 1 if exact else 0, _stream()),
                             "roma_tiny_corr_posembed"))
    return out
