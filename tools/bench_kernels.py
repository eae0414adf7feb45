#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (HIP events on the launch stream).  Prints one line per case:
algorithmic bytes, time, GB/s, fraction of the 8 TB/s HBM spec."""
import argparse
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops  # noqa: E402
from tests.golden import recipes as R  # noqa: E402

LC_SHAPES = [("L16", 512, 40, 7), ("L8", 512, 70, 3), ("L4", 256, 140, 2), ("U8", 512, 108, 3), ("U4", 256, 216, 2)]
PEAK = 8.0e12


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--flow", default="coherent")
    ap.add_argument("--variant", default="auto", help="local_corr kernel: auto | tile8x4 | tile8x8 | rows8")
    args = ap.parse_args()
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    es = 4 if dt == torch.float32 else 2
    B = 2 * args.pairs
    tot_b = tot_t = 0
    for name, C, h, r in LC_SHAPES:
        K = (2 * r + 1) ** 2
        f0 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        f1 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        if args.flow == "coherent":
            flow = torch.from_numpy(R.coherent_flow("bench", B, h, h)).cuda()
        else:
            flow = torch.from_numpy(R.adversarial_flow("bench", B, h, h)).cuda()
        out = ops.nhwc_empty(B, K, h, h, dt, "cuda")
        t = timeit(lambda: ops.local_correlation(f0, f1, r, flow=flow, out=out, variant=args.variant))
        nbytes = 2 * B * C * h * h * es + B * 2 * h * h * 4 + B * K * h * h * es
        tot_b += nbytes
        tot_t += t
        print(f"local_corr {name:4s} C={C} h={h} r={r} B={B} {args.dtype} {args.flow}: {t*1e6:8.1f} us  {nbytes/1e6:8.2f} MB  "
              f"{nbytes/t/1e9:8.1f} GB/s  {nbytes/t/PEAK*100:5.1f}% of 8TB/s", flush=True)
    print(f"local_corr ALL5: {tot_t*1e6:.1f} us  {tot_b/1e6:.2f} MB  {tot_b/tot_t/1e9:.1f} GB/s  {tot_b/tot_t/PEAK*100:.1f}%")

    # depthwise conv at the refiner shapes
    for D, h in [(1384, 40), (1144, 70), (576, 140), (144, 280), (24, 560), (1144, 108), (576, 216), (144, 432), (24, 864)]:
        x = torch.randn(B, D, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        w = torch.randn(25, D, device="cuda")
        sc, sh = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
        y = torch.empty_like(x)
        t = timeit(lambda: ops.dwconv5x5_bn_relu(x, w, sc, sh, out=y), iters=20)
        nb = 2 * B * D * h * h * es
        print(f"dwconv D={D} h={h}: {t*1e6:8.1f} us {nb/1e6:8.2f} MB {nb/t/1e9:8.1f} GB/s", flush=True)

    # cos kernel, cls refine, kde
    x = torch.randn(B, 1600, 512, device="cuda")
    y = torch.randn(B, 1600, 512, device="cuda")
    t = timeit(lambda: ops.cos_kernel(x, y), iters=20)
    print(f"cos_kernel B={B} 1600x1600x512 fp32-MFMA: {t*1e6:.1f} us  {2*B*1600*1600*512/t/1e12:.1f} TFLOP/s")
    xn = torch.nn.functional.normalize(torch.randn(2, 1600, 64, device="cuda"), dim=-1)
    K = torch.exp((xn @ xn.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(1600, device="cuda")
    F = torch.randn(1, 1600, 512, device="cuda")
    t = timeit(lambda: ops.spd_solve(K, F, check=None), 10, 2)
    print(f"spd_solve B=2 n=1600 m=512 (1 + 25 forward-step + 25 back-step launches): {t*1e3:.3f} ms", flush=True)
    A = K[:, :64, :64].contiguous()
    Wb = torch.empty(2, 64, 64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    from roma_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    t = timeit(lambda: lib.roma_chol_diag_block(A.data_ptr(), 64, 64 * 64, Wb.data_ptr(), 64, 64 * 64, 64, 2, info.data_ptr(), 0, st), 50)
    print(f"chol_diag_block 64x64 B=2: {t*1e6:.1f} us per launch (back to back: A is re-factored, timing only)", flush=True)
    rows = torch.randn(B, 1600, 4097, device="cuda").to(dt)
    t = timeit(lambda: ops.cls_rows_to_flow(rows, B, 40, 40), iters=20)
    print(f"cls_rows_to_flow: {t*1e6:.1f} us  {rows.numel()*es/t/1e9:.1f} GB/s")
    # warp (grid_sample) at the refiner shapes, finalize, narrow pointwise, fused head, tiny corr+soft-argmax
    for C, h in [(512, 40), (512, 70), (256, 140), (64, 280), (9, 560), (512, 108), (256, 216), (64, 432), (9, 864)]:
        pitch = 24 if C == 9 else C          # the scale-1 refiner warps 9 channels inside its 24-channel pixel
        y = torch.randn(B, h, h, pitch, device="cuda").to(dt)[..., :C].permute(0, 3, 1, 2)
        flow = torch.from_numpy(R.coherent_flow("bench", B, h, h)).cuda()
        o = ops.nhwc_empty(B, C, h, h, dt, "cuda", pitch=pitch)
        t = timeit(lambda: ops.warp_bilinear(y, flow, out=o), iters=20)
        nb = 2 * B * C * h * h * es + B * 2 * h * h * 4
        print(f"warp_bilinear C={C} h={h}: {t*1e6:8.1f} us {nb/1e6:8.2f} MB {nb/t/1e9:8.1f} GB/s", flush=True)
    fl = torch.rand(B, 2, 864, 864, device="cuda") * 2 - 1
    ce = torch.randn(B, 1, 864, 864, device="cuda")
    c16 = torch.randn(B, 1, 40, 40, device="cuda")
    t = timeit(lambda: ops.match_finalize(fl, ce, c16), iters=20)
    nb = B * 864 * 864 * (3 + 5) * 4
    print(f"match_finalize 864: {t*1e6:.1f} us  {nb/t/1e9:.1f} GB/s")
    for h in (560, 864):
        x = torch.randn(B * h * h, 24, device="cuda").to(dt)
        w = torch.randn(24, 24, device="cuda"); bb = torch.randn(24, device="cuda")
        t = timeit(lambda: ops.pointwise_small(x, w, bb), iters=20)
        print(f"pointwise_small D=24 h={h}: {t*1e6:.1f} us  {2*x.numel()*es/t/1e9:.1f} GB/s")
        xx = x.view(B, h, h, 24); wo = torch.randn(24, 3, device="cuda"); bo = torch.randn(3, device="cuda")
        f2 = torch.rand(B, 2, h, h, device="cuda"); c2 = torch.randn(B, 1, h, h, device="cuda")
        t = timeit(lambda: ops.refiner_head(xx, wo, bo, f2, c2, 1e-3, 1e-3), iters=20)
        print(f"refiner_head D=24 h={h}: {t*1e6:.1f} us  {(x.numel()*es + B*h*h*16)/t/1e9:.1f} GB/s")
    f0 = torch.randn(16, 64, 60, 80, device="cuda"); f1 = torch.randn(16, 64, 60, 80, device="cuda")
    for exact in (False, True):
        t = timeit(lambda: ops.tiny_corr_posembed(f0, f1, exact=exact), iters=5, warm=2)
        print(f"tiny_corr_posembed B=16 60x80 C=64 exact={exact}: {t*1e3:.3f} ms  {2*16*4800*4800*64/t/1e12:.1f} TFLOP/s (fp32 MFMA), {16/t:.0f} pairs/s")
    if dt != torch.float32:
        import torch.nn.functional as F
        for Ba in (B, 16):
            qkv = torch.randn(Ba, 1664, 3, 16, 64, device="cuda").to(dt)
            o = torch.empty(Ba, 1664, 1024, device="cuda", dtype=dt)
            t = timeit(lambda: ops.attention(qkv, 1601, out=o), iters=20)
            pp = qkv.permute(2, 0, 3, 1, 4)
            t2 = timeit(lambda: F.scaled_dot_product_attention(pp[0], pp[1][:, :, :1601], pp[2][:, :, :1601]), iters=20)
            fl = 4 * Ba * 16 * 1664 * 1601 * 64
            print(f"attention B={Ba} 16 heads x 64, 1664 queries x 1601 keys: {t*1e6:.1f} us  {fl/t/1e12:.0f} TFLOP/s   (torch SDPA: {t2*1e6:.1f} us)", flush=True)
    pts = torch.rand(40000, 4, device="cuda") * 2 - 1
    t = timeit(lambda: ops.kde(pts, half=True), iters=5, warm=1)
    print(f"kde N=40000: {t*1e3:.3f} ms  {40000*40000/t/1e12:.2f} T pair-exp/s")


if __name__ == "__main__":
    main()
