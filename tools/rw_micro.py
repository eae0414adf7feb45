#!/usr/bin/env python3
"""The fused D = 576 refiner block against the two kernels it replaces: python tools/rw_micro.py [h ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
D = 576
for h in [int(a) for a in sys.argv[1:]] or [140, 216]:
    B = 2
    x = torch.randn(B, h, h, D, device="cuda").half()
    w25 = torch.randn(25, D, device="cuda") * 0.2
    sc, sh, bias = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.1, torch.randn(D, device="cuda") * 0.1
    wt = (torch.randn(D, D, device="cuda") / D ** 0.5).half()
    wp = ops.refiner_wide_pack(wt)
    w25p = ops.refiner_wide_taps(w25)
    out = torch.empty_like(x)
    wtt = wt.t().contiguous()
    bh = bias.half()

    def fused():
        ops.refiner_block_wide(x, w25p, sc, sh, wp, bias, out=out)

    def split():
        t = ops.dwconv5x5_bn_relu(x.permute(0, 3, 1, 2), w25, sc, sh)
        torch.addmm(bh, t.permute(0, 2, 3, 1).reshape(-1, D), wtt)

    for name, fn in (("fused", fused), ("dwconv+addmm", split)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20 * 1e3
        fl = 2.0 * B * h * h * D * D
        print(f"h={h} {name:14s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s (1x1 only)  {2 * x.numel() * 2 / t / 1e6:6.2f} TB/s in+out", flush=True)
