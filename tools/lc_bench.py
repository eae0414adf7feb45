#!/usr/bin/env python3
"""local_correlation: every kernel variant at the five 560->864 call shapes, coherent flow, fp16 channels-last.
   python tools/lc_bench.py [--pairs 1 16] [--variants auto tile8x8 rows8] [--flow coherent]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
from tests.golden import recipes as R
SH = [("L16", 512, 40, 7), ("L8", 512, 70, 3), ("L4", 256, 140, 2), ("U8", 512, 108, 3), ("U4", 256, 216, 2)]
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, nargs="+", default=[1, 16])
ap.add_argument("--variants", nargs="+", default=["auto", "tile8x4", "tile8x8", "rows8"])
ap.add_argument("--flow", default="coherent")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]


def timeit(fn, iters, warm=5):
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


for pairs in a.pairs:
    B = 2 * pairs
    data = {}
    for name, C, h, r in SH:
        f0 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        f1 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        mk = R.coherent_flow if a.flow == "coherent" else R.adversarial_flow
        flow = torch.from_numpy(mk("bench", B, h, h)).cuda()
        out = ops.nhwc_empty(B, (2 * r + 1) ** 2, h, h, dt, "cuda")
        nbytes = 2 * B * C * h * h * 2 + B * 2 * h * h * 4 + B * (2 * r + 1) ** 2 * h * h * 2
        data[name] = (f0, f1, flow, out, r, nbytes)
    print(f"--- {pairs} pair(s) per launch, {a.flow} flow, {a.dtype}: us per launch (%% of 8 TB/s on algorithmic bytes)")
    for v in a.variants:
        tot_t = tot_b = 0
        line = f"{v:8s}"
        for name, C, h, r in SH:
            f0, f1, flow, out, r, nb = data[name]
            t = timeit(lambda: ops.local_correlation(f0, f1, r, flow=flow, out=out, variant=v), a.iters)
            tot_t += t; tot_b += nb
            line += f"  {name} {t*1e6:7.1f} ({nb/t/8e10:4.1f}%)"
        no16_t = tot_t - 0
        print(line + f"  | ALL5 {tot_t*1e6:8.1f} us = {tot_b/tot_t/8e10:4.1f}%", flush=True)
