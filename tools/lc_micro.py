#!/usr/bin/env python3
"""local_correlation alone, one shape, for rocprofv3 runs: python tools/lc_micro.py U4 --pairs 8 --iters 10"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
from tests.golden import recipes as R
SH = {"L16": (512, 40, 7), "L8": (512, 70, 3), "L4": (256, 140, 2), "U8": (512, 108, 3), "U4": (256, 216, 2)}
ap = argparse.ArgumentParser()
ap.add_argument("shape")
ap.add_argument("--pairs", type=int, default=1)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--flow", default="coherent")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--variant", default="auto", choices=["auto", "tile8x4", "tile8x8", "rows8"])
a = ap.parse_args()
C, h, r = SH[a.shape]
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
B = 2 * a.pairs
f0 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
f1 = torch.randn(B, C, h, h, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
flow = torch.from_numpy(R.coherent_flow("bench", B, h, h) if a.flow == "coherent" else R.adversarial_flow("bench", B, h, h)).cuda()
out = ops.nhwc_empty(B, (2 * r + 1) ** 2, h, h, dt, "cuda")
for _ in range(a.iters):
    ops.local_correlation(f0, f1, r, flow=flow, out=out, variant=a.variant)
torch.cuda.synchronize()
print("done", a.shape, B)
