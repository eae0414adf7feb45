#!/usr/bin/env python3
"""depthwise 5x5 + BN + ReLU alone, one shape, for rocprofv3 runs: python tools/dw_micro.py 576 216 --iters 5"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("D", type=int)
ap.add_argument("h", type=int)
ap.add_argument("--B", type=int, default=2)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
x = torch.randn(a.B, a.D, a.h, a.h, device="cuda").half().contiguous(memory_format=torch.channels_last)
w = torch.randn(25, a.D, device="cuda")
sc, sh = torch.rand(a.D, device="cuda") + 0.5, torch.randn(a.D, device="cuda")
y = torch.empty_like(x)
for _ in range(a.iters):
    ops.dwconv5x5_bn_relu(x, w, sc, sh, out=y)
torch.cuda.synchronize()
print("done", a.D, a.h)
