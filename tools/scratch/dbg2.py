import sys, os
import torch
sys.path.insert(0, "/root/repo")
import roma_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = sys.argv[1]
from roma_amd import ops
from oracle import roma_oracle as O
from tests.golden import recipes as R
torch.manual_seed(0)
B, C, h, w, r = 1, 8, 6, 6, 1
dt = torch.float16
f0 = torch.randn(B, C, h, w).to(dt); f1 = torch.randn(B, C, h, w).to(dt)
ref = O.local_correlation(f0.float(), f1.float(), r, flow=None)
out = ops.local_correlation(f0.cuda().contiguous(memory_format=torch.channels_last), f1.cuda().contiguous(memory_format=torch.channels_last), r, flow=None)
print("ident flow maxerr", (out.float().cpu() - ref).abs().max().item())
print("ref[0,:,2,2]", ref[0, :, 2, 2])
print("out[0,:,2,2]", out[0, :, 2, 2].float().cpu())
# NCHW path (scalar staging)
out2 = ops.local_correlation(f0.cuda(), f1.cuda(), r, flow=None)
print("nchw maxerr", (out2.float().cpu() - ref).abs().max().item())
