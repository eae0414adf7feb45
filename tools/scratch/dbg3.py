import sys, os, math
import torch
sys.path.insert(0, "/root/repo")
from roma_amd import ops
B, C, h, w, r = 1, 8, 6, 6, 1
for dt in (torch.float32, torch.float16):
    f0 = torch.ones(B, C, h, w).to(dt)
    f1 = torch.zeros(B, C, h, w)
    f1[0, 0] = torch.arange(36).reshape(6, 6).float()
    f1 = f1.to(dt)
    out = ops.local_correlation(f0.cuda().contiguous(memory_format=torch.channels_last), f1.cuda().contiguous(memory_format=torch.channels_last), r, flow=None)
    print(dt, "spatial: out[0,:,2,3]*sqrt8 =", (out[0, :, 2, 3].float().cpu() * math.sqrt(8)).tolist(), "expect", [8,9,10,14,15,16,20,21,22])
    for j in range(8):
        f0 = torch.zeros(B, C, h, w); f0[0, j] = 1
        f1 = torch.arange(1, 9).float().reshape(1, 8, 1, 1).expand(B, C, h, w).contiguous()
        out = ops.local_correlation(f0.to(dt).cuda().contiguous(memory_format=torch.channels_last), f1.to(dt).cuda().contiguous(memory_format=torch.channels_last), r, flow=None)
        print(dt, "chan", j, "centre*sqrt8 =", float(out[0, 4, 2, 3]) * math.sqrt(8))
