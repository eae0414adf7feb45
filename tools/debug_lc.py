import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
from oracle import roma_oracle as O
from tests.golden import recipes as R
torch.manual_seed(0)
for dt in (torch.float32, torch.float16, torch.bfloat16):
    for (B, C, h, w, r) in [(1, 8, 6, 6, 1), (1, 32, 12, 10, 7), (1, 64, 12, 10, 2), (1, 128, 9, 9, 3), (2, 512, 40, 40, 7), (2, 512, 70, 70, 3)]:
        for kind in ("coh", "adv", "far"):
            f0 = torch.randn(B, C, h, w).to(dt); f1 = torch.randn(B, C, h, w).to(dt)
            flow = torch.from_numpy(R.coherent_flow("d", B, h, w) if kind == "coh" else R.adversarial_flow("d", B, h, w))
            if kind == "far":
                flow = flow * 0 + 2.5
            ref = O.local_correlation(f0.float(), f1.float(), r, flow=flow)
            out = ops.local_correlation(f0.cuda().contiguous(memory_format=torch.channels_last), f1.cuda().contiguous(memory_format=torch.channels_last), r, flow=flow.cuda())
            torch.cuda.synchronize()
            err = (out.float().cpu() - ref).abs()
            print(dt, (B, C, h, w, r), kind, "maxerr %.3e" % err.max().item(), "ref max %.2f" % ref.abs().max().item(), "bad frac %.3f" % (err > 1e-2).float().mean().item(), flush=True)
