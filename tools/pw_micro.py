# stand-alone timing of roma_pointwise_mfma (the D = 144 1x1 of the scale-2 refiner): python tools/pw_micro.py [h ...]
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
C, kpad = 144, 160
for h in [int(a) for a in sys.argv[1:]] or [280, 432]:
    M = 2 * h * h
    x = torch.randn(M, C, device="cuda").half()
    wt = torch.zeros(kpad, kpad, device="cuda").half(); wt[:C, :C] = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
    b = torch.zeros(kpad, device="cuda")
    out = torch.empty_like(x)
    for _ in range(5): ops.pointwise_mfma(x, wt, b, C, out=out)
    ref = (x.float() @ wt[:C, :C].float().t()).half()
    err = float((out.float() - ref.float()).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.pointwise_mfma(x, wt, b, C, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e-3
    print(f"pointwise_mfma C={C} h={h}: {t*1e6:.1f} us  {2 * x.numel() * 2 / t / 1e12:.2f} TB/s in+out  max err vs torch {err:.3g}")
