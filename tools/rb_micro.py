import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
C, kpad, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(2, h, h, C, device="cuda").half()
w25 = torch.zeros(25, kpad, device="cuda").half(); w25[:, :C] = torch.randn(25, C, device="cuda").half() * 0.2
wt = torch.zeros(kpad, kpad, device="cuda").half(); wt[:C, :C] = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
sc = torch.ones(kpad, device="cuda"); sh = torch.zeros(kpad, device="cuda"); b = torch.zeros(kpad, device="cuda")
out = torch.empty_like(x)
for _ in range(5): ops.refiner_block(x, w25, sc, sh, wt, b, C, out=out)
torch.cuda.synchronize(); print("done")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ops.refiner_block(x, w25, sc, sh, wt, b, C, out=out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e-3
print(f"refiner_block C={C} h={h}: {t*1e6:.1f} us  {2 * x.numel() * 2 / t / 1e12:.2f} TB/s in+out")
