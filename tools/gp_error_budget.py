#!/usr/bin/env python3
"""Where does the fp32 GP's error against an fp64 evaluation come from?  (sacre_coeur features, bench weights)"""
import torch, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from PIL import Image
from roma_amd import ops
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights
from roma_amd.matcher import preprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = build_roma((560, 560), upsample_preds=True, amp_dtype=torch.float32)
load_synthetic_weights(m, seed=0); m = m.cuda().eval()
ims = [Image.open(os.path.join(ROOT, "tests/golden/assets", f"sacre_coeur_{n}.jpg")).convert("RGB") for n in "AB"]
X = torch.cat([preprocess(im, (560, 560))[None] for im in ims]).cuda()
f16 = m.encoder(X)[16]
x = m.decoder.project("16", f16, torch.float32)
xs = x.permute(0, 2, 3, 1).reshape(2, 1600, -1).float().contiguous()
ys = torch.cat((xs[1:], xs[:1])).contiguous()
gp = m.decoder.gps["16"]
F = gp.basis(1, 40, 40, xs.device).expand(2, -1, -1).contiguous()
T, sig = gp.K.T, gp.sigma_noise
def cosk64(u, v):
    u, v = u.double(), v.double()
    g = torch.einsum("bnd,bmd->bnm", u, v) / (u.norm(dim=-1)[..., None] * v.norm(dim=-1)[:, None] + 1e-6)
    return ((g - 1.0) / T).exp()
I = torch.eye(1600, device=xs.device, dtype=torch.float64)
Kyy64, Kxy64 = cosk64(ys, ys) + sig * I, cosk64(xs, ys)
Z64 = torch.cholesky_solve(F.double(), torch.linalg.cholesky(Kyy64))
mu64 = Kxy64 @ Z64
print("cond(Kyy)", float(torch.linalg.cond(Kyy64[0])), "|Z| max", float(Z64.abs().max()), "|mu| max", float(mu64.abs().max()))
def err(mu): return float((mu.double() - mu64).abs().max())
Kyy32, Kxy32 = ops.cos_kernel(ys, ys, T=T, diag_add=sig), ops.cos_kernel(xs, ys, T=T)
print("K entry rel err: Kyy", float(((Kyy32.double() - Kyy64).abs() / Kyy64).max()), "Kxy", float(((Kxy32.double() - Kxy64).abs() / Kxy64).max()))
ZA = ops.spd_solve(Kyy32, F, refine=0)
print("A  fp32-MFMA K, fp32 solve, fp32 bmm :", err(Kxy32 @ ZA))
ZB = ops.spd_solve(Kyy64.float(), F, refine=0)
print("F  as A but the product accumulated in fp64 (operands fp32)      :", err((Kxy32.double() @ ZA.double()).float()))
print("B  K rounded from fp64, fp32 solve   :", err(Kxy64.float() @ ZB), " (Z err", float((ZB.double() - Z64).abs().max()), ")")
print("B' as B but Kxy fp32-MFMA            :", err(Kxy32 @ ZB))
print("B2 Z exact(fp64->fp32), Kxy fp32-MFMA:", err(Kxy32 @ Z64.float()), "; Kxy rounded from fp64:", err(Kxy64.float() @ Z64.float()))
# C: iterative refinement with an fp64 residual against the fp64-rounded-to-fp32 matrix
K32 = Kyy64.float()
r = (F.double() - K32.double() @ ZB.double()).float()
ZC = ZB + ops.spd_solve(K32, r, refine=0)
print("C  B + 1 refinement step (residual in fp64 against the fp32-stored K):", err(Kxy64.float() @ ZC), " Z err", float((ZC.double() - Z64).abs().max()))
r = (F.double() - Kyy64 @ ZB.double()).float()
ZD = ZB.double() + ops.spd_solve(K32, r, refine=0).double()
print("D  refinement against the fp64 K, Z kept fp64, mu = Kxy64 @ Z:", float((Kxy64 @ ZD - mu64).abs().max()))
ZG = ops.spd_solve(Kyy32, F, refine=1)
print("G  product path: fp32-MFMA K, fp32 solve + 1 refinement step through the finished factor, fp64-accumulated product:",
      err((Kxy32.double() @ ZG.double()).float()), " Z err", float((ZG.double() - Z64).abs().max()))
ZH = ops.spd_solve(Kyy32, F, refine=2)
print("H  as G with 2 refinement steps:", err((Kxy32.double() @ ZH.double()).float()), " Z err", float((ZH.double() - Z64).abs().max()))
for rf in (0, 1):
    ZI = ops.spd_solve(Kyy64.float(), F, refine=rf)
    print(f"I{rf} K_yy, K_xy rounded from fp64 (entry error 6e-8), fp32 solve + {rf} refinement, fp64-accumulated product:",
          err((Kxy64.float().double() @ ZI.double()).float()), " Z err", float((ZI.double() - Z64).abs().max()))
    print(f"J{rf} as I{rf} but K_xy from the fp32-MFMA kernel:", err((Kxy32.double() @ ZI.double()).float()))
# torch's own fp32 paths for comparison
Kt = torch.exp((torch.einsum("bnd,bmd->bnm", ys, ys) / (ys.norm(dim=-1)[..., None] * ys.norm(dim=-1)[:, None] + 1e-6) - 1) / T) + sig * I.float()
Kxt = torch.exp((torch.einsum("bnd,bmd->bnm", xs, ys) / (xs.norm(dim=-1)[..., None] * ys.norm(dim=-1)[:, None] + 1e-6) - 1) / T)
print("E  torch fp32 einsum K + inv (the reference's literal method, on GPU):", err(Kxt @ (torch.linalg.inv(Kt) @ F)))
print("   torch-fp32 K entry rel err:", float(((Kt.double() - Kyy64).abs() / Kyy64).max()))
