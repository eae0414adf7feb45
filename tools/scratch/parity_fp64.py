import sys, time, torch
sys.path.insert(0, "/root/repo")
from PIL import Image
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights
from roma_amd.matcher import preprocess
from oracle import roma_oracle as O
torch.set_grad_enabled(False); torch.set_num_threads(16)
m = build_roma((560, 560), amp_dtype=torch.float32); load_synthetic_weights(m)
def mk(dt):
    o = O.roma_model((560, 560), (864, 864)); o.load_state_dict(m.state_dict()); o.encoder.dinov2_vitl14[0].load_state_dict(m.encoder.dinov2_vitl14[0].state_dict()); o.encoder.dinov2_vitl14[0].eval()
    if dt == torch.float64:
        o = o.double(); o.encoder.dinov2_vitl14[0] = o.encoder.dinov2_vitl14[0].double()
    return o
ims = [Image.open(f"/root/repo/tests/golden/assets/sacre_coeur_{n}.jpg").convert("RGB") for n in "AB"]
X = torch.cat([preprocess(im, (560, 560))[None] for im in ims])
def scale16(o, X):
    """certainty logit + class logits of the coarse stage (before any refiner), as the oracle's Decoder computes them"""
    pyr = o.encoder(X)
    f = pyr[16]; g = torch.cat((f.chunk(2)[1], f.chunk(2)[0]))
    dt = X.dtype
    a, c = o.decoder.proj["16"](f.to(dt)), o.decoder.proj["16"](g.to(dt))
    gp = o.decoder.gps["16"]
    # GP.forward with dtype preserved
    b, ch, h, w = a.shape
    import math
    fb = torch.cos(8 * math.pi * gp.pos_conv(O.pixel_grid(b, h, w).to(dt)))
    xs, ys, fs = a.flatten(2).transpose(1, 2), c.flatten(2).transpose(1, 2), fb.flatten(2).transpose(1, 2)
    Kyy, Kxy = O.cos_kernel(ys, ys, gp.T), O.cos_kernel(xs, ys, gp.T)
    mu = Kxy @ (O._inv_single_thread(Kyy + gp.sigma_noise * torch.eye(h * w, dtype=dt)[None]) @ fs)
    mu = mu.transpose(1, 2).reshape(b, -1, h, w)
    cls, cert = o.decoder.embedding_decoder(mu, a)
    return mu, cls, cert
t = time.time(); mu32, cls32, cert32 = scale16(mk(torch.float32), X); print("cpu32 %.1fs" % (time.time() - t), flush=True)
t = time.time(); mu64, cls64, cert64 = scale16(mk(torch.float64), X.double()); print("cpu64 %.1fs" % (time.time() - t), flush=True)
mg = m.cuda().eval()
pyr = mg.encoder(X.cuda()); dec = mg.decoder
x = dec.project("16", pyr[16], torch.float32); y = torch.cat((x[1:], x[:1]))
xs = x.permute(0, 2, 3, 1).reshape(2, 1600, -1); ys = y.permute(0, 2, 3, 1).reshape(2, 1600, -1)
mu = dec.gps["16"].posterior_rows(xs.float().contiguous(), ys.float().contiguous(), 40, 40)
rows = dec.embedding_decoder.forward_rows(torch.cat((mu, xs), dim=2))
certg = rows[:, :, -1].reshape(2, 1, 40, 40).cpu(); mug = mu.transpose(1, 2).reshape(2, -1, 40, 40).cpu()
print("GP posterior mu:   |cpu32-fp64| %.2e   |gpu32-fp64| %.2e   |gpu32-cpu32| %.2e   (|mu| max %.2f)" % ((mu32.double() - mu64).abs().max(), (mug.double() - mu64).abs().max(), (mug - mu32).abs().max(), mu64.abs().max()))
print("certainty logit16: |cpu32-fp64| %.2e   |gpu32-fp64| %.2e   |gpu32-cpu32| %.2e   (|logit| max %.1f)" % ((cert32.double() - cert64).abs().max(), (certg.double() - cert64).abs().max(), (certg - cert32).abs().max(), cert64.abs().max()))
# --- variants of the GPU GP ---
from roma_amd import ops
gpm = dec.gps["16"]
F32 = gpm.basis(2, 40, 40, xs.device).contiguous()
xs32, ys32 = xs.float().contiguous(), ys.float().contiguous()
Kyy = ops.cos_kernel(ys32, ys32, T=0.2, diag_add=0.1); Kxy = ops.cos_kernel(xs32, ys32, T=0.2)
def rep(name, mu_rows):
    mm = mu_rows.transpose(1, 2).reshape(2, -1, 40, 40).cpu().double()
    print("%-34s |.-fp64| %.2e  |.-cpu32| %.2e" % (name, (mm - mu64).abs().max(), (mm - mu32.double()).abs().max()), flush=True)
rep("K fp32 (MFMA) + spd_solve fp32", Kxy @ ops.spd_solve(Kyy.clone(), F32))
Z = torch.cholesky_solve(F32.double(), torch.linalg.cholesky(Kyy.double()))
rep("K fp32 (MFMA) + fp64 solve", (Kxy.double() @ Z))
rep("K fp32 (MFMA) + torch inv fp32", Kxy @ (torch.linalg.inv(Kyy) @ F32))
def cosk64(a, b):
    a, b = a.double(), b.double()
    c = torch.einsum("bnd,bmd->bnm", a, b) / (a.norm(dim=-1)[..., None] * b.norm(dim=-1)[:, None] + 1e-6)
    return ((c - 1.0) / 0.2).exp()
K64 = cosk64(ys32, ys32) + 0.1 * torch.eye(1600, device=xs.device, dtype=torch.float64)
Z = torch.cholesky_solve(F32.double(), torch.linalg.cholesky(K64))
rep("K fp64 + fp64 solve (fp32 features)", cosk64(xs32, ys32) @ Z)
print("cond estimate of K_yy+0.1I: %.1e" % float(torch.linalg.cond(K64[0].cpu())))
