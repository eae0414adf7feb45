import sys, time, torch
sys.path.insert(0, "/root/repo")
from PIL import Image
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights
from roma_amd.matcher import preprocess
from oracle import roma_oracle as O
torch.set_grad_enabled(False); torch.set_num_threads(16)
m = build_roma((560, 560), amp_dtype=torch.float32); load_synthetic_weights(m); m.upsample_res = (864, 864)
o = O.roma_model((560, 560), (864, 864)); o.load_state_dict(m.state_dict()); o.encoder.dinov2_vitl14[0].load_state_dict(m.encoder.dinov2_vitl14[0].state_dict()); o.encoder.dinov2_vitl14[0].eval()
m = m.cuda().eval()
ims = [Image.open(f"/root/repo/tests/golden/assets/sacre_coeur_{n}.jpg").convert("RGB") for n in "AB"]
lo = [preprocess(im, (560, 560))[None] for im in ims]; hi = [preprocess(im, (864, 864))[None] for im in ims]
t = time.time()
X = torch.cat(lo); pyr_o = o.encoder(X); swapped = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr_o.items()}
co = o.decoder(pyr_o, swapped)
Xh = torch.cat(hi); pyr_oh = o.encoder(Xh, upsample=True); swh = {s: torch.cat((f.chunk(2)[1], f.chunk(2)[0])) for s, f in pyr_oh.items()}
uo = o.decoder(pyr_oh, swh, upsample=True, flow=co[1]["flow"], certainty=co[1]["certainty"], scale_factor=864 / 560)
print("oracle done %.1fs" % (time.time() - t), flush=True)
for rep in range(3):
    pyr = m.encoder(X.cuda()); c = m.decoder(pyr, None, swapped_pair=True)
    pyrh = m.encoder(Xh.cuda(), upsample=True); u = m.decoder(pyrh, None, upsample=True, flow=c[1]["flow"], certainty=c[1]["certainty"], scale_factor=864 / 560, swapped_pair=True)
    if rep == 0:
        for s in (16, 8, 4, 2, 1):
            print("feat", s, "max|d| %.2e (|f| max %.2f)" % ((pyr[s].float().cpu() - pyr_o[s]).abs().max(), pyr_o[s].abs().max()))
    for name, a, bb in (("coarse", c, co), ("up", u, uo)):
        for s in sorted(a, reverse=True):
            print(rep, name, s, "flow max|d| %.2e  cert-logit max|d| %.2e (|logit| max %.1f)" % ((a[s]["flow"].cpu() - bb[s]["flow"]).abs().max(), (a[s]["certainty"].cpu() - bb[s]["certainty"]).abs().max(), bb[s]["certainty"].abs().max()), flush=True)
