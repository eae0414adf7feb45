import sys, torch
sys.path.insert(0, "/root/repo")
from PIL import Image
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
from roma_amd.matcher import preprocess
torch.set_grad_enabled(False)
m = build_roma((560, 560), amp_dtype=torch.float32); load_synthetic_weights(m); m.upsample_res = (864, 864); m = m.cuda().eval()
ims = [Image.open(f"/root/repo/tests/golden/assets/sacre_coeur_{n}.jpg").convert("RGB") for n in "AB"]
pin = [preprocess(im, (560, 560))[None].cuda() for im in ims] + [preprocess(im, (864, 864))[None].cuda() for im in ims]
for name, inp in (("real", pin), ("synthetic", [t.cuda() for t in synthetic_pair(0)])):
    X = torch.cat((inp[0], inp[1])); pyr = m.encoder(X); c = m.decoder(pyr, None, swapped_pair=True)
    for s in (16, 8, 4, 2, 1):
        f, ce = c[s]["flow"], c[s]["certainty"]
        print(name, "scale", s, "flow absmax %.3f frac|f|>1 %.3f  cert mean %.2f min %.2f max %.2f" % (f.abs().max(), (f.abs() > 1).float().mean(), ce.mean(), ce.min(), ce.max()), flush=True)
    w, ce = m.match_tensors(*inp)
    print(name, "final cert mean %.4f frac==0 %.3f frac==1 %.3f" % (ce.mean(), (ce == 0).float().mean(), (ce == 1).float().mean()))
