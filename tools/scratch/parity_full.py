import sys, time, torch, numpy as np, os, subprocess, tempfile, json
sys.path.insert(0, "/root/repo")
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
torch.set_grad_enabled(False)
out = tempfile.mkdtemp()
env = dict(os.environ, HIP_VISIBLE_DEVICES="")
p = subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--out", out, "--threads", "16"], cwd="/root/repo", env=env)
m = build_roma((560, 560), amp_dtype=torch.float32); load_synthetic_weights(m); m.upsample_res = (864, 864); m = m.cuda().eval()
pair = [t.cuda() for t in synthetic_pair(0)]
w, c = m.match_tensors(*pair)
# flow coherence stats at each level
X = torch.cat((pair[0], pair[1])); pyr = m.encoder(X); cc = m.decoder(pyr, None, swapped_pair=True)
for s in (16, 8, 4, 2, 1):
    f = cc[s]["flow"]; dx = (f[:, :, :, 1:] - f[:, :, :, :-1]); print("scale", s, "flow range", float(f.min()), float(f.max()), "median |dflow/dx| in px", float(dx[:, 0].abs().median()) * f.shape[-1] / 2, flush=True)
p.wait()
rw = torch.from_numpy(np.load(out + "/warp.npy")); rc = torch.from_numpy(np.load(out + "/certainty.npy"))
dw = (w.cpu() - rw).abs(); dc = (c.cpu() - rc).abs()
print("fp32 parity full: warp max %.3e frac>1e-3 %.3e ; cert max %.3e frac>1e-3 %.3e" % (dw.max(), (dw > 1e-3).float().mean(), dc.max(), (dc > 1e-3).float().mean()))
print("cert mean", float(c.mean()), "frac cert>0.05", float((c > 0.05).float().mean()))
