#!/usr/bin/env python3
"""Per-stage wall times of one 560->864 match on the GPU (sync after each stage), progress lines flushed as it goes."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
from roma_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--lo", type=int, default=560)
ap.add_argument("--hi", type=int, default=864)
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[args.dtype]
torch.set_grad_enabled(False)


def say(*a):
    print(f"[{time.strftime('%H:%M:%S')}]", *a, flush=True)


say("building")
m = build_roma((args.lo, args.lo), amp_dtype=dt)
load_synthetic_weights(m)
m.upsample_res = (args.hi, args.hi)
m = m.cuda().eval()
pair = [t.cuda() for t in synthetic_pair(0, (args.lo, args.lo), (args.hi, args.hi))]
say("built")


class T:
    def __init__(self, name):
        self.name = name
    def __enter__(self):
        torch.cuda.synchronize(); self.t = time.time()
    def __exit__(self, *a):
        torch.cuda.synchronize(); say(f"  {self.name}: {(time.time()-self.t)*1e3:.2f} ms")


for rep in range(args.reps):
    say(f"rep {rep}")
    X = torch.cat((pair[0], pair[1]))
    with T("vgg 560"):
        pyr = m.encoder.cnn(X, dt)
    with T("vit 560"):
        t = m.encoder._vit(X.device).patch_tokens(X.to(dt))
    with T("encoder(560) total"):
        pyr = m.encoder(X)
    dec = m.decoder
    orig = {s: r.forward for s, r in dec.conv_refiner.items()}
    for s, r in dec.conv_refiner.items():
        def wrapped(*a, _s=s, _f=orig[s], **k):
            with T(f"refiner {_s} @{tuple(a[0].shape[-2:])}"):
                return _f(*a, **k)
        r.forward = wrapped
    with T("decoder coarse total"):
        c = dec(pyr, None, swapped_pair=True)
    Xh = torch.cat((pair[2], pair[3]))
    with T("vgg 864"):
        pyrh = m.encoder(Xh, upsample=True)
    with T("decoder upsample total"):
        u = dec(pyrh, None, upsample=True, flow=c[1]["flow"], certainty=c[1]["certainty"], scale_factor=864 / 560, swapped_pair=True)
    with T("finalize"):
        ops.match_finalize(u[1]["flow"], u[1]["certainty"], c[16]["certainty"])
    for s, r in dec.conv_refiner.items():
        r.forward = orig[s]
    with T("match_tensors total"):
        m.match_tensors(*pair)
