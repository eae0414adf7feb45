"""CPU baseline + parity reference for bench.py — TEST INFRASTRUCTURE (oracle), run as a separate CPU-only process:

    python -m oracle.cpu_baseline --out DIR [--lo 560 --hi 864 --pair 0 --threads T]

Builds the oracle model with the same synthetic weights bench.py gives the GPU model (roma_amd.synthetic, seed 0),
runs ONE full symmetric 560->864 match on synthetic pair `--pair`, and writes DIR/cpu_baseline.json (seconds, threads)
and DIR/warp.npy, DIR/certainty.npy for the parity gate.  A fresh process keeps the OpenMP pool at the cgroup's CPU
share from the first parallel region on (the GPU process starts with one thread per host core)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--lo", type=int, default=560)
    ap.add_argument("--hi", type=int, default=864)
    ap.add_argument("--pair", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--images", nargs=2, default=None, help="two image files: match this real pair instead of a synthetic one")
    a = ap.parse_args()
    if a.threads:
        os.environ["OMP_NUM_THREADS"] = str(a.threads)
        os.environ["MKL_NUM_THREADS"] = str(a.threads)
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import roma_oracle as O
    from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
    if a.threads:
        torch.set_num_threads(a.threads)
    torch.set_grad_enabled(False)
    t0 = time.time()
    m = O.roma_model((a.lo, a.lo), (a.hi, a.hi))
    load_synthetic_weights(m, seed=0)
    m.encoder.dinov2_vitl14[0].eval()
    if a.images:
        from PIL import Image
        ims = [Image.open(f).convert("RGB") for f in a.images]
        pair = [O.preprocess(im, (a.lo, a.lo))[None] for im in ims] + [O.preprocess(im, (a.hi, a.hi))[None] for im in ims]
    else:
        pair = synthetic_pair(a.pair, (a.lo, a.lo), (a.hi, a.hi))
    build_s = time.time() - t0
    print(f"[cpu_baseline] model built in {build_s:.1f}s, {torch.get_num_threads()} threads; matching ...", file=sys.stderr, flush=True)
    t0 = time.time()
    warp, cert = m.match_tensors(*pair)
    dt = time.time() - t0
    os.makedirs(a.out, exist_ok=True)
    np.save(os.path.join(a.out, "warp.npy"), warp.numpy())
    np.save(os.path.join(a.out, "certainty.npy"), cert.numpy())
    json.dump({"seconds": dt, "threads": torch.get_num_threads(), "lo": a.lo, "hi": a.hi, "pair": a.pair},
              open(os.path.join(a.out, "cpu_baseline.json"), "w"))
    print(f"[cpu_baseline] one pair in {dt:.1f}s", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
