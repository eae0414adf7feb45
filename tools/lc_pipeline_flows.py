#!/usr/bin/env python3
"""local_correlation on the flows the random-init pipeline really produces (gpurun_out/pipeline_flows.npz, tools/dump_pipeline_flows.py):
   python tools/lc_pipeline_flows.py [--variants tile8x4 tile8x8 auto]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--file", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "pipeline_flows.npz"))
ap.add_argument("--variants", nargs="+", default=["tile8x4", "tile8x8", "auto"])
a = ap.parse_args()
d = np.load(a.file)
for v in a.variants:
    tot = 0.0
    line = f"{v:8s}"
    for k in d.files:
        C = int(k.split("_")[0][1:]); h = int(k.split("_h")[1].split("_")[0]); r = int(k.split("_r")[1].split("_")[0])
        flow = torch.from_numpy(d[k]).cuda()
        B = flow.shape[0]
        f0 = torch.randn(B, C, h, h, device="cuda").half().contiguous(memory_format=torch.channels_last)
        f1 = torch.randn(B, C, h, h, device="cuda").half().contiguous(memory_format=torch.channels_last)
        out = ops.nhwc_empty(B, (2 * r + 1) ** 2, h, h, torch.float16, "cuda")
        for _ in range(3):
            ops.local_correlation(f0, f1, r, flow=flow, out=out, variant=v)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.local_correlation(f0, f1, r, flow=flow, out=out, variant=v)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20 * 1e3
        tot += t
        line += f"  h{h} r{r} {t:7.1f}"
    print(line + f"  | sum {tot:7.1f} us", flush=True)
