#!/usr/bin/env python3
"""The flows that reach roma_local_corr inside one bench.py step (random-init weights), saved for offline statistics:
   python tools/dump_pipeline_flows.py gpurun_out/pipeline_flows.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from roma_amd import ops
from roma_amd.synthetic import synthetic_pair
dev = torch.device("cuda")
model = bench.build_model(dev, torch.float16)
pair = [t.to(dev) for t in synthetic_pair(0)]
saved = {}
orig = ops.local_correlation


def spy(f0, f1, r, padding_mode="zeros", flow=None, sample_mode="bilinear", **kw):
    saved[f"C{f0.shape[1]}_h{f0.shape[2]}_r{r}_{len(saved)}"] = flow.detach().float().cpu().numpy()
    return orig(f0, f1, r, padding_mode=padding_mode, flow=flow, sample_mode=sample_mode, **kw)


ops.local_correlation = spy
import roma_amd.matcher as M
if hasattr(M, "ops"):
    M.ops.local_correlation = spy
with torch.inference_mode():
    model.match_tensors(*pair)
torch.cuda.synchronize()
np.savez_compressed(sys.argv[1], **saved)
print({k: v.shape for k, v in saved.items()})
