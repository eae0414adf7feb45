import sys, time, torch
sys.path.insert(0, "/root/repo")
from oracle import roma_oracle as O
from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
torch.set_grad_enabled(False)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3: torch.set_num_threads(int(sys.argv[3]))
m = O.roma_model((lo, lo), (hi, hi)); load_synthetic_weights(m); m.encoder.dinov2_vitl14[0].eval()
pair = synthetic_pair(0, (lo, lo), (hi, hi))
import torch.autograd.profiler as prof
t = time.time(); 
with torch.autograd.profiler.profile() as p:
    w, c = m.match_tensors(*pair)
print("threads", torch.get_num_threads(), "time", time.time() - t, flush=True)
print(p.key_averages().table(sort_by="cpu_time_total", row_limit=12))
