import sys, time, torch
sys.path.insert(0, "/root/repo")
from roma_amd.model_zoo import build_roma
from roma_amd.synthetic import load_synthetic_weights, synthetic_pair
torch.set_grad_enabled(False)
m = build_roma((560, 560), amp_dtype=torch.float16); load_synthetic_weights(m); m.upsample_res = (864, 864); m = m.cuda().eval()
pair = [t.cuda() for t in synthetic_pair(0)]
for _ in range(3): m.match_tensors(*pair)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); m.match_tensors(*pair); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host issue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
# back-to-back (queue stays full)
t0 = time.perf_counter()
for _ in range(10): m.match_tensors(*pair)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10 steps: host {1e2*(t1-t0):.2f} ms/step, total {1e2*(t2-t0):.2f} ms/step")
