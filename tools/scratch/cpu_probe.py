import os, time, torch
import torch.nn.functional as F
print("cpu_count", os.cpu_count(), "threads default", torch.get_num_threads(), flush=True)
for nt in (16, 8, 32):
    torch.set_num_threads(nt)
    a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
    a @ b
    t = time.time(); 
    for _ in range(3): a @ b
    dt = (time.time() - t) / 3
    x = torch.randn(2, 64, 280, 280); w = torch.randn(64, 64, 3, 3)
    F.conv2d(x, w, padding=1)
    t = time.time(); F.conv2d(x, w, padding=1); dc = time.time() - t
    f1 = torch.randn(2, 512, 4900); idx = torch.randint(0, 4900, (2, 1, 4900)).expand(2, 512, 4900)
    t = time.time(); torch.gather(f1, 2, idx); dg = time.time() - t
    print(f"threads {nt}: matmul2048 {dt*1e3:.1f} ms ({2*2048**3/dt/1e9:.0f} GFLOP/s), conv {dc*1e3:.1f} ms, gather {dg*1e3:.1f} ms", flush=True)
