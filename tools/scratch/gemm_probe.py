import time, torch
torch.set_grad_enabled(False)
shapes = [("L16", 3200, 1384), ("L8", 9800, 1144), ("L4", 39200, 576), ("L2", 156800, 144), ("L1", 627200, 24),
          ("U8", 23328, 1144), ("U4", 93312, 576), ("U2", 373248, 144), ("U1", 1492992, 24)]
tot = 0
for name, M, D in shapes:
    x = torch.randn(M, D, device="cuda", dtype=torch.half); w = torch.randn(D, D, device="cuda", dtype=torch.half); b = torch.randn(D, device="cuda", dtype=torch.half)
    for _ in range(3): torch.addmm(b, x, w)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): torch.addmm(b, x, w)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    tot += t * 9
    print(f"{name}: M={M} D={D}: {t*1e6:8.1f} us  {2*M*D*D/t/1e12:7.1f} TF/s  {(2*M*D*2)/t/1e9:8.1f} GB/s (in+out)", flush=True)
print("x9 blocks total ms:", tot * 1e3)
# proj GEMMs and transformer pieces
for name, M, K, N in [("proj16", 3200, 1024, 512), ("proj8", 9800, 512, 512), ("proj4", 39200, 256, 256), ("proj2", 156800, 128, 64), ("proj1", 627200, 64, 9),
                      ("proj8u", 23328, 512, 512), ("proj4u", 93312, 256, 256), ("proj2u", 373248, 128, 64), ("proj1u", 1492992, 64, 9), ("to_out", 3200, 1024, 4097)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.half); w = torch.randn(K, N, device="cuda", dtype=torch.half); b = torch.randn(N, device="cuda", dtype=torch.half)
    for _ in range(3): torch.addmm(b, x, w)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): torch.addmm(b, x, w)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"{name}: {M}x{K}x{N}: {t*1e6:8.1f} us  {2*M*K*N/t/1e12:7.1f} TF/s  {(M*K+M*N)*2/t/1e9:8.1f} GB/s", flush=True)
