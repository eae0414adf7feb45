import sys, time, torch
sys.path.insert(0, "/root/repo")
from roma_amd import ops
torch.manual_seed(0)
dev = "cuda"
for n, d in ((1600, 512), (100, 512), (54, 32), (777, 64)):
    x = torch.nn.functional.normalize(torch.randn(2, n, 64, device=dev), dim=-1)
    K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n, device=dev)
    F = torch.randn(2, n, d, device=dev)
    ref = torch.linalg.solve(K.double().cpu(), F.double().cpu())
    for nb in (64, 32):
        X = ops.spd_solve(K.clone(), F, nb=nb)
        torch.cuda.synchronize()
        print(f"n={n} nb={nb}: err vs fp64 {float((X.cpu().double() - ref).abs().max()):.2e}  (|X| max {float(ref.abs().max()):.1f})", flush=True)
x = torch.nn.functional.normalize(torch.randn(2, 1600, 512, device=dev), dim=-1)
K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(1600, device=dev)
F = torch.randn(2, 1600, 512, device=dev)
for nb in (64, 32):
    for _ in range(2): ops.spd_solve(K.clone(), F, nb=nb)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): ops.spd_solve(K.clone(), F, nb=nb)
    torch.cuda.synchronize(); print(f"spd_solve nb={nb}: {(time.time()-t)/10*1e3:.2f} ms", flush=True)
L = torch.linalg.cholesky(K); torch.cuda.synchronize(); t = time.time()
for _ in range(5): torch.cholesky_solve(F, torch.linalg.cholesky(K))
torch.cuda.synchronize(); print(f"torch cholesky+solve: {(time.time()-t)/5*1e3:.2f} ms")
