import sys, time, torch
nt = int(sys.argv[1]); n = int(sys.argv[2])
torch.set_num_threads(nt)
g = torch.Generator().manual_seed(0)
x = torch.randn(2, n, 64, generator=g)
K = torch.exp((torch.nn.functional.normalize(x, dim=-1) @ torch.nn.functional.normalize(x, dim=-1).transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n)
t = time.time()
try:
    Ki = torch.linalg.inv(K)
    print(f"threads {nt} n {n}: inv {time.time()-t:.2f}s, err {float((Ki @ K - torch.eye(n)).abs().max()):.2e}", flush=True)
except Exception as e:
    print(f"threads {nt} n {n}: FAILED {e}", flush=True)
t = time.time()
L = torch.linalg.cholesky(K); print(f"  cholesky {time.time()-t:.2f}s", flush=True)
