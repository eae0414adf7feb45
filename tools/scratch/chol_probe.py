import time, torch
torch.manual_seed(0)
dev = "cuda"
x = torch.nn.functional.normalize(torch.randn(2, 1600, 512, device=dev), dim=-1)
K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(1600, device=dev)
F = torch.randn(2, 1600, 512, device=dev)

def timeit(name, fn, n=5):
    try:
        out = fn(); torch.cuda.synchronize()
        t = time.time()
        for _ in range(n): out = fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.time()-t)/n*1e3:.2f} ms", flush=True)
        return out
    except Exception as e:
        print(name, "FAILED", repr(e)[:200], flush=True)

ref = timeit("default cholesky+cholesky_solve", lambda: torch.cholesky_solve(F, torch.linalg.cholesky(K)))
for lib in ("cusolver", "magma"):
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
    except Exception as e:
        print("cannot set", lib, e); continue
    timeit(f"[{lib}] cholesky", lambda: torch.linalg.cholesky(K))
    timeit(f"[{lib}] cholesky+solve", lambda: torch.cholesky_solve(F, torch.linalg.cholesky(K)))
    timeit(f"[{lib}] cholesky per-matrix loop", lambda: [torch.linalg.cholesky(K[i]) for i in range(2)])
    timeit(f"[{lib}] linalg.solve", lambda: torch.linalg.solve(K, F))
    timeit(f"[{lib}] linalg.inv", lambda: torch.linalg.inv(K))
torch.backends.cuda.preferred_linalg_library("default")

def blocked_chol_solve(K, F, nb=128):
    """right-looking blocked Cholesky + two blocked triangular solves, all as small potrf/trsm + big GEMMs"""
    B, n, _ = K.shape
    A = K.clone()
    for j in range(0, n, nb):
        e = min(j + nb, n)
        L = torch.linalg.cholesky(A[:, j:e, j:e])
        A[:, j:e, j:e] = L
        if e < n:
            P = torch.linalg.solve_triangular(L, A[:, e:, j:e].transpose(1, 2), upper=False).transpose(1, 2)
            A[:, e:, j:e] = P
            A[:, e:, e:] -= P @ P.transpose(1, 2)
    L = torch.tril(A)
    Y = torch.linalg.solve_triangular(L, F, upper=False)
    return torch.linalg.solve_triangular(L.transpose(1, 2), Y, upper=True)

for nb in (64, 128, 256):
    o = timeit(f"blocked nb={nb}", lambda: blocked_chol_solve(K, F, nb))
    if o is not None and ref is not None: print("   err vs ref", float((o - ref).abs().max()), flush=True)

# Newton-Schulz / CG alternatives: CG with fixed iterations
def cg(K, F, iters=30):
    X = torch.zeros_like(F); R = F.clone(); P = R.clone(); rs = (R * R).sum(1, keepdim=True)
    for _ in range(iters):
        KP = K @ P
        a = rs / (P * KP).sum(1, keepdim=True)
        X = X + a * P; R = R - a * KP
        rs2 = (R * R).sum(1, keepdim=True)
        P = R + (rs2 / rs) * P; rs = rs2
    return X
for it in (20, 40):
    o = timeit(f"CG {it} iters", lambda: cg(K, F, it))
    if o is not None and ref is not None: print("   err vs ref", float((o - ref).abs().max()), flush=True)
