# phase timeline of the diagonal workgroup of roma_chol_step (instrumented scratch copy of chol.hip): 100 MHz wall clock stamps
import ctypes, os, torch
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../scratch", os.path.basename(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(here, "libchol_prof.so"))
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.roma_chol_diag_block.argtypes = [vp, i32, i64, vp, i32, i64, i32, i32, vp, i32, vp]
lib.roma_chol_step.argtypes = [vp, i32, i64, i32, i32, i32, i32, vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, i32, i32, vp]
torch.manual_seed(0)
dev = "cuda"
B, n, m, nb = 2, 1600, 512, 64
x = torch.nn.functional.normalize(torch.randn(B, n, 64, device=dev), dim=-1)
K = torch.exp((x @ x.transpose(1, 2) - 1) / 0.2) + 0.1 * torch.eye(n, device=dev)
F = torch.randn(B, n, m, device=dev)
steps = [(j, min(j + nb, n)) for j in range(0, n, nb)]
names = ["start", "loaded", "panels", "updated", "factor_in", "panels4", "inv_diag", "inv_16", "inv_32", "factored", "end", "p0.diag", "p0.solve", "p0.upd", "p1.diag", "p1.solve"]
for rep in range(2):
    A = torch.cat((K, F), dim=2).contiguous()
    W = torch.empty((B, len(steps), nb, nb), device=dev)
    R = torch.empty((B, len(steps), nb, n + m), device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.roma_chol_diag_block(A.data_ptr(), A.stride(1), A.stride(0), W.data_ptr(), nb, W.stride(0), nb, B, info.data_ptr(), 0, st)
    rows = []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(steps) + 1)]
    for s, (j, e) in enumerate(steps):
        last = s + 1 == len(steps)
        ev[s].record()
        lib.roma_chol_step(A.data_ptr(), A.stride(1), A.stride(0), n, n + m, j, e - j, W[:, s].data_ptr(), nb, W.stride(0), R[:, s].data_ptr(), n + m,
                           R.stride(0), None if last else W[:, s + 1].data_ptr(), nb, W.stride(0), info.data_ptr(), nb * (s + 1), B, st)
        if rep == 1:
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 16)()
            lib.chol_prof_read(buf)
            rows.append([buf[i] for i in range(16)])
    ev[len(steps)].record()
    torch.cuda.synchronize()
    if rep == 0:
        print("chain of 25 steps, back to back: %.1f us" % (ev[0].elapsed_time(ev[len(steps)]) * 1e3))
print("step  " + "  ".join("%9s" % names[i] for i in [1,2,3,4,11,12,13,14,15,5,6,7,8,10]) + "   (us since the workgroup started)")
for s, r in enumerate(rows[:-1:4]):
    print("%4d  " % s + "  ".join("%9.2f" % ((r[i] - r[0]) / 100.0) for i in [1,2,3,4,11,12,13,14,15,5,6,7,8,10]))
