import ctypes, os, sys, torch
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../scratch", os.path.basename(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(here, "libtoep_prof.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lib.roma_refiner_block.argtypes = [vp] * 7 + [i32] * 8 + [vp]
C, kpad, h = 24, 32, 864
x = torch.randn(2, h, h, C, device="cuda").half()
w25 = torch.zeros(25, kpad, device="cuda").half(); w25[:, :C] = torch.randn(25, C, device="cuda").half() * 0.2
wt = torch.zeros(kpad, kpad, device="cuda").half(); wt[:C, :C] = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
sc = torch.ones(kpad, device="cuda"); sh = torch.zeros(kpad, device="cuda"); b = torch.zeros(kpad, device="cuda")
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = lib.roma_refiner_block(x.data_ptr(), w25.data_ptr(), sc.data_ptr(), sh.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), 2, C, h, h, kpad, 1, C, C, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
lib.toep_prof_read(buf)
names = ["fetch issue", "depthwise", "barrier", "1x1", "barrier", "out staging", "scatter", "barrier", "stores", "barrier", "zero slots"]
t0 = min(buf[w * 16] for w in range(4))
print("rc", rc, " cycles (s_memtime) per part, per wave")
print("wave start " + " ".join("%11s" % n for n in names) + "  total")
for w in range(4):
    r = [buf[w * 16 + i] for i in range(12)]
    print("%4d %5d " % (w, r[0] - t0) + " ".join("%11d" % (r[i + 1] - r[i]) for i in range(11)) + " %6d" % (r[11] - r[0]))
