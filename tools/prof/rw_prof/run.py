import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../.."))
from roma_amd import ops
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../scratch", os.path.basename(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(here, "librw_prof.so"))
vp, i32 = ctypes.c_void_p, ctypes.c_int
lib.roma_refiner_block_wide.argtypes = [vp] * 7 + [i32] * 7 + [vp]
D, h = 576, 216
torch.manual_seed(0)
x = torch.randn(2, h, h, D, device="cuda").half()
w25 = (torch.randn(25, D) * 0.2).half()
wt = (torch.randn(D, D) / D ** 0.5).half()
w25p = ops.refiner_wide_taps(w25).cuda(); wp = ops.refiner_wide_pack(wt).cuda()
sc = torch.ones(D, device="cuda"); sh = torch.zeros(D, device="cuda"); b = torch.zeros(D, device="cuda")
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = lib.roma_refiner_block_wide(x.data_ptr(), w25p.data_ptr(), sc.data_ptr(), sh.data_ptr(), wp.data_ptr(), b.data_ptr(), out.data_ptr(), 2, h, h, D, D, D, 1, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.rw_prof_read(buf)
names = ["issue dma/loads", "dw (waves 4-7)", "mfma", "dw (waves 0-3)", "store_x", "vmcnt(0)", "barrier"]
t0 = min(buf[w * 16] for w in range(8))
print("rc", rc, " workgroup 300, phase kp = 8: cycles (s_memtime) per part, per wave; start offsets relative to the first wave")
print("wave  start  " + "  ".join("%16s" % n for n in names) + "   total")
for w in range(8):
    r = [buf[w * 16 + i] for i in range(8)]
    print("%4d %6d  " % (w, r[0] - t0) + "  ".join("%16d" % (r[i + 1] - r[i]) for i in range(7)) + "  %6d" % (r[7] - r[0]))
