import sys, time, torch, torch.nn.functional as F
torch.set_grad_enabled(False)
cfg = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), (256, 256), "M", (256, 512), (512, 512), (512, 512), (512, 512)]
def run(res, fmt, dt):
    ws = [(torch.randn(o, i, 3, 3, device="cuda", dtype=dt).contiguous(memory_format=fmt), torch.randn(o, device="cuda", dtype=dt)) for (i, o) in [c for c in cfg if c != "M"]]
    x0 = torch.randn(2, 3, res, res, device="cuda", dtype=dt).contiguous(memory_format=fmt)
    def fwd():
        x = x0; k = 0
        for c in cfg:
            if c == "M": x = F.max_pool2d(x, 2, 2)
            else:
                x = F.relu_(F.conv2d(x, ws[k][0], ws[k][1], padding=1)); k += 1
        return x
    for _ in range(3): fwd()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): fwd()
    torch.cuda.synchronize(); return (time.time() - t) / 10 * 1e3
for dt in (torch.float16, torch.bfloat16):
    for fmt, name in ((torch.channels_last, "NHWC"), (torch.contiguous_format, "NCHW")):
        print(dt, name, "560: %.2f ms  864: %.2f ms" % (run(560, fmt, dt), run(864, fmt, dt)), flush=True)
