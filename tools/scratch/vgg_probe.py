import sys, time, torch
sys.path.insert(0, "/root/repo")
from roma_amd.encoders import VGG19
from roma_amd.synthetic import synthetic_state_dict
torch.set_grad_enabled(False)
v = VGG19().eval(); v.load_state_dict(synthetic_state_dict({k: t.shape for k, t in v.state_dict().items()})); v = v.cuda()
def bench(tag):
    for res in (560, 864):
        x = torch.randn(2, 3, res, res, device="cuda")
        for _ in range(3): v(x, torch.float16)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10): v(x, torch.float16)
        torch.cuda.synchronize(); print(tag, res, "%.2f ms" % ((time.time() - t) / 10 * 1e3), flush=True)
bench("default")
torch.backends.cudnn.benchmark = True
bench("cudnn.benchmark=True")
