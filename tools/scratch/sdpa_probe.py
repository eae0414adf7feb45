import torch, torch.nn.functional as F
torch.manual_seed(0)
q, k, v = (torch.randn(2, 8, 1600, 128) for _ in range(3))
ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double())
out = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda())
print("sdpa fp32 gpu vs fp64: %.2e" % (out.cpu().double() - ref).abs().max())
cpu = F.scaled_dot_product_attention(q, k, v)
print("sdpa fp32 cpu vs fp64: %.2e" % (cpu.double() - ref).abs().max())
qq, kk, vv = q.cuda(), k.cuda(), v.cuda()
a = ((qq * 128 ** -0.5) @ kk.transpose(-2, -1)).softmax(-1) @ vv
print("math fp32 gpu vs fp64: %.2e" % (a.cpu().double() - ref).abs().max())
with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.MATH):
    m = F.scaled_dot_product_attention(qq, kk, vv)
print("sdpa MATH backend gpu vs fp64: %.2e" % (m.cpu().double() - ref).abs().max())
x = torch.randn(3200, 1024); w = torch.randn(4097, 1024) * 6 / 32
r = x.double() @ w.double().t()
print("linear fp32 gpu vs fp64: %.2e (max |r| %.1f)" % (((x.cuda() @ w.cuda().t()).cpu().double() - r).abs().max(), r.abs().max()))
print("linear fp32 cpu vs fp64: %.2e" % ((x @ w.t()).double() - r).abs().max())
g = torch.randn(3200, 4096)
print("gelu gpu vs cpu: %.2e" % (F.gelu(g.cuda()).cpu() - F.gelu(g)).abs().max())
ln = torch.nn.LayerNorm(1024); t = torch.randn(3200, 1024) * 3
print("layernorm gpu vs cpu: %.2e" % (ln.cuda()(t.cuda()).cpu() - torch.nn.LayerNorm(1024)(t)).abs().max())
