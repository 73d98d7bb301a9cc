import os, sys, torch
sys.path.insert(0, "/root/repo")
from rwkv_lm_ext_amd import wkv6_op
dev = torch.device("cuda", 0)
B, T, H, wlo, whi = 1, 77, 1, 0.5, 2.5
C = H * 64
g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
bf = torch.bfloat16
r, k, v, gy = (torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf) for _ in range(4))
w = (wlo + (whi - wlo) * torch.rand(B, T, C, device=dev, generator=g)).to(bf)
u = (torch.randn(H, 64, device=dev, generator=g) * 0.3).to(bf)
s0 = (torch.randn(B, H, 64, 64, device=dev, generator=g) * 0.3).to(bf)
outs = {}
for which in ("12", "64"):
    os.environ["WKV6_BWD"] = which
    ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)
    wkv6_op.forward_ex(r, k, v, w, u, H, s0=s0, ckpt=ckpt)
    outs[which] = wkv6_op.backward_ex(r, k, v, w, u, gy, H, s0=s0, want_gs=True, ckpt=ckpt)
torch.cuda.synchronize()
for n, a, b in zip(("gr", "gk", "gv", "gw"), outs["12"], outs["64"]):
    d = (a.float() - b.float()).abs()[0]
    idx = torch.nonzero(d > 0.02 * a.float().abs().max())
    print(n, "max", d.max().item(), "count", idx.shape[0], "tokens", sorted(set(idx[:, 0].tolist()))[:40], "channels", sorted(set(idx[:, 1].tolist()))[:64])
    for t_, c_ in idx[:6].tolist():
        print("    ", t_, c_, a[0, t_, c_].item(), b[0, t_, c_].item())
