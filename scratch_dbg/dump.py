import sys, os, numpy as np, torch
sys.path.insert(0, '.')
from rwkv_lm_ext_amd import wkv6_op
B, T, H = 1, 64, 1
g = torch.Generator().manual_seed(2)
C = H * 64
r, k, v, gy = [(torch.randn(B, T, C, generator=g)).bfloat16() for _ in range(4)]
w = (torch.randn(B, T, C, generator=g) * 1.0 - 1.0).bfloat16()
u = (torch.randn(C, generator=g) * 0.5).bfloat16()
d = [t.cuda() for t in (r, k, v, w, u, gy)]
out = wkv6_op.backward_ex(d[0], d[1], d[2], d[3], d[4].view(H, 64), d[5], H)
np.savez(sys.argv[1], **{n: o.float().cpu().numpy() for n, o in zip(['gr', 'gk', 'gv', 'gw'], out)})
