import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import wkv6_oracle as orc
from rwkv_lm_ext_amd import wkv6_op
B, T, H = 1, 64, 1
g = torch.Generator().manual_seed(2)
C = H * 64
r, k, v, gy = [(torch.randn(B, T, C, generator=g)).bfloat16() for _ in range(4)]
w = (torch.randn(B, T, C, generator=g) * 1.0 - 1.0).bfloat16()
u = (torch.randn(C, generator=g) * 0.5).bfloat16()
f = lambda t: t.float().numpy().astype(np.float64)
ref = orc.backward(f(r), f(k), f(v), f(w), f(u).reshape(H, 64), f(gy))
d = [t.cuda() for t in (r, k, v, w, u, gy)]
out = wkv6_op.backward_ex(d[0], d[1], d[2], d[3], d[4].view(H, 64), d[5], H)
for n, o in zip(['gr', 'gw'], [out[0], out[3]]):
    e = np.abs(o.float().cpu().numpy()[0] - ref[n][0])        # [T, C]
    sc = np.abs(ref[n][0]).max()
    print(n, 'per token max err / scale:')
    print(np.array2string(e.max(1) / sc, precision=4, max_line_width=200))
    print(n, 'per channel max err / scale:')
    print(np.array2string(e.max(0) / sc, precision=4, max_line_width=200))
