import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import wkv6_oracle as orc
from rwkv_lm_ext_amd import wkv6_op
from tests.conftest import bf16_report
def run(B, T, H, seed):
    g = torch.Generator().manual_seed(seed)
    C = H * 64
    r, k, v, gy = [(torch.randn(B, T, C, generator=g)).bfloat16() for _ in range(4)]
    w = (torch.randn(B, T, C, generator=g) * 1.0 - 1.0).bfloat16()
    u = (torch.randn(C, generator=g) * 0.5).bfloat16()
    f = lambda t: t.float().numpy().astype(np.float64)
    ref = orc.backward(f(r), f(k), f(v), f(w), f(u).reshape(H, 64), f(gy))
    d = [t.cuda() for t in (r, k, v, w, u, gy)]
    r_, k_, v_, w_, u_, gy_ = d
    out = wkv6_op.backward_ex(r_, k_, v_, w_, u_.view(H, 64), gy_, H)
    names = ['gr', 'gk', 'gv', 'gw', 'gu']
    for n, o in zip(names, out):
        key = 'gu_b' if n == 'gu' else n
        rr = ref[key]
        rep = bf16_report(o.float().cpu().numpy().reshape(rr.shape), rr, floor=0.1 if n == 'gw' else 1e-3)
        print(B, T, H, n, rep)
for cfg in [(1, 16, 1, 1), (1, 64, 1, 2), (2, 200, 2, 3), (1, 1024, 2, 4)]:
    run(*cfg)
