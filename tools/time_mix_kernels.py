#!/usr/bin/env python3
"""Time the elementwise HIP kernels of csrc/wkv6_mix.hip at B x T = 48 x 512 rows, C = 2048 (BASELINE configs[2] layer shape)
and report their algorithmic HBM bandwidth:   python tools/time_mix_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkv_lm_ext_amd import mix_op                                 # noqa: E402

dev = torch.device("cuda", 0)
B, T, C, H = 48, 512, 2048, 32
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)
tc = B * T * C


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


rows = []
for ns, has_m in ((5, True), (1, False)):
    x = rnd(B, T, C).requires_grad_(True)
    maa = rnd(ns, C).requires_grad_(True)
    m = rnd(ns, B, T, C).requires_grad_(True) if has_m else None
    dout = rnd(ns, B, T, C)
    out = mix_op.ddlerp(x, maa, m)
    fwd = lambda: mix_op.ddlerp(x, maa, m)
    def bwd():
        x.grad = maa.grad = None
        if m is not None:
            m.grad = None
        out.backward(dout, retain_graph=True)
    bytes_f = (2 + (4 if has_m else 2) * ns) if has_m else 4
    bytes_b = (2 + 2 * ns + 2 * ns + 2 + 2 * ns) if has_m else 6
    rows.append((f"ddlerp fwd NS={ns}", timeit(fwd), bytes_f))
    rows.append((f"ddlerp bwd NS={ns} (+ torch partial sums)", timeit(bwd), bytes_b))
y, gt = rnd(B * T, C).requires_grad_(True), rnd(B * T, C).requires_grad_(True)
gamma, beta = rnd(C).requires_grad_(True), rnd(C).requires_grad_(True)
dout = rnd(B * T, C)
out = mix_op.group_norm_gate(y, gt, gamma, beta, H, 64e-5)
rows.append(("gn_gate fwd", timeit(lambda: mix_op.group_norm_gate(y, gt, gamma, beta, H, 64e-5)), 6))
def gbwd():
    y.grad = gt.grad = gamma.grad = beta.grad = None
    out.backward(dout, retain_graph=True)
rows.append(("gn_gate bwd (+ torch partial sums)", timeit(gbwd), 10))
print(f"# B x T = {B} x {T}, C = {C}: {tc / 1e6:.1f} M token-channels; HIP-event time per autograd call (kernel + allocations + the small torch reductions)")
for name, us, b in rows:
    print(f"{name:44s} {us:8.1f} us   {b:3d} B/tc   {b * tc / us / 1e6:5.2f} TB/s  = {100 * b * tc / us / 1e6 / 8:4.1f} % of 8 TB/s")
