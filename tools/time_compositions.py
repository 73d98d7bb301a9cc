#!/usr/bin/env python3
"""Forward+backward time of the bidirectional time-mix compositions (B: src/model_bi.py:325-350, C: src/model_ext.py:421-437)
with in-kernel reversal (wkv6_*_rev_ex, ddlerp rev_n) and with the reference's torch.gather formulation.
Shape: BASELINE configs[2] (B=48, T=512, C=2048, mask lengths U[64,512])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkv_lm_ext_amd import callers                               # noqa: E402

B, T, C = 48, 512, 2048
dev = torch.device("cuda", 0)
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
x0 = torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf)
dout = torch.randn(B, T, C, device=dev, generator=g).to(bf)
lens = torch.randint(64, T + 1, (B,), device=dev, generator=g)
mask = (torch.arange(T, device=dev).unsqueeze(0) < lens.unsqueeze(1)).int()
rev_idx = callers.reverse_x_idx(mask, T)


def module(in_kernel):
    torch.manual_seed(1)
    tm = callers.Tmix_x060(C, C).to(dev).to(bf)
    with torch.no_grad():
        for n, p in tm.named_parameters():
            if "time_" in n or "ln_x" in n:
                p.copy_(torch.randn_like(p) * 0.1)
        tm.time_decay.sub_(3.0)
    if not in_kernel:
        inner = tm.wkv
        tm.wkv = lambda *a: inner(*a)
    return tm


def run(tm, comp, n=20):
    def step():
        x = x0.clone().requires_grad_(True)
        out = tm.forward_bi_b(x, mask) if comp == "b" else tm.forward_bi_c(x, rev_idx, mask)
        out.backward(dout)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for comp in ("b", "c"):
    for rnd in range(2):
        tm = module(True)
        t_pair = run(tm, comp)
        tm.pair_launch = False
        t_in = run(tm, comp)
        t_ga = run(module(False), comp)
        print(f"composition {comp.upper()}: fwd+bwd of one time-mix layer  pair launch {t_pair:.3f} ms   in-kernel reversal, two launches "
              f"{t_in:.3f} ms   gather formulation {t_ga:.3f} ms", flush=True)

# the operator calls alone: the two problems of a layer as one launch per pass against two
from rwkv_lm_ext_amd import wkv6_op as op                         # noqa: E402
H = C // 64
mk = lambda: [torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf) for _ in range(3)] + \
    [(torch.randn(B, T, C, device=dev, generator=g) * 0.7 - 2.0).to(bf)]
p0, p1 = mk(), mk()
u = (torch.randn(H, 64, device=dev, generator=g) * 0.3).to(bf)
gy0, gy1 = dout, dout.flip(1).contiguous()
rev_n = lens.to(torch.int32)
ck = [op.new_checkpoint(B, T, C, H, dev) for _ in range(2)]
names = ("r", "k", "v", "w")


def two():
    op.forward_ex(*p0, u, H, ckpt=ck[0])
    op.forward_rev_ex(*p1, u, H, rev_n, op.REV_ALL, ckpt=ck[1])
    op.backward_ex(*p0, u, gy0, H, ckpt=ck[0])
    op.backward_rev_ex(*p1, u, gy1, H, rev_n, op.REV_ALL, ckpt=ck[1])


def one():
    sets = [dict(zip(names, p0), ckpt=ck[0]), dict(zip(names, p1), ckpt=ck[1], rev_n=rev_n, rev_mask=op.REV_ALL)]
    op.forward_pair_ex(H, u, sets)
    sets[0]["gy"], sets[1]["gy"] = gy0, gy1
    op.backward_pair_ex(H, u, sets)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rnd in range(3):
    print(f"operator only, both directions fwd+bwd (B=48, T=512, full length): two launches per pass {timeit(two):.4f} ms   "
          f"one launch per pass {timeit(one):.4f} ms", flush=True)
