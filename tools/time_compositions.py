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
    t_in, t_ga = run(module(True), comp), run(module(False), comp)
    print(f"composition {comp.upper()}: fwd+bwd of one time-mix layer  in-kernel reversal {t_in:.3f} ms   gather formulation {t_ga:.3f} ms", flush=True)
