#!/usr/bin/env python3
"""One RWKV-6 time-mix layer (callers.Tmix_x060, C = 2048, B x T = 48 x 512) forward + backward and forward only, with the
GroupNorm * gate epilogue fused into the operator's forward kernel and as a separate kernel:
    python tools/time_tmix_layer.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkv_lm_ext_amd import callers                                # noqa: E402

dev = torch.device("cuda", 0)
B, T, C = 48, 512, 2048
torch.manual_seed(0)
tm = callers.Tmix_x060(C, C).to(dev, torch.bfloat16)
with torch.no_grad():
    for n, p in tm.named_parameters():
        if p.dim() >= 2:
            p.normal_(0.0, 0.02)
        if "time_decay" in n and p.dim() == 3:
            p.copy_(torch.linspace(-6, -1, p.numel(), device=dev).view_as(p))
    tm.ln_x.weight.fill_(1.0)
x = torch.randn(B, T, C, device=dev).to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(B, T, C, device=dev).to(torch.bfloat16)


def timeit(fn, n=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def train():
    x.grad = None
    tm.zero_grad(set_to_none=True)
    tm(x).backward(dout)


def infer():
    with torch.no_grad():
        tm(x)


for rnd in range(3):
    for fuse in (True, False):
        tm.fuse_epilogue = fuse
        print(f"round {rnd} epilogue {'fused' if fuse else 'separate'}: fwd+bwd {timeit(train):.3f} ms, forward only {timeit(infer):.3f} ms", flush=True)
