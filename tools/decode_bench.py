#!/usr/bin/env python3
"""Latency of the stateful inference operator rwkv6 (cuda/rwkv6.cu) for decode-sized calls: T tokens per sequence, fp32 state
[B,H,64,64] updated in place, bf16 r/k/v/u/y, fp32 decay.  Bytes that must move per call: the state, read and written."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rwkv_lm_ext_amd.wkv6_op import rwkv6                          # noqa: E402

dev = torch.device("cuda", 0)
bf = torch.bfloat16
H, C = 32, 2048
for B, T in ((1, 1), (8, 1), (64, 1), (256, 1), (64, 4), (8, 16)):
    g = torch.Generator(device=dev).manual_seed(0)
    r, k, v = (torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf) for _ in range(3))
    w = torch.exp(-torch.exp(torch.randn(B, T, C, device=dev, generator=g) - 2.0)).contiguous()
    u = (torch.randn(H, 64, device=dev, generator=g) * 0.3).to(bf)
    state = torch.zeros(B, H, 64, 64, device=dev)
    y = torch.empty(B, T, C, device=dev, dtype=bf)
    fn = lambda: rwkv6.forward_bf16(B, T, C, H, state, r, k, v, w, u, y)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    gb = 2 * B * H * 64 * 64 * 4 / 1e9
    print(f"B={B:4d} T={T:3d}: {us:8.1f} us per call   state traffic {gb * 1e3:7.1f} MB -> {gb / (us * 1e-6):8.1f} GB/s", flush=True)
