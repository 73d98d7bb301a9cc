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


# Launch-bound regime (small batches): the operator launches on torch's current stream with caller-owned buffers, so a decode
# step's calls can be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed without the per-call binding overhead.
B, T, L = 8, 1, 24                                                  # 24 layers' worth of calls per replay
g = torch.Generator(device=dev).manual_seed(1)
r, k, v = (torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf) for _ in range(3))
w = torch.exp(-torch.exp(torch.randn(B, T, C, device=dev, generator=g) - 2.0)).contiguous()
u = (torch.randn(H, 64, device=dev, generator=g) * 0.3).to(bf)
states = [torch.zeros(B, H, 64, 64, device=dev) for _ in range(L)]
y = torch.empty(B, T, C, device=dev, dtype=bf)


def step():
    for st in states:
        rwkv6.forward_bf16(B, T, C, H, st, r, k, v, w, u, y)


step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
torch.cuda.synchronize()
y_eager = None
for name, fn in (("eager", step), ("graph replay", graph.replay)):
    for st in states:
        st.zero_()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"decode step of {L} calls, B={B}: {name:13s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
    if y_eager is None:
        y_eager = (y.clone(), states[0].clone())
    else:
        print("graph replay reproduces the eager results:", torch.equal(y, y_eager[0]) and torch.equal(states[0], y_eager[1]), flush=True)
