#!/usr/bin/env python3
"""Run the forward and the backward of the operator a few times at config 2 (profiling driver for rocprofv3):
    python tools/run_bwd.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op                               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
B, T, H = 8, 4096, 32
C = H * 64
r, k, v, w, u, gy = synth(B, T, H, dev)
y = torch.empty_like(r)
ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)
for _ in range(n):
    wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
    wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
torch.cuda.synchronize()
