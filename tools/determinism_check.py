#!/usr/bin/env python3
"""Run the config-2 forward + backward many times on the same inputs and compare every output bit for bit with the first run
(the kernels have no atomics: any difference would be a race, e.g. in the LDS-DMA checkpoint prefetch or a missing wait)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op                               # noqa: E402

dev = torch.device("cuda", 0)
bad = 0
for (B, T, H, runs) in ((8, 4096, 32, 300), (4, 2048, 32, 300), (3, 777, 5, 300)):
    C = 64 * H
    r, k, v, w, u, gy = synth(B, T, H, dev)
    ref = None
    for i in range(runs):
        ck = wkv6_op.new_checkpoint(B, T, C, H, dev)
        y = wkv6_op.forward_ex(r, k, v, w, u, H, ckpt=ck)
        out = (y,) + tuple(wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ck)[:5])
        if ref is None:
            ref = [t.clone() for t in out]
        else:
            for n, a, b in zip("y gr gk gv gw gu".split(), out, ref):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"MISMATCH B={B} T={T} H={H} run {i} {n}: {(a.float() - b.float()).abs().max().item():.3e}", flush=True)
    torch.cuda.synchronize()
    print(f"B={B} T={T} H={H}: {runs} runs compared", flush=True)
# wkv6_bi (fp32 side buffers, length-ordered dispatch, accumulate instantiation) and the partially reversed operator
B, T, H = 48, 512, 32
C = 64 * H
r, k, v, w, u, gy = synth(B, T, H, dev)
g = torch.Generator(device=dev).manual_seed(1)
lens = torch.randint(64, 513, (B,), device=dev, generator=g)
mask = (torch.arange(T, device=dev).view(1, T) < (lens.view(B, 1) - 1)).to(torch.int32).contiguous()
rev_n = lens.to(torch.int32).clamp(max=T).contiguous()
ref = None
for i in range(100):
    ws = wkv6_op.bi_new_workspace(B, T, C, H, dev)
    y = wkv6_op.bi_forward_ex(mask, r, k, v, w, u, H, ws=ws)
    out = (y,) + tuple(wkv6_op.bi_backward_ex(mask, r, k, v, w, u, gy, H, ws=ws))
    ck = wkv6_op.new_checkpoint(B, T, C, H, dev)
    yr = wkv6_op.forward_rev_ex(r, k, v, w, u, H, rev_n, wkv6_op.REV_K | wkv6_op.REV_V | wkv6_op.REV_Y, ckpt=ck)
    out += (yr,) + tuple(wkv6_op.backward_rev_ex(r, k, v, w, u, gy, H, rev_n, wkv6_op.REV_K | wkv6_op.REV_V | wkv6_op.REV_Y, ckpt=ck))
    if ref is None:
        ref = [t.clone() for t in out]
    else:
        for j, (a, b) in enumerate(zip(out, ref)):
            if not torch.equal(a, b):
                bad += 1
                print(f"MISMATCH bi/rev run {i} output {j}", flush=True)
torch.cuda.synchronize()
print("wkv6_bi + reversed operator, B=48 T=512 H=32: 100 runs compared", flush=True)
print("determinism check:", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
