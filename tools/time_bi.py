#!/usr/bin/env python3
"""wkv6_bi forward / backward time by row-length distribution (where does the bidirectional operator's per-token cost come from?).

    python tools/time_bi.py            # B=48, T=512: lengths U[64,512] (BASELINE configs[2]), all 512, all 256, all 64; B=12, T=2048 full
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op                               # noqa: E402

dev = torch.device("cuda", 0)
H = 32


def run(B, T, lens, label, iters=40):
    C = 64 * H
    r, k, v, w, u, gy = synth(B, T, H, dev)
    mask = (torch.arange(T, device=dev).view(1, T) < (lens.view(B, 1) - 1)).to(torch.int32).contiguous()
    ws = wkv6_op.bi_new_workspace(B, T, C, H, dev)
    fwd = lambda: wkv6_op.bi_forward_ex(mask, r, k, v, w, u, H, ws=ws)
    bwd = lambda: wkv6_op.bi_backward_ex(mask, r, k, v, w, u, gy, H, ws=ws)
    out = []
    for f in (fwd, bwd):
        for _ in range(10):
            fwd(); bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters)
    ntok = int(lens.sum().item())
    units = ntok * H * 2                                           # (token, head, direction)
    g64 = int(((lens + 63) // 64).sum().item()) * H * 2
    s32 = int(((lens + 31) // 32).sum().item()) * H * 2
    print(f"{label:34s} fwd {out[0]:.4f} ms  bwd {out[1]:.4f} ms | valid tokens {ntok:6d} | ns per (token, head, dir): fwd {out[0] * 1e6 / units:.3f} bwd {out[1] * 1e6 / units:.3f}"
          f" | us per 64-token group and slot: fwd {out[0] * 1e3 / (g64 / 256):.2f}  per 32-token stage: bwd {out[1] * 1e3 / (s32 / 256):.2f}", flush=True)


torch.manual_seed(0)
g = torch.Generator(device="cpu").manual_seed(1)
run(48, 512, torch.randint(64, 513, (48,), generator=g).to(dev), "B=48 T=512 lengths U[64,512]")
run(48, 512, torch.full((48,), 512, device=dev), "B=48 T=512 all 512")
run(48, 512, torch.full((48,), 256, device=dev), "B=48 T=512 all 256")
run(48, 512, torch.full((48,), 64, device=dev), "B=48 T=512 all 64")
run(12, 2048, torch.full((12,), 2048, device=dev), "B=12 T=2048 all 2048")
run(8, 4096, torch.full((8,), 4096, device=dev), "B=8 T=4096 all 4096 (one row per slot)")
