#!/usr/bin/env python3
"""GPU development check of the two-level backward (WKV6_BWD=64) against the 12-wave backward (WKV6_BWD=12) and, for small
shapes, the oracle; then same-process timing of both at config 2.  The switch is read by the library at every call.

    python tools/check_bwd64.py [--no-time]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--no-time", action="store_true")
ap.add_argument("--no-oracle", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
os.environ["WKV6_SPLIT"] = "0"      # the two-level kernel for every shape (small batch*head counts would keep the 12-wave one)


def run(which, r, k, v, w, u, gy, H, s0=None, use_ckpt=True):
    os.environ["WKV6_BWD"] = which
    B, T, C = r.shape
    ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev) if use_ckpt else None
    y = wkv6_op.forward_ex(r, k, v, w, u, H, s0=s0, ckpt=ckpt)
    out = wkv6_op.backward_ex(r, k, v, w, u, gy, H, s0=s0, want_gs=s0 is not None, ckpt=ckpt)
    torch.cuda.synchronize()
    return (y,) + tuple(out)


names = ("y", "gr", "gk", "gv", "gw", "gu", "gs")
bad = 0
for (B, T, H, wlo, whi, with_s0, use_ckpt) in [(1, 64, 1, -6, 0, False, True), (2, 83, 2, -6, 1, False, True), (1, 200, 2, -3, 1, True, True),
                                                (2, 130, 1, -2, 2.3, True, False), (3, 17, 1, -6, 1, False, True), (1, 1, 1, -1, 0, False, True),
                                                (2, 257, 2, -8, -3, True, True), (1, 77, 1, 0.5, 2.5, True, True)]:
    C = H * 64
    g = torch.Generator(device=dev).manual_seed(B * 1000 + T)
    bf = torch.bfloat16
    r, k, v, gy = (torch.randn(B, T, C, device=dev, generator=g).mul_(0.5).to(bf) for _ in range(4))
    w = (wlo + (whi - wlo) * torch.rand(B, T, C, device=dev, generator=g)).to(bf)
    u = (torch.randn(H, 64, device=dev, generator=g) * 0.3).to(bf)
    s0 = (torch.randn(B, H, 64, 64, device=dev, generator=g) * 0.3).to(bf) if with_s0 else None
    o12 = run("12", r, k, v, w, u, gy, H, s0, use_ckpt)
    o64 = run("64", r, k, v, w, u, gy, H, s0, use_ckpt)
    line = f"B{B} T{T} H{H} w[{wlo},{whi}] s0={with_s0} ckpt={use_ckpt}: 64-vs-12 "
    for n, a, b in zip(names, o12, o64):
        if a is None:
            continue
        a, b = a.float(), b.float()
        e = ((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item()
        nan = bool(torch.isnan(b).any())
        line += f"{n} {e:.1e}{'(NaN!)' if nan else ''} "
        if not (e < 2e-2) or nan:
            bad += 1
    print(line, flush=True)
    if not args.no_oracle and B * T * H <= 600:
        from oracle import wkv6_oracle as orc
        f = lambda t: t.float().cpu().numpy()
        ref = orc.backward(f(r), f(k), f(v), f(w), f(u), f(gy), s0=None if s0 is None else f(s0))
        line = "      vs oracle (max-normalised): "
        for n, t in zip(("gr", "gk", "gv", "gw"), o64[1:5]):
            e = np.abs(f(t) - ref[n]).max() / max(np.abs(ref[n]).max(), 1e-30)
            e12 = np.abs(f(o12[names.index(n)]) - ref[n]).max() / max(np.abs(ref[n]).max(), 1e-30)
            line += f"{n} {e:.1e} (12-wave {e12:.1e})  "
            if not (e < 8e-3):
                bad += 1
        e = np.abs(f(o64[5]) - ref["gu_b"]).max() / max(np.abs(ref["gu_b"]).max(), 1e-30)
        line += f"gu {e:.1e} "
        if s0 is not None:
            e = np.abs(f(o64[6]) - ref["gs_b"]).max() / max(np.abs(ref["gs_b"]).max(), 1e-30)
            line += f"gs {e:.1e}"
        print(line, flush=True)
print("MISMATCHES:", bad, flush=True)

if not args.no_time:
    B, T, H = 8, 4096, 32
    C = H * 64
    r, k, v, w, u, gy = synth(B, T, H, dev)
    y = torch.empty_like(r)

    def timeit(fn, n=40):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    res = {}
    for rnd in range(3):
        for which in ("12", "64"):
            os.environ["WKV6_BWD"] = which
            ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)
            fwd = lambda: wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
            bwd = lambda: wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
            for _ in range(20):
                fwd(); bwd()
            res.setdefault(which, []).append((round(timeit(fwd), 4), round(timeit(bwd), 4)))
    for which, v_ in res.items():
        print(f"WKV6_BWD={which}: (fwd_ms, bwd_ms) per round {v_}", flush=True)
    # full-size agreement of the two backwards
    outs = {}
    for which in ("12", "64"):
        os.environ["WKV6_BWD"] = which
        ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)
        wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
        outs[which] = wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
    torch.cuda.synchronize()
    line = "config 2, 64-vs-12: "
    for n, a, b in zip(names[1:6], outs["12"], outs["64"]):
        a, b = a.float(), b.float()
        line += f"{n} max {((a - b).abs().max() / a.abs().max()).item():.1e} differing {(a != b).float().mean().item():.3f}  "
    print(line, flush=True)
