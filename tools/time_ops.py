#!/usr/bin/env python3
"""Time the forward and the backward of the WKV6 operator separately (config 2 by default) with HIP events.
Used by tools/ablate.sh with RWKV_AMD_LIB pointing at a variant library."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--T", type=int, default=4096)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--only", default="both", choices=["both", "fwd", "bwd"])
ap.add_argument("--no-ckpt", action="store_true", help="forward without checkpoints (inference-style call)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, T, H = args.B, args.T, args.H
C = H * 64
r, k, v, w, u, gy = synth(B, T, H, dev)
y = torch.empty_like(r)
ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)


def run(fn, n):
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


fwd = lambda: wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=None if args.no_ckpt else ckpt)
bwd = lambda: wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)
import time as _time
_t0 = _time.perf_counter()
while _time.perf_counter() - _t0 < float(os.environ.get("WKV6_PREWARM_S", "0.6")):     # sustained clocks first (profiles/r06_dvfs_transient.txt)
    for _ in range(16):
        fwd()
        if args.only != "fwd":
            bwd()
    torch.cuda.synchronize()
out = {}
if args.only == "both" and os.environ.get("WKV6_CLOCKS", "0") == "1":
    # inside fwd+bwd steps, launch by launch from the clock ring (what bench.py times): mean duration, clock and cycles per loop unit
    n = 4 * args.iters
    with wkv6_op.ClockProbe(dev, n_slots=64, n_launches=n) as ring:
        for _ in range(n):
            fwd()
            bwd()
        torch.cuda.synchronize()
        rec = ring.read()
    for side, unit in (("fwd", 64), ("bwd", 32)):
        us, ghz = rec[side + "_us_launches"], rec[side + "_ghz_launches"]
        if us and None not in us:
            m_us, m_ghz = sum(us) / len(us), sum(ghz) / len(ghz)
            out[f"step_{side}_us"] = round(m_us, 1)
            out[f"step_{side}_ghz"] = round(m_ghz, 3)
            out[f"step_{side}_cyc_per_{unit}tok"] = round(m_us * m_ghz * 1e3 / ((T + unit - 1) // unit))
probe = wkv6_op.ClockProbe(dev, n_slots=B * H) if os.environ.get("WKV6_CLOCKS", "0") == "1" else None   # in-run shader clocks (plain kernels)
if args.only in ("both", "fwd"):
    out["fwd_ms"] = round(run(fwd, args.iters), 4)
if args.only in ("both", "bwd"):
    out["bwd_ms"] = round(run(bwd, args.iters), 4)
if probe is not None:
    ck = probe.read()
    probe.close()
    out.update(ck)
    if ck.get("fwd_ghz") and "fwd_ms" in out:
        out["fwd_cycles_per_64_tokens"] = round(out["fwd_ms"] * 1e-3 * ck["fwd_ghz"] * 1e9 / ((T + 63) // 64))
    if ck.get("bwd_ghz") and "bwd_ms" in out:
        out["bwd_cycles_per_32_tokens"] = round(out["bwd_ms"] * 1e-3 * ck["bwd_ghz"] * 1e9 / ((T + 31) // 32))
print(os.environ.get("ABL_NAME", "default"), out, flush=True)
if os.environ.get("WKV6_STAMP", "0") == "1":       # diagnostic library: per-wave phase cycles of one backward launch
    import ctypes
    from rwkv_lm_ext_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(B * H * 16 * 8, dtype=torch.int64, device=dev)
    lib.wkv6_set_debug_buffer.argtypes = [ctypes.c_void_p]
    lib.wkv6_set_debug_buffer.restype = None
    lib.wkv6_set_debug_buffer(buf.data_ptr())
    fwd()
    torch.cuda.synchronize()
    d = buf.view(B * H, 16, 8).double().mean(0)
    ng = (T + 63) // 64
    print("forward, cycles per 64-token group and wave (avg over workgroups):")
    print("  consumers 0-3 = [body, barrier]   producers 4-7 = [load wait, prep, load issue, barrier]")
    for wv in range(8):
        print(f"  wave {wv:2d}: " + "  ".join(f"{(d[wv, k].item() / ng):8.0f}" for k in range(6)))
    buf.zero_()
    bwd()
    torch.cuda.synchronize()
    d = buf.view(B * H, 16, 8).double().mean(0)         # average over workgroups: [wave][phase]
    ns = (T + 31) // 32
    print("backward, cycles per 32-token stage and wave (avg over workgroups):")
    print("  row waves 0-3  = [tiles + ckpt wait, rebuild, pre-phase, chain, barrier]")
    print("  col waves 4-7  = [pre-phase, chain, barrier]")
    print("  producers 8-11 = [load wait, prep, load issue, barrier]")
    for wv in range(12):
        print(f"  wave {wv:2d}: " + "  ".join(f"{(d[wv, k].item() / ns):8.0f}" for k in range(6)))
    print("  records 12 (row wave 0) / 13 (column wave 4), per stage.  -DWKV6_STAMP5=1: cycles inside the polls of tag group [DA SC GA GB GC GD GE GF];")
    print("  -DWKV6_STAMP5=3: cycles since the start of the stage at  row: [GA settled, GB published, GC asked for, GC there]  column: [GA published, GB asked for, GB there, GC published]")
    for rec in (12, 13, 14, 15):     # (STAMP5=3: record 12 + w // 2, slots 4 (w % 2) .. + 3 = the four events of wave w, w = 0 .. 7)
        print(f"  rec  {rec:2d}: " + "  ".join(f"{(d[rec, k].item() / ns):8.0f}" for k in range(8)))
