#!/usr/bin/env python3
"""Same-box, same-process A/B of several builds of librwkv6_amd.so (round 4, VERDICT r3 item 2: where did the forward's +8 % come from?).

    python tools/fwd_ab.py name=path/lib.so [name=path/lib.so ...] [--bwd name] [--rounds 3] [--clock name,name]

Every library is loaded with its own ctypes handle (RTLD_LOCAL), so the forward kernels of different builds run in ONE process on
ONE box, on the same tensors, alternating:
  * "alone":   each build's forward back to back (HIP events around 40 launches);
  * "in step": each build's forward inside fwd+bwd steps whose backward is always the SAME build (--bwd, default: the last
               library), timed with events around the forward only -- the condition the driver's bench measures;
  * --clock:   for builds compiled with -DWKV6_CLOCK, the in-kernel shader clock of the forward (d s_memtime / d s_memrealtime),
               read after the "in step" loop.
Only wkv6_forward_ckpt_ex / wkv6_backward_ex / wkv6_backward_workspace_bytes are bound: their signatures are the same in rounds 2-4.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402

VP, I, SZ, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint
W_RAW, CKPT_VALID, PARTIALS_F32 = 1, 32, 128

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--bwd", default=None)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--clock", default="")
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()

libs = {}
for spec in args.libs:
    name, path = spec.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.wkv6_forward_ckpt_ex.restype = I
    lib.wkv6_forward_ckpt_ex.argtypes = [I] * 4 + [VP] * 9 + [SZ, U, VP]
    lib.wkv6_backward_ex.restype = I
    lib.wkv6_backward_ex.argtypes = [I] * 4 + [VP] * 14 + [SZ, U, VP]
    lib.wkv6_backward_workspace_bytes.restype = SZ
    lib.wkv6_backward_workspace_bytes.argtypes = [I] * 4
    libs[name] = lib
bwd_name = args.bwd or list(libs)[-1]

dev = torch.device("cuda", 0)
B, T, H = 8, 4096, 32
C = H * 64
r, k, v, w, u, gy = synth(B, T, H, dev)
y = torch.empty_like(r)
gr, gk, gv, gw = (torch.empty_like(r) for _ in range(4))
gu = torch.empty(B, C, dtype=torch.float32, device=dev)
nbytes = max(lib.wkv6_backward_workspace_bytes(B, T, C, H) for lib in libs.values())
ckpt = torch.empty(nbytes, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()


def fwd(lib):
    rc = lib.wkv6_forward_ckpt_ex(B, T, C, H, p(r), p(k), p(v), p(w), p(u), None, None, p(y), p(ckpt), nbytes, W_RAW, st)
    assert rc == 0, rc


def bwd(lib):
    rc = lib.wkv6_backward_ex(B, T, C, H, p(r), p(k), p(v), p(w), p(u), None, p(gy), p(gr), p(gk), p(gv), p(gw), p(gu), None,
                              p(ckpt), nbytes, W_RAW | CKPT_VALID | PARTIALS_F32, st)
    assert rc == 0, rc


def time_alone(lib, n):
    for _ in range(10):
        fwd(lib)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fwd(lib)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def time_in_step(lib, blib, n):
    for _ in range(10):
        fwd(lib)
        bwd(blib)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    for i in range(n):
        ev[i][0].record()
        fwd(lib)
        ev[i][1].record()
        bwd(blib)
        ev[i][2].record()
    torch.cuda.synchronize()
    return (sum(e[0].elapsed_time(e[1]) for e in ev) / n, sum(e[1].elapsed_time(e[2]) for e in ev) / n)


# device pre-warm (sustained clocks), as bench.py does
for _ in range(64):
    fwd(libs[bwd_name])
    bwd(libs[bwd_name])
torch.cuda.synchronize()
print(f"backward build in the steps: {bwd_name}", flush=True)
for rnd in range(args.rounds):
    for name, lib in libs.items():
        a = time_alone(lib, args.iters)
        f, b_ = time_in_step(lib, libs[bwd_name], args.iters)
        print(f"round {rnd} {name:>8s}: forward alone {a:.4f} ms | in fwd+bwd steps: forward {f:.4f} ms, backward({bwd_name}) {b_:.4f} ms",
              flush=True)

for name in [n for n in args.clock.split(",") if n]:
    lib = libs[name]
    buf = torch.zeros(B * H * 16 * 8, dtype=torch.int64, device=dev)
    lib.wkv6_set_debug_buffer.argtypes = [VP]
    lib.wkv6_set_debug_buffer.restype = None
    lib.wkv6_set_debug_buffer(buf.data_ptr())
    import time
    t0 = time.time()
    while time.time() - t0 < 2.0:                 # >= 2 s of the step mix, then read the last forward's stamps
        for _ in range(50):
            bwd(libs[bwd_name])
            fwd(lib)
        torch.cuda.synchronize()
    d = buf.view(B * H, 16, 8)[:, :8, :].double().cpu()
    ghz = (d[:, :, 6] / d[:, :, 7] * 0.1).flatten()
    us = (d[:, :, 7] / 100.0).flatten()
    print(f"{name}: forward in fwd+bwd steps, in-kernel clock median {ghz.median().item():.3f} GHz (min {ghz.min().item():.3f}, "
          f"max {ghz.max().item():.3f}); wave lifetime median {us.median().item():.1f} us", flush=True)
    lib.wkv6_set_debug_buffer(None)
