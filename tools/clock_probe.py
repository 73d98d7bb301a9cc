#!/usr/bin/env python3
"""In-kernel shader clock of the chunked kernels, the way MI355X_MICROARCH.md 'DVFS give-back' item 6 prescribes:
d(s_memtime) / d(s_memrealtime) x 100 MHz, one stamp pair around each wave's whole life (library built with -DWKV6_CLOCK:
no per-phase stamp in the loop), read after >= 2 s of back-to-back launches on random data; median over workgroups.

    RWKV_AMD_LIB=<clock build>/lib.so RWKV_AMD_NO_SELFTEST=1 python tools/clock_probe.py [--seconds 2.5]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import wkv6_op, _lib                         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--T", type=int, default=4096)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--seconds", type=float, default=2.5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, T, H = args.B, args.T, args.H
C = H * 64
r, k, v, w, u, gy = synth(B, T, H, dev)
y = torch.empty_like(r)
ckpt = wkv6_op.new_checkpoint(B, T, C, H, dev)
lib = _lib.load()
buf = torch.zeros(B * H * 16 * 8, dtype=torch.int64, device=dev)
lib.wkv6_set_debug_buffer.argtypes = [ctypes.c_void_p]
lib.wkv6_set_debug_buffer.restype = None
lib.wkv6_set_debug_buffer(buf.data_ptr())
fwd = lambda: wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
bwd = lambda: wkv6_op.backward_ex(r, k, v, w, u, gy, H, ckpt=ckpt)


def probe(name, fn, nwaves, mix=None):
    """Run `fn` (or alternately fn / mix, the way a training step does) back to back for args.seconds, then read the stamps of the
    last launch of `fn`."""
    t0 = time.time()
    n = 0
    while time.time() - t0 < args.seconds:
        for _ in range(50):
            if mix is not None:
                mix()
            fn()
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        if mix is not None:
            mix()
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    d = buf.view(B * H, 16, 8)[:, :nwaves, :].double().cpu()
    ghz = (d[:, :, 6] / d[:, :, 7] * 0.1).flatten()
    us = (d[:, :, 7] / 100.0).flatten()           # wave lifetime in microseconds (100 MHz reference)
    print(f"{name}: {n} launches in {args.seconds} s, {ms:.4f} ms per {'pair' if mix else 'launch'}; in-kernel clock median "
          f"{ghz.median().item():.3f} GHz (min {ghz.min().item():.3f}, max {ghz.max().item():.3f}); wave lifetime median {us.median().item():.1f} us",
          flush=True)


probe("forward alone", fwd, 8)
probe("backward alone", bwd, 12)
probe("forward in fwd+bwd steps", fwd, 8, mix=bwd)
probe("backward in fwd+bwd steps", bwd, 12, mix=fwd)
