#!/usr/bin/env python3
"""Which SIMD each wave of the forward's workgroups runs on (HW_REG_HW_ID read-back; needs a -DWKV6_CLOCK library, RWKV_AMD_LIB).

    RWKV_AMD_LIB=build_ab/pair0c/lib.so python tools/hwid_map.py [--B 8 --T 4096]
"""
import argparse
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth                                           # noqa: E402
from rwkv_lm_ext_amd import _lib, wkv6_op                         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--T", type=int, default=4096)
ap.add_argument("--H", type=int, default=32)
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
wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)              # (self-test, module load)
torch.cuda.synchronize()
lib.wkv6_set_debug_buffer(buf.data_ptr())
wkv6_op.forward_ex(r, k, v, w, u, H, y=y, ckpt=ckpt)
torch.cuda.synchronize()
d = buf.view(B * H, 16, 8).cpu()
role_simd = collections.defaultdict(collections.Counter)
hw_simd = collections.defaultdict(collections.Counter)
pairs = collections.Counter()
for bh in range(B * H):
    by_simd = collections.defaultdict(list)
    for wid in range(8):
        x = int(d[bh, wid, 5])
        if x == 0:
            continue
        hw, simd = x >> 32, (x >> 4) & 3
        role = ("P%d" % (wid - 4)) if wid >= 4 else ("C%d" % wid)
        role_simd[role][simd] += 1
        hw_simd[hw][simd] += 1
        by_simd[simd].append(role)
    pairs[tuple(sorted("".join(sorted(x[0] for x in v)) for v in by_simd.values()))] += 1
print("hardware wave -> SIMD (count over workgroups):")
for hw in sorted(hw_simd):
    print(f"  wave {hw}: {dict(hw_simd[hw])}")
print("role -> SIMD:")
for role in sorted(role_simd):
    print(f"  {role}: {dict(role_simd[role])}")
print("roles sharing a SIMD, per workgroup (C = consumer, P = producer):")
for kx, n in pairs.most_common():
    print(f"  {n:4d} workgroups: {kx}")
cyc = d[:, :8, 6].double().mean(0)
print("whole-life cycles per wave role (mean over workgroups): " + "  ".join(f"{'CCCCPPPP'[i]}{i & 3}={cyc[i].item():.0f}" for i in range(8)))
