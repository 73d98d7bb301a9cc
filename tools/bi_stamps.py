#!/usr/bin/env python3
"""Cycles of a call of the persistent wkv6_bi backward launch by role: whole body, inside the stage loop, outside it (a -DWKV6_STAMP build;
    RWKV_AMD_LIB=build_ab/<stamp variant>/lib.so python tools/bi_stamps.py) -- profiles/r05_bi_call_overhead.txt."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.getcwd())
from bench import synth
from rwkv_lm_ext_amd import wkv6_op, _lib
dev = torch.device("cuda", 0)
B, T, H = 48, 512, 32
C = 64 * H
r, k, v, w, u, gy = synth(B, T, H, dev)
g = torch.Generator(device="cpu").manual_seed(1)
for label, lens in (("U[64,512]", torch.randint(64, 513, (B,), generator=g).to(dev)), ("all 512", torch.full((B,), 512, device=dev)), ("all 64", torch.full((B,), 64, device=dev))):
    mask = (torch.arange(T, device=dev).view(1, T) < (lens.view(B, 1) - 1)).to(torch.int32).contiguous()
    ws = wkv6_op.bi_new_workspace(B, T, C, H, dev)
    lib = _lib.load()
    buf = torch.zeros(B * H * 16 * 8, dtype=torch.int64, device=dev)
    lib.wkv6_set_debug_buffer.argtypes = [ctypes.c_void_p]; lib.wkv6_set_debug_buffer.restype = None
    lib.wkv6_set_debug_buffer(buf.data_ptr())
    for _ in range(3):
        wkv6_op.bi_forward_ex(mask, r, k, v, w, u, H, ws=ws); wkv6_op.bi_backward_ex(mask, r, k, v, w, u, gy, H, ws=ws)
    torch.cuda.synchronize(); buf.zero_()
    wkv6_op.bi_backward_ex(mask, r, k, v, w, u, gy, H, ws=ws)
    torch.cuda.synchronize()
    d = buf.view(B * H, 16, 8).double()
    # rows are (order-sorted) bh indices; the record is the row's LAST call (the reversed half)
    nst = ((lens + 31) // 32).double().mean().item()
    print(f"== {label}: mean stages per call {nst:.2f}")
    for name, ws_, nph in (("row", range(0, 4), 5), ("column", range(4, 8), 3), ("producer", range(8, 12), 4)):
        rec = d[:, list(ws_), :].mean(1)                      # [rows][8]
        loop = rec[:, :nph].sum(1); body = rec[:, 6]
        ok = body > 0
        print(f"  {name:9s} body {body[ok].mean().item():9.0f} cycles, inside the stage loop {loop[ok].mean().item():9.0f}, outside {(body - loop)[ok].mean().item():8.0f}; per stage {(loop[ok] / 1).mean().item() / nst:7.0f}")
