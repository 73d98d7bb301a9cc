#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own CPU paths and pin the oracle.

Run in the build container only (needs /root/reference, which never travels):

    python oracle/gen_golden.py

Reference functions executed (imported from /root/reference, never copied):
  * fla/ops/rwkv6/recurrent_naive.py:8-36  naive_recurrent_rwkv6  (+ autograd for
    grads, as the reference's own self-check does at :113-119; the hand-written
    naive_recurrent_rwkv6_bwd is broken)
  * src/model_encoder_run.py:30-62         run_rwkv6_forward (NO_CUDA=1)
Each fixture stores inputs (float32 values that are exactly bf16-representable)
and the reference outputs; the script asserts that oracle/wkv6_oracle.c and
oracle/wkv6_torch_naive.py reproduce them before anything is written.
"""
import os
import sys

REF = "/root/reference"
os.environ.setdefault("NO_CUDA", "1")
os.environ.setdefault("RWKV_HEAD_SIZE_A", "64")
os.environ.setdefault("RWKV_FLOAT_MODE", "fp32")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch

from fla.ops.rwkv6.recurrent_naive import naive_recurrent_rwkv6          # reference
from src.model_encoder_run import run_rwkv6_forward                      # reference

from oracle import wkv6_oracle as orc
from oracle.wkv6_torch_naive import wkv6_naive_fwd_bwd, wkv6_naive

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
N = 64


def bf16r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def make_inputs(seed, B, T, H, w_kind="init"):
    g = torch.Generator().manual_seed(seed)
    C = H * N
    r = bf16r(torch.randn(B, T, C, generator=g) * 0.5)
    k = bf16r(torch.randn(B, T, C, generator=g) * 0.5)
    v = bf16r(torch.randn(B, T, C, generator=g) * 0.5)
    if w_kind == "init":      # the model's own decay init ramp (src/model.py:408-411) + noise
        ramp = torch.tensor([-6 + 5 * (n / (C - 1)) ** (0.7 + 1.3 * 0.5) for n in range(C)])
        w = ramp.view(1, 1, C) + 0.1 * torch.randn(B, T, C, generator=g)
    elif w_kind == "stress":  # strong decays, d in ~(0.2, 0.95)
        w = -1.0 + 0.5 * torch.randn(B, T, C, generator=g)
    else:                     # "extreme": some channels forget within one token, some never decay
        w = -1.0 + 1.5 * torch.randn(B, T, C, generator=g)
        w[..., 0::7] = 2.5
        w[..., 3::11] = -9.0
    w = bf16r(w)
    u = bf16r(torch.randn(H, N, generator=g) * 0.3)
    gy = bf16r(torch.randn(B, T, C, generator=g))
    return r, k, v, w, u, gy


def to_fla(x, H):                      # [B,T,C] -> [B,H,T,N]
    B, T, C = x.shape
    return x.view(B, T, H, N).transpose(1, 2).contiguous()


def from_fla(x):                       # [B,H,T,N] -> [B,T,C]
    B, H, T, n = x.shape
    return x.transpose(1, 2).reshape(B, T, H * n)


def ref_fwd_bwd(r, k, v, w, u, gy, s0=None, reverse=False, use_u=True):
    """Reference forward + autograd grads on the [B,T,C] layout.
    s0 is in the kernels' layout [.., value j, key i]; fla wants [B,H,K,V]."""
    B, T, C = r.shape
    H = u.shape[0]
    leaves = [x.clone().requires_grad_(True) for x in (r, k, v, w, u)]
    rr, kk, vv, ww, uu = leaves
    s_leaf = None
    init = None
    if s0 is not None:
        s_leaf = s0.clone().requires_grad_(True)
        init = s_leaf.transpose(-1, -2)
        if init.dim() == 3:
            init = init.unsqueeze(0).expand(B, H, N, N)
    flip = (lambda x: x.flip(1)) if reverse else (lambda x: x)
    o = naive_recurrent_rwkv6(to_fla(flip(rr), H), to_fla(flip(kk), H), to_fla(flip(vv), H),
                              to_fla(flip(-torch.exp(ww)), H),
                              uu if use_u else torch.zeros_like(uu), initial_state=init)
    y = flip(from_fla(o))
    if gy is None:
        return y.detach(), None
    y.backward(gy)
    g = {n: (x.grad if x.grad is not None else torch.zeros_like(x))
         for n, x in zip(("gr", "gk", "gv", "gw", "gu"), leaves)}
    if s_leaf is not None:
        g["gs"] = s_leaf.grad
    return y.detach(), g


def ref_final_state(r, k, v, w, u, s0):
    """Final state of the reference recurrence, read out with probe tokens:
    r = one-hot(i), k = v = 0, decay 1 (raw w = -1e9)  =>  y[T+i][j] = S_T[i][j]."""
    B, T, C = r.shape
    H = u.shape[0]
    probe_r = torch.zeros(B, N, C)
    for i in range(N):
        probe_r[:, i, i::N] = 1.0
    z = torch.zeros(B, N, C)
    r2, k2, v2 = torch.cat([r, probe_r], 1), torch.cat([k, z], 1), torch.cat([v, z], 1)
    w2 = torch.cat([w, torch.full((B, N, C), -1e9)], 1)
    y, _ = ref_fwd_bwd(r2, k2, v2, w2, u, None, s0=s0)
    S = y[:, T:].reshape(B, N, H, N).permute(0, 2, 1, 3)            # [B,H,i,j]
    return S.transpose(-1, -2).contiguous()                         # [B,H,j,i]


def close(a, b, tol, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{what}: max-normalised error {err:.3e} > {tol}"
    return err


def check_oracle(name, inp, y, g, s0=None, tol=2e-5):
    r, k, v, w, u, gy = (x.numpy() for x in inp)
    s0n = None if s0 is None else s0.numpy()
    errs = {"y": close(orc.forward(r, k, v, w, u, s0n), y, tol, name + " oracle y")}
    og = orc.backward(r, k, v, w, u, gy, s0n)
    if s0 is not None and s0.dim() == 4:
        og["gs"] = og["gs_b"]          # per-sample state: per-sample gradient
    for n in g:
        errs[n] = close(og[n], g[n], tol, f"{name} oracle {n}")
    # the torch port (cpu_baseline) against the same reference
    yt, gt = wkv6_naive_fwd_bwd(*inp[:5], inp[5], s0=s0)
    close(yt, y, tol, name + " torch-port y")
    for n in g:
        close(gt[n], g[n], tol, f"{name} torch-port {n}")
    print(f"  {name}: oracle vs reference max-normalised errors " +
          " ".join(f"{n}={e:.1e}" for n, e in errs.items()))


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k_: np.asarray(v_) for k_, v_ in arrs.items()})
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.set_grad_enabled(True)
    # ---- plain wkv6, three decay regimes + edge lengths ------------------------------------------
    for name, (seed, B, T, H, kind) in {
        "wkv6_init":    (0, 2, 32, 2, "init"),
        "wkv6_stress":  (1, 2, 40, 2, "stress"),
        "wkv6_extreme": (2, 1, 24, 2, "extreme"),
        "wkv6_T1":      (3, 1, 1, 1, "stress"),
        "wkv6_T2":      (4, 1, 2, 1, "stress"),
        "wkv6_T3":      (5, 2, 3, 1, "stress"),
    }.items():
        inp = make_inputs(seed, B, T, H, kind)
        r, k, v, w, u, gy = inp
        y, g = ref_fwd_bwd(*inp)
        y2 = run_rwkv6_forward(r, k, v, w, u)            # second reference CPU path (fwd only)
        e = close(y2, y, 2e-5, name + " reference paths disagree")
        print(f"{name}: fla-naive vs model_encoder_run fwd = {e:.1e}")
        check_oracle(name, inp, y, g)
        save(name, r=r, k=k, v=v, w=w, u=u, gy=gy, y=y, **g)

    # ---- wkv6state: learnable initial state [H,N,N] shared over the batch ------------------------
    inp = make_inputs(10, 2, 32, 2, "init")
    gen = torch.Generator().manual_seed(110)
    s = bf16r(torch.randn(2, N, N, generator=gen) * 0.5)
    y, g = ref_fwd_bwd(*inp, s0=s)
    print("wkv6_state:")
    check_oracle("wkv6_state", inp, y, g, s0=s)
    save("wkv6_state", **dict(zip("r k v w u gy".split(), inp)), s=s, y=y, **g)

    # ---- wkv6infctx: per-sample carried state [B,H,N,N], final state out -------------------------
    inp = make_inputs(11, 2, 48, 2, "stress")
    gen = torch.Generator().manual_seed(111)
    s = bf16r(torch.randn(2, 2, N, N, generator=gen) * 0.5)
    y, g = ref_fwd_bwd(*inp, s0=s)
    s_final = ref_final_state(*inp[:5], s0=s)
    print("wkv6_infctx:")
    check_oracle("wkv6_infctx", inp, y, g, s0=s)
    _, so = orc.forward(*(x.numpy() for x in inp[:5]), s.numpy(), return_state=True)
    print(f"  final state: oracle vs reference(probe) = {close(so, s_final, 2e-5, 'final state'):.1e}")
    _, st = wkv6_naive(*inp[:5], s0=s, return_state=True)
    close(st, s_final, 2e-5, "torch-port final state")
    save("wkv6_infctx", **dict(zip("r k v w u gy".split(), inp)), s=s, y=y, s_final=s_final, **g)

    # ---- wkv6_bi: forward scan + reverse scan (u=0) over [0..L_b] --------------------------------
    B, T, H = 3, 40, 2
    inp = make_inputs(12, B, T, H, "stress")
    r, k, v, w, u, gy = inp
    mask = torch.ones(B, T, dtype=torch.int32)
    mask[0, 30:] = 0          # L_0 = 30
    mask[1, 17:] = 0          # L_1 = 17
    mask[2, 0:] = 0           # L_2 = 0 (single token)
    mask[0, 35] = 1           # a stray 1 after the first 0 must not matter
    ys = torch.zeros(B, T, H * N)
    gs_ = {n: torch.zeros_like(x) for n, x in zip(("gr", "gk", "gv", "gw"), (r, k, v, w))}
    gs_["gu"] = torch.zeros_like(u)
    for b in range(B):
        L = int((mask[b] == 0).nonzero()[0]) + 1 if (mask[b] == 0).any() else T
        sl = [x[b:b + 1, :L].contiguous() for x in (r, k, v, w)]
        y1, g1 = ref_fwd_bwd(*sl, u, gy[b:b + 1, :L].contiguous())
        y2, g2 = ref_fwd_bwd(*sl, u, gy[b:b + 1, :L].contiguous(), reverse=True, use_u=False)
        ys[b, :L] = (y1 + y2)[0]
        for n in ("gr", "gk", "gv", "gw"):
            gs_[n][b, :L] = (g1[n] + g2[n])[0]
        gs_["gu"] += g1["gu"]
    mn = mask.numpy()
    args = [x.numpy() for x in (r, k, v, w, u)]
    print("wkv6_bi:")
    print(f"  oracle y = {close(orc.bi_forward(mn, *args), ys, 2e-5, 'bi y'):.1e}")
    ob = orc.bi_backward(mn, *args, gy.numpy())
    for n in gs_:
        print(f"  oracle {n} = {close(ob[n], gs_[n], 2e-5, 'bi ' + n):.1e}")
    save("wkv6_bi", r=r, k=k, v=v, w=w, u=u, gy=gy, mask=mask, y=ys, **gs_)


if __name__ == "__main__":
    main()
