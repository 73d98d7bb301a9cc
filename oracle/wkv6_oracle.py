"""ctypes front-end of oracle/wkv6_oracle.c (double-precision CPU restatement of
the reference's WKV6 kernels).  TEST INFRASTRUCTURE ONLY -- see the header of
wkv6_oracle.c for the reference file:line each formula follows and for how the
oracle is pinned (oracle/gen_golden.py -> tests/golden/).

All array arguments are float32 numpy arrays (bf16 test inputs are passed as
their exact float32 values); ``w`` is the RAW decay parameter.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwkv6_oracle.so")
_lib = None


def build(force=False):
    """Compile wkv6_oracle.c with gcc (a second or two)."""
    src = os.path.join(_HERE, "wkv6_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libwkv6_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _opt(a):
    if a is None:
        return None, None
    return _f(a)


def forward(r, k, v, w, u, s0=None, return_state=False):
    """y[B,T,C] (and final state [B,H,N,N]) for r,k,v,w [B,T,C], u [H,N].

    s0: None | [H,N,N] (wkv6state) | [B,H,N,N] (wkv6infctx), layout [.., value j, key i].
    """
    B, T, C = r.shape
    H = u.shape[0]
    N = C // H
    r_, rp = _f(r); k_, kp = _f(k); v_, vp = _f(v); w_, wp = _f(w); u_, up = _f(u)
    s_, sp = _opt(s0)
    per_batch = int(s0 is not None and np.ndim(s0) == 4)
    y = np.empty((B, T, C), np.float32)
    so = np.empty((B, H, N, N), np.float32) if return_state else None
    rc = lib().wkv6_oracle_forward(B, T, C, H, rp, kp, vp, wp, up, sp, per_batch,
                                   y.ctypes.data_as(ctypes.c_void_p),
                                   so.ctypes.data_as(ctypes.c_void_p) if return_state else None)
    assert rc == 0
    return (y, so) if return_state else y


def backward(r, k, v, w, u, gy, s0=None):
    """Returns dict(gr,gk,gv,gw [B,T,C]; gu [H,N] summed over B; gu_b [B,C];
    gs_b [B,H,N,N] per-batch dL/dS_0 when s0 is given, gs = sum over B)."""
    B, T, C = r.shape
    H = u.shape[0]
    N = C // H
    r_, rp = _f(r); k_, kp = _f(k); v_, vp = _f(v); w_, wp = _f(w); u_, up = _f(u)
    g_, gp = _f(gy)
    s_, sp = _opt(s0)
    per_batch = int(s0 is not None and np.ndim(s0) == 4)
    out = {n: np.empty((B, T, C), np.float32) for n in ("gr", "gk", "gv", "gw")}
    gu_b = np.empty((B, C), np.float32)
    gs_b = np.empty((B, H, N, N), np.float32) if s0 is not None else None
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().wkv6_oracle_backward(B, T, C, H, rp, kp, vp, wp, up, sp, per_batch, gp,
                                    P(out["gr"]), P(out["gk"]), P(out["gv"]), P(out["gw"]),
                                    P(gu_b), P(gs_b) if gs_b is not None else None)
    assert rc == 0
    out["gu_b"] = gu_b
    out["gu"] = gu_b.astype(np.float64).sum(0).reshape(H, N).astype(np.float32)
    if gs_b is not None:
        out["gs_b"] = gs_b
        out["gs"] = gs_b.astype(np.float64).sum(0).astype(np.float32)
    return out


def bi_forward(mask, r, k, v, w, u):
    B, T, C = r.shape
    H = u.shape[0]
    m = np.ascontiguousarray(mask, dtype=np.int32)
    r_, rp = _f(r); k_, kp = _f(k); v_, vp = _f(v); w_, wp = _f(w); u_, up = _f(u)
    y = np.empty((B, T, C), np.float32)
    rc = lib().wkv6_oracle_bi_forward(B, T, C, H, m.ctypes.data_as(ctypes.c_void_p),
                                      rp, kp, vp, wp, up, y.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return y


def bi_backward(mask, r, k, v, w, u, gy):
    B, T, C = r.shape
    H = u.shape[0]
    N = C // H
    m = np.ascontiguousarray(mask, dtype=np.int32)
    r_, rp = _f(r); k_, kp = _f(k); v_, vp = _f(v); w_, wp = _f(w); u_, up = _f(u)
    g_, gp = _f(gy)
    out = {n: np.empty((B, T, C), np.float32) for n in ("gr", "gk", "gv", "gw")}
    gu_b = np.empty((B, C), np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib().wkv6_oracle_bi_backward(B, T, C, H, m.ctypes.data_as(ctypes.c_void_p),
                                       rp, kp, vp, wp, up, gp,
                                       P(out["gr"]), P(out["gk"]), P(out["gv"]), P(out["gw"]), P(gu_b))
    assert rc == 0
    out["gu_b"] = gu_b
    out["gu"] = gu_b.astype(np.float64).sum(0).reshape(H, N).astype(np.float32)
    return out
