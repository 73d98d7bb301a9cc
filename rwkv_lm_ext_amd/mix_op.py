"""Fused elementwise neighbours of the WKV6 operator in the RWKV-6 blocks, served by librwkv6_amd.so
(csrc/wkv6_mix.hip): the token-shift / data-dependent-lerp chain in front of the time-mix projections (src/model.py:435-448,
SURVEY.md row n4), the per-head GroupNorm + gate behind the operator (src/model.py:462-468, row n1), and the elementwise glue of
the channel-mix FFN (token shift + two lerps, squared ReLU, sigmoid gate: src/model.py:636-644).

bf16 GPU tensors only; like the operator itself there is no CPU path.  Each op is a torch.autograd.Function whose
forward and backward are one HIP kernel each; parameter gradients come back as fp32 partial rows summed here."""
import torch

from . import _lib
from .wkv6_op import _ptr, _stream_ptr                   # same pointer / stream conventions as the operator

_NPARTS = 1024


def _require(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16):
        raise RuntimeError(f"{name} must be a bf16 GPU tensor (the fused time-mix ops have no CPU path)")
    return t.contiguous()


class _DDLerp(torch.autograd.Function):
    """out[s] = x + (shift(x) - x) * (maa[s] + m[s]);  x [B,T,C], maa [NS,C], m [NS,B,T,C] or None -> out [NS,B,T,C].
    rev_n (int32 [B] or None): the shift runs over the stream whose first rev_n[b] tokens are reversed."""

    @staticmethod
    def forward(ctx, x, maa, m, shifted0, rev_n):
        x, maa = _require(x, "x"), _require(maa, "maa")
        m = None if m is None else _require(m, "m")
        shifted0 = None if shifted0 is None else _require(shifted0, "shifted0")
        B, T, C = x.shape
        if rev_n is not None and not (rev_n.dtype == torch.int32 and rev_n.is_contiguous() and tuple(rev_n.shape) == (B,)
                                      and rev_n.device == x.device):
            raise RuntimeError("rev_n must be a contiguous int32 [B] tensor on the device of x")
        NS = maa.shape[0]
        out = torch.empty((NS, B, T, C), device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            rc = _lib.load().wkv6_ddlerp_rev_forward(B, T, C, NS, _ptr(x), _ptr(shifted0), _ptr(m), _ptr(maa), _ptr(rev_n),
                                                     _ptr(out), _stream_ptr())
        _lib.check(rc, "ddlerp forward")
        ctx.save_for_backward(x, maa, m, shifted0, rev_n)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, maa, m, shifted0, rev_n = ctx.saved_tensors
        dout = _require(dout, "dout")
        B, T, C = x.shape
        NS = maa.shape[0]
        nparts = min(_NPARTS, B * T)
        dx = torch.empty_like(x)
        dm = None if m is None else torch.empty_like(m)
        part = torch.empty((nparts, NS, C), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            rc = _lib.load().wkv6_ddlerp_rev_backward(B, T, C, NS, _ptr(x), _ptr(shifted0), _ptr(m), _ptr(maa), _ptr(rev_n),
                                                      _ptr(dout), _ptr(dx), _ptr(dm), _ptr(part), nparts, _stream_ptr())
        _lib.check(rc, "ddlerp backward")
        dshift = None
        if shifted0 is not None and ctx.needs_input_grad[3]:
            # The token in front of the row (the infctx carry: the previous chunk's last token, src/model.py:1134-1190 passes the
            # shift states through torch_checkpoint without detaching them) enters only the stream's first token t0:
            # d shifted0[b] = sum_s dout[s,b,t0] (maa_s + m[s,b,t0]),  t0 = rev_n[b] - 1 where a reversed span starts the stream, else 0.
            if rev_n is None:
                d0 = dout[:, :, 0].float()
                m0 = None if m is None else m[:, :, 0].float()
            else:
                t0 = (rev_n.clamp(0, T).long() - 1).clamp_min(0).view(1, B, 1, 1).expand(NS, B, 1, C)
                d0 = dout.gather(2, t0)[:, :, 0].float()
                m0 = None if m is None else m.gather(2, t0)[:, :, 0].float()
            wgt = maa.float().view(NS, 1, C) + (0.0 if m0 is None else m0)
            dshift = (d0 * wgt).sum(0).to(shifted0.dtype)
        return dx, part.sum(0).to(maa.dtype), dm, dshift, None


def ddlerp(x, maa, m=None, shifted0=None, rev_n=None):
    """maa: [NS,C] (or anything reshapeable to it, e.g. five [1,1,C] parameters stacked)."""
    return _DDLerp.apply(x, maa.reshape(-1, x.shape[-1]), m, shifted0, rev_n)


def gn_gate_forward(y, g, gamma, beta, H, eps):
    """(out, stats): out = GroupNorm_H(y) * g on [rows, C]; stats fp32 [rows, H, 2] = mean, rstd (for gn_gate_backward)."""
    y, g, gamma, beta = _require(y, "y"), _require(g, "g"), _require(gamma, "gamma"), _require(beta, "beta")
    C = y.shape[-1]
    rows = y.numel() // C
    out = torch.empty_like(y)
    stats = torch.empty((rows, H, 2), device=y.device, dtype=torch.float32)
    with torch.cuda.device(y.device):
        rc = _lib.load().wkv6_gn_gate_forward(rows, C, H, _ptr(y), _ptr(g), _ptr(gamma), _ptr(beta), float(eps),
                                              _ptr(out), _ptr(stats), _stream_ptr())
    _lib.check(rc, "gn_gate forward")
    return out, stats


def gn_gate_backward(y, g, gamma, beta, stats, dout, H):
    """(dy, dg, dgamma, dbeta) of gn_gate_forward; the parameter gradients are summed from fp32 per-workgroup partial rows."""
    dout = _require(dout, "dout")
    C = y.shape[-1]
    rows = y.numel() // C
    nparts = min(_NPARTS, rows)
    dy, dg = torch.empty_like(y), torch.empty_like(g)
    pg = torch.empty((nparts, C), device=y.device, dtype=torch.float32)
    pb = torch.empty((nparts, C), device=y.device, dtype=torch.float32)
    with torch.cuda.device(y.device):
        rc = _lib.load().wkv6_gn_gate_backward(rows, C, H, _ptr(y), _ptr(g), _ptr(gamma), _ptr(beta), _ptr(stats),
                                               _ptr(dout), _ptr(dy), _ptr(dg), _ptr(pg), _ptr(pb), nparts, _stream_ptr())
    _lib.check(rc, "gn_gate backward")
    return dy, dg, pg.sum(0).to(gamma.dtype), pb.sum(0).to(beta.dtype)


class _GroupNormGate(torch.autograd.Function):
    """out = GroupNorm_H(y) * g  on [rows, C] with C = 64 H."""

    @staticmethod
    def forward(ctx, y, g, gamma, beta, H, eps):
        out, stats = gn_gate_forward(y, g, gamma, beta, H, eps)
        ctx.save_for_backward(_require(y, "y"), _require(g, "g"), gamma, beta, stats)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, dout):
        y, g, gamma, beta, stats = ctx.saved_tensors
        return gn_gate_backward(y, g, gamma, beta, stats, dout, ctx.H) + (None, None)


def group_norm_gate(y, g, gamma, beta, n_head, eps):
    return _GroupNormGate.apply(y, g, gamma, beta, n_head, eps)


class _SqRelu(torch.autograd.Function):
    """relu(x)^2 (src/model.py:640-641) in one pass; one rounding where the eager form rounds relu and the square separately (relu is
    exact, so the results are identical)."""

    @staticmethod
    def forward(ctx, x):
        x = _require(x, "x")
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = _lib.load().wkv6_sqrelu_forward(x.numel(), _ptr(x), _ptr(out), _stream_ptr())
        _lib.check(rc, "sqrelu forward")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dout = _require(dout, "dout")
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            rc = _lib.load().wkv6_sqrelu_backward(x.numel(), _ptr(x), _ptr(dout), _ptr(dx), _stream_ptr())
        _lib.check(rc, "sqrelu backward")
        return dx


class _SigMul(torch.autograd.Function):
    """sigmoid(r) * kv (src/model.py:643-644) in one pass, rounded once."""

    @staticmethod
    def forward(ctx, r, kv):
        r, kv = _require(r, "r"), _require(kv, "kv")
        if r.shape != kv.shape:
            raise RuntimeError("sigmoid_mul: r and kv must have the same shape")
        out = torch.empty_like(r)
        with torch.cuda.device(r.device):
            rc = _lib.load().wkv6_sigmul_forward(r.numel(), _ptr(r), _ptr(kv), _ptr(out), _stream_ptr())
        _lib.check(rc, "sigmul forward")
        ctx.save_for_backward(r, kv)
        return out

    @staticmethod
    def backward(ctx, dout):
        r, kv = ctx.saved_tensors
        dout = _require(dout, "dout")
        dr, dkv = torch.empty_like(r), torch.empty_like(kv)
        with torch.cuda.device(r.device):
            rc = _lib.load().wkv6_sigmul_backward(r.numel(), _ptr(r), _ptr(kv), _ptr(dout), _ptr(dr), _ptr(dkv), _stream_ptr())
        _lib.check(rc, "sigmul backward")
        return dr, dkv


def sqrelu(x):
    return _SqRelu.apply(x)


def sigmoid_mul(r, kv):
    return _SigMul.apply(r, kv)


def fusable(*tensors):
    """The HIP elementwise path applies: bf16 GPU tensors whose element count is a multiple of 8."""
    return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.numel() % 8 == 0 for t in tensors)
