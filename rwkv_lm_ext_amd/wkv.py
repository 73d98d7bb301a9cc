"""Autograd layer: WKV_6 / WKV_6STATE / WKV_6_BI and the RUN_CUDA_RWKV6* helpers of the reference.

Mirrors, argument for argument:
  WKV_6.apply(B, T, C, H, r, k, v, w, u) -> y                    src/model.py:191-233  (dup src/model_bi.py:49-94)
  RUN_CUDA_RWKV6(B, T, C, H, r, k, v, w, u)                       src/model.py:235-236
  WKV_6STATE.apply(B, T, C, H, r, k, v, w, u, s) -> y             src/model.py:137-182 ('states': s [H,N,N])
                                                                  src/model.py:83-128  ('infctx': s [B,H,N,N])
  RUN_CUDA_RWKV6_STATE(B, T, C, H, r, k, v, w, u, s)              src/model.py:184-185 (states -> y)
                                                                  src/model.py:130-132 (infctx -> (y, s))
  WKV_6_BI.apply(B, T, C, H, mask, r, k, v, w, u) -> y            cuda/wkv6_bi.py:13-60

Same dtype/contiguity asserts and the same gradient tuples.  Differences, all deliberate:
  * the raw bf16 decay `w` goes straight to the kernel (no fp32 `ew = -exp(w.float())` pass, no fp32
    tensor kept alive for backward -- src/model.py:210-211);
  * gu / gs are reduced over the batch in fp32 and rounded once (the reference sums bf16 partials
    in bf16, src/model.py:232);
  * infctx keeps the INITIAL state for backward (the reference saves the tensor the kernel then
    overwrites); the caller's `s` is still updated in place and returned;
  * WKV_6_BI's backward is the adjoint of its forward.
"""
import os

import torch

from . import wkv6_op

HEAD_SIZE = int(os.environ.get("RWKV_HEAD_SIZE_A", "64"))


def _assert_inputs(C, H, *tensors):
    for t in tensors:
        assert t.dtype == torch.bfloat16
        assert t.is_contiguous()
    assert HEAD_SIZE == C // H


def _keep_ckpt(ctx):
    """Let the forward store its fp32 state checkpoints (4 B per token-channel, e.g. 268 MB per layer at B=8, T=4096, C=2048)
    for the backward?  Only when a backward will follow, and not with RWKV_AMD_NO_CKPT=1: then nothing is kept between the two
    calls and the backward rebuilds the checkpoints with a state pass (+≈0.23 ms at that shape) -- the choice when activation
    memory matters more than time."""
    return any(ctx.needs_input_grad) and os.environ.get("RWKV_AMD_NO_CKPT", "0") != "1"


def _sum_bf16(partials, shape):
    return partials.float().sum(0).to(torch.bfloat16).view(shape)


class WKV_6(torch.autograd.Function):
    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u)
            ctx.B, ctx.T, ctx.C, ctx.H = B, T, C, H
            ctx.save_for_backward(r, k, v, w, u)
            # when a backward will follow, let the forward store its per-64-token state checkpoints (fp32, 4 B per
            # token-channel) so the backward does not have to recompute them with a state pass
            ctx.ckpt = wkv6_op.new_checkpoint(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            return wkv6_op.forward_ex(r, k, v, w, u, H, ckpt=ctx.ckpt)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            gy = gy.contiguous()
            r, k, v, w, u = ctx.saved_tensors
            gr, gk, gv, gw, gu, _ = wkv6_op.backward_ex(r, k, v, w, u, gy, ctx.H, ckpt=ctx.ckpt)
            ctx.ckpt = None
            gu = _sum_bf16(gu, (ctx.H, ctx.C // ctx.H))
            return (None, None, None, None, gr, gk, gv, gw, gu)


class WKV_6_REV(torch.autograd.Function):
    """WKV_6 on partially reversed sequences: what the reference writes as
    `reverse_x(WKV_6.apply(.., reverse_x(k, idx), reverse_x(v, idx), ..), idx)` (src/model_bi.py:331-348, rev_mask =
    REV_K | REV_V | REV_Y) or with every tensor reversed (src/model_ext.py:421-437, REV_ALL), without the gathers:
    `rev_n[b]` = number of leading tokens of row b that are reversed (= mask.sum(1), src/model_ext.py:410-417)."""

    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u, rev_n, rev_mask):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u)
            ctx.H, ctx.C, ctx.rev_mask = H, C, rev_mask
            ctx.save_for_backward(r, k, v, w, u, rev_n)
            ctx.ckpt = wkv6_op.new_checkpoint(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            return wkv6_op.forward_rev_ex(r, k, v, w, u, H, rev_n, rev_mask, ckpt=ctx.ckpt)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            r, k, v, w, u, rev_n = ctx.saved_tensors
            gr, gk, gv, gw, gu = wkv6_op.backward_rev_ex(r, k, v, w, u, gy.contiguous(), ctx.H, rev_n, ctx.rev_mask,
                                                          ckpt=ctx.ckpt)
            ctx.ckpt = None
            return (None, None, None, None, gr, gk, gv, gw, _sum_bf16(gu, (ctx.H, ctx.C // ctx.H)), None, None)


class WKV_6_PAIR(torch.autograd.Function):
    """The two operator calls of a bidirectional time-mix layer as ONE launch per pass (second half):
    problem 0 = the plain forward-direction call WKV_6.apply(r0, k0, v0, w0, u), problem 1 = the reversed-direction call
    WKV_6_REV.apply(r1, k1, v1, w1, u, rev_n, rev_mask) -- composition B (src/model_bi.py:331-348): r1 = r0, w1 = w0, k1 = k0,
    v1 = v0 with rev_mask = K | V | Y; composition C (src/model_ext.py:421-437): the reversed stream's own projections with
    rev_mask = ALL.  Returns (y0, y1), bit-identical to the two separate calls."""

    @staticmethod
    def forward(ctx, B, T, C, H, r0, k0, v0, w0, r1, k1, v1, w1, u, rev_n, rev_mask):
        with torch.no_grad():
            _assert_inputs(C, H, r0, k0, v0, w0, u)
            _assert_inputs(C, H, r1, k1, v1, w1, u)
            ctx.H, ctx.C, ctx.rev_mask = H, C, rev_mask
            ctx.save_for_backward(r0, k0, v0, w0, r1, k1, v1, w1, u, rev_n)
            ctx.ckpts = [wkv6_op.new_checkpoint(B, T, C, H, r0.device) if _keep_ckpt(ctx) else None for _ in range(2)]
            sets = [dict(r=r0, k=k0, v=v0, w=w0, ckpt=ctx.ckpts[0]),
                    dict(r=r1, k=k1, v=v1, w=w1, ckpt=ctx.ckpts[1], rev_n=rev_n, rev_mask=rev_mask)]
            return wkv6_op.forward_pair_ex(H, u, sets)

    @staticmethod
    def backward(ctx, gy0, gy1):
        with torch.no_grad():
            r0, k0, v0, w0, r1, k1, v1, w1, u, rev_n = ctx.saved_tensors
            shape = (ctx.H, ctx.C // ctx.H)
            if ctx.ckpts[0] is None:                       # RWKV_AMD_NO_CKPT=1: nothing was kept, each backward runs its own state pass
                g0 = wkv6_op.backward_ex(r0, k0, v0, w0, u, gy0.contiguous(), ctx.H)
                g1 = wkv6_op.backward_rev_ex(r1, k1, v1, w1, u, gy1.contiguous(), ctx.H, rev_n, ctx.rev_mask)
                return (None, None, None, None, *g0[:4], *g1[:4], _sum_bf16(g0[4], shape) + _sum_bf16(g1[4], shape), None, None)
            sets = [dict(r=r0, k=k0, v=v0, w=w0, gy=gy0.contiguous(), ckpt=ctx.ckpts[0]),
                    dict(r=r1, k=k1, v=v1, w=w1, gy=gy1.contiguous(), ckpt=ctx.ckpts[1], rev_n=rev_n, rev_mask=ctx.rev_mask)]
            g0, g1 = wkv6_op.backward_pair_ex(ctx.H, u, sets)
            ctx.ckpts = [None, None]                   # (a second backward through this node, retain_graph=True, runs self-contained)
            gu = _sum_bf16(g0[4], shape) + _sum_bf16(g1[4], shape)     # as autograd adds the two calls' bf16 gu
            return (None, None, None, None, *g0[:4], *g1[:4], gu, None, None)


class WKV_6_GN(torch.autograd.Function):
    """WKV_6 followed by the time-mix block's `ln_x` GroupNorm(H) and gate multiply (src/model.py:462-468) in ONE forward kernel
    (SURVEY.md 8f row n1): out = GroupNorm_H(WKV6(r,k,v,w,u); gamma, beta, eps) * g.  The operator's output y makes no round trip
    through HBM on the way to the normalisation; when no backward follows it is not written at all.  Backward = the GroupNorm /
    gate backward kernel (mix_op) feeding the operator's backward."""

    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u, g, gamma, beta, eps):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u)
            train = any(ctx.needs_input_grad)
            ckpt = wkv6_op.new_checkpoint(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            g = g.contiguous()
            res = wkv6_op.forward_gn_ex(r, k, v, w, u, H, g, gamma, beta, eps, ckpt=ckpt, want_y=train, want_stats=train)
            ctx.fused = res is not None
            if res is None:                                    # the library cannot fuse this shape: the two kernels
                from . import mix_op
                y = wkv6_op.forward_ex(r, k, v, w, u, H, ckpt=ckpt)
                out, stats = mix_op.gn_gate_forward(y.view(B * T, C), g.view(B * T, C), gamma, beta, H, eps)
                out = out.view(B, T, C)
            else:
                out, y, stats = res
            ctx.H, ctx.C, ctx.ckpt = H, C, ckpt
            if train:
                ctx.save_for_backward(r, k, v, w, u, g, gamma, beta, y, stats)
            return out

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            from . import mix_op
            r, k, v, w, u, g, gamma, beta, y, stats = ctx.saved_tensors
            B, T, C = r.shape
            dy, dg, dgamma, dbeta = mix_op.gn_gate_backward(y.view(B * T, C), g.view(B * T, C), gamma, beta, stats,
                                                            dout.contiguous().view(B * T, C), ctx.H)
            gr, gk, gv, gw, gu, _ = wkv6_op.backward_ex(r, k, v, w, u, dy.view(B, T, C), ctx.H, ckpt=ctx.ckpt)
            ctx.ckpt = None
            gu = _sum_bf16(gu, (ctx.H, ctx.C // ctx.H))
            return (None, None, None, None, gr, gk, gv, gw, gu, dg.view(B, T, C), dgamma, dbeta, None)


def RUN_CUDA_RWKV6_GN(B, T, C, H, r, k, v, w, u, g, gamma, beta, eps):
    return WKV_6_GN.apply(B, T, C, H, r, k, v, w, u, g, gamma, beta, eps)


def RUN_CUDA_RWKV6(B, T, C, H, r, k, v, w, u):
    return WKV_6.apply(B, T, C, H, r, k, v, w, u)


class WKV_6STATE(torch.autograd.Function):
    """RWKV_TRAIN_TYPE='states': learnable initial state s [H,N,N] shared by the batch."""

    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u, s):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u, s)
            ctx.B, ctx.T, ctx.C, ctx.H = B, T, C, H
            ctx.save_for_backward(r, k, v, w, u, s)
            ctx.ckpt = wkv6_op.new_checkpoint(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            return wkv6_op.forward_ex(r, k, v, w, u, H, s0=s, ckpt=ctx.ckpt)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            gy = gy.contiguous()
            r, k, v, w, u, s = ctx.saved_tensors
            H, N = ctx.H, ctx.C // ctx.H
            gr, gk, gv, gw, gu, gs = wkv6_op.backward_ex(r, k, v, w, u, gy, H, s0=s, want_gs=True, ckpt=ctx.ckpt)
            ctx.ckpt = None
            return (None, None, None, None, gr, gk, gv, gw, _sum_bf16(gu, (H, N)), _sum_bf16(gs, (H, N, N)))


class WKV_6STATE_INFCTX(torch.autograd.Function):
    """RWKV_TRAIN_TYPE='infctx': per-sample carried state s [B,H,N,N], overwritten with the final state."""

    @staticmethod
    def forward(ctx, B, T, C, H, r, k, v, w, u, s):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u, s)
            ctx.B, ctx.T, ctx.C, ctx.H = B, T, C, H
            s_init = s.clone()
            ctx.save_for_backward(r, k, v, w, u, s_init)
            ctx.ckpt = wkv6_op.new_checkpoint(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            # s <- final state, written through the raw pointer exactly like the reference kernel does
            return wkv6_op.forward_ex(r, k, v, w, u, H, s0=s_init, s_out=s, ckpt=ctx.ckpt)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            gy = gy.contiguous()
            r, k, v, w, u, s_init = ctx.saved_tensors
            H, N = ctx.H, ctx.C // ctx.H
            gr, gk, gv, gw, gu, gs = wkv6_op.backward_ex(r, k, v, w, u, gy, H, s0=s_init, want_gs=True, ckpt=ctx.ckpt)
            ctx.ckpt = None
            # the reference sums gs over the batch to [H,N,N] although s is per sample (src/model.py:126);
            # the per-sample gradient is the mathematically correct one for a per-sample state
            return (None, None, None, None, gr, gk, gv, gw, _sum_bf16(gu, (H, N)), gs.to(torch.bfloat16))


def RUN_CUDA_RWKV6_STATE(B, T, C, H, r, k, v, w, u, s):
    """'states' flavour (src/model.py:184-185): returns y."""
    return WKV_6STATE.apply(B, T, C, H, r, k, v, w, u, s)


def RUN_CUDA_RWKV6_INFCTX(B, T, C, H, r, k, v, w, u, s):
    """'infctx' flavour (src/model.py:130-132): returns (y, s) with s updated in place."""
    x = WKV_6STATE_INFCTX.apply(B, T, C, H, r, k, v, w, u, s)
    return x, s


class WKV_6_BI(torch.autograd.Function):
    @staticmethod
    def forward(ctx, B, T, C, H, mask, r, k, v, w, u):
        with torch.no_grad():
            _assert_inputs(C, H, r, k, v, w, u)
            assert mask.dtype == torch.int
            assert mask.is_contiguous()
            ctx.B, ctx.T, ctx.C, ctx.H = B, T, C, H
            ctx.mask = mask
            ctx.save_for_backward(r, k, v, w, u)
            # when a backward will follow, both scans leave their state checkpoints in a buffer kept on ctx: 16 B per
            # token-channel (8 per scan); the fp32 side buffers of the two calls are scratch of each call, not retained
            ctx.ws = wkv6_op.bi_new_kept(B, T, C, H, r.device) if _keep_ckpt(ctx) else None
            return wkv6_op.bi_forward_ex(mask, r, k, v, w, u, H, ws=ctx.ws)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            assert gy.dtype == torch.bfloat16
            gy = gy.contiguous()
            r, k, v, w, u = ctx.saved_tensors
            gr, gk, gv, gw, gu = wkv6_op.bi_backward_ex(ctx.mask, r, k, v, w, u, gy, ctx.H, ws=ctx.ws)
            ctx.ws = None
            return (None, None, None, None, None, gr, gk, gv, gw, _sum_bf16(gu, (ctx.H, ctx.C // ctx.H)))


def RUN_CUDA_RWKV6_BI(B, T, C, H, mask, r, k, v, w, u):
    """cuda/wkv6_bi.py:59-60 (there also called RUN_CUDA_RWKV6)."""
    return WKV_6_BI.apply(B, T, C, H, mask, r, k, v, w, u)


def select_for_train_type(train_type=None):
    """The reference picks one RUN_CUDA_RWKV6_STATE at import time from os.environ['RWKV_TRAIN_TYPE']
    (src/model.py:76, 133); missing variables default to '' here instead of raising KeyError (Q10)."""
    tt = os.environ.get("RWKV_TRAIN_TYPE", "") if train_type is None else train_type
    if tt == "infctx":
        return RUN_CUDA_RWKV6_INFCTX
    if tt == "states":
        return RUN_CUDA_RWKV6_STATE
    return RUN_CUDA_RWKV6


class RWKV_6(torch.autograd.Function):
    """Stateful forward-only WKV for inference (src/model_run.py:49-73): fp32 state [H,N,N] (B = 1) or [B,H,N,N],
    updated in place and returned; w is the raw decay parameter."""

    @staticmethod
    def forward(ctx, B, T, C, H, state, r, k, v, w, u):
        with torch.no_grad():
            assert HEAD_SIZE == C // H
            assert state.dtype == torch.float32
            for t in (r, k, v, w, u):
                assert t.is_contiguous()
            eew = torch.exp(-torch.exp(w.float())).contiguous()
            y = torch.empty((B, T, C), device=w.device, dtype=r.dtype)
            if r.dtype == torch.bfloat16:
                wkv6_op.rwkv6.forward_bf16(B, T, C, H, state, r, k, v, eew, u, y)
            elif r.dtype == torch.float32:
                wkv6_op.rwkv6.forward_fp32(B, T, C, H, state, r, k, v, eew, u, y)
            elif r.dtype == torch.float16:
                wkv6_op.rwkv6.forward_fp16(B, T, C, H, state, r, k, v, eew, u, y)
            else:
                raise RuntimeError(f"unsupported dtype {r.dtype}")
            return y, state


def RUN_RWKV_6(B, T, C, H, state, r, k, v, w, u):
    return RWKV_6.apply(B, T, C, H, state, r, k, v, w, u)
