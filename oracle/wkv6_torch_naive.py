"""Pure-PyTorch token-serial WKV6 -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

A restatement of the algorithm of the reference's pure-PyTorch CPU path
(fla/ops/rwkv6/recurrent_naive.py:8-36: per-token outer product, bonus term,
decayed state update, fp32) written directly on the operator's own [B,T,C]
layout with raw decay ``w`` (d = exp(-exp(w)), src/model.py:210 +
cuda/wkv6_cuda.cu:26) and the kernels' state layout s[.., value j, key i]
(cuda/wkv6state_cuda.cu:15).  Gradients come from autograd, exactly as the
reference's own self-check obtains them (recurrent_naive.py:113-119).

It is what ``bench.py`` times as ``cpu_baseline`` (kind "port"): the reference's
python cannot travel to the GPU box.  Pinned against the reference import by
oracle/gen_golden.py.
"""
import torch


def wkv6_naive(r, k, v, w, u, s0=None, return_state=False, dtype=torch.float32):
    """r,k,v,w: [B,T,C]; u: [H,N]; s0: None | [H,N,N] | [B,H,N,N] -> y [B,T,C]."""
    B, T, C = r.shape
    H, N = u.shape
    r4, k4, v4 = (x.to(dtype).view(B, T, H, N) for x in (r, k, v))
    decay = torch.exp(-torch.exp(w.to(dtype))).view(B, T, H, N)
    bonus = u.to(dtype).view(1, H, N, 1)
    if s0 is None:
        S = torch.zeros(B, H, N, N, dtype=dtype, device=r.device)       # [b,h,key i,value j]
    else:
        S = s0.to(dtype).transpose(-1, -2)
        if S.dim() == 3:
            S = S.unsqueeze(0).expand(B, H, N, N)
    ys = []
    for t in range(T):
        kv = k4[:, t].unsqueeze(-1) * v4[:, t].unsqueeze(-2)               # [B,H,i,j]
        ys.append(torch.einsum("bhi,bhij->bhj", r4[:, t], S + bonus * kv))
        S = decay[:, t].unsqueeze(-1) * S + kv
    y = torch.stack(ys, 1).reshape(B, T, C)
    if return_state:
        return y, S.transpose(-1, -2).contiguous()
    return y


def wkv6_naive_fwd_bwd(r, k, v, w, u, gy, s0=None, dtype=torch.float32):
    """Forward + autograd backward; returns (y, dict of grads)."""
    leaves = [x.detach().to(dtype).clone().requires_grad_(True) for x in (r, k, v, w, u)]
    s = None if s0 is None else s0.detach().to(dtype).clone().requires_grad_(True)
    y = wkv6_naive(*leaves, s0=s, dtype=dtype)
    y.backward(gy.to(dtype))
    g = {n: (x.grad if x.grad is not None else torch.zeros_like(x))
         for n, x in zip(("gr", "gk", "gv", "gw", "gu"), leaves)}
    if s is not None:
        g["gs"] = s.grad
    return y.detach(), g
