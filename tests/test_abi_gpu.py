"""Every reference-signature entry point on the GPU, with the reference's dtypes, against the oracle: the four module objects
(`wkv6_cuda`, `wkv6_bi_cuda`, `wkv6state_cuda`, `wkv6infctx_cuda`: cuda/wkv6_op.cpp:8-13, wkv6_bi_op.cpp:8-13,
wkv6state_op.cpp:8-13, wkv6infctx_op.cpp:8-13), their `torch.ops.*` registrations, and the same calls through the compiled C++
shim, whose functions call the plain `*_cuda_*` C symbols: no workspace argument, so each of them runs the library's
stream-ordered scratch allocation, and `gu` / `gs` come back as the reference's bf16 per-batch partials.

Tolerances: gradients at the suite's bf16 contract (test_wkv6_gpu.check); gu / gs after the reference's own reduction -- a bf16
`torch.sum` over the batch of bf16 partials (src/model.py:181, 232) -- max-normalised <= 8e-3 (two bf16 roundings)."""
import numpy as np
import pytest
import torch

from conftest import max_norm_err
from test_wkv6_gpu import check, dev, host, rand_inputs

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
PART_BF16 = 8e-3


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from rwkv_lm_ext_amd import wkv6_op
    return wkv6_op


@pytest.fixture(scope="module")
def shim():
    from rwkv_lm_ext_amd import torch_shim
    return torch_shim.load(prefix="shim")


def _case(seed, B, T, H, with_mask=False):
    r, k, v, w, u, gy = rand_inputs(seed, B, T, H, "stress")
    d = dict(r=r, k=k, v=v, w=w, u=u, gy=gy)
    t = {n: dev(a, BF) for n, a in d.items()}
    t["ew"] = (-torch.exp(t["w"].float())).contiguous()               # src/model.py:210: the wkv6 / wkv6_bi ops take fp32 -exp(w)
    if with_mask:
        mask = torch.ones(B, T, dtype=torch.int32)
        for b, cut in enumerate([T, T // 3, 1][:B]):
            mask[b, cut:] = 0
        d["mask"], t["mask"] = mask.numpy(), mask.cuda()
    return d, t


def _empties(B, T, C, n=4):
    return [torch.empty(B, T, C, device="cuda", dtype=BF) for _ in range(n)]


def _check_grads(outs, ref, what):
    for t, n in zip(outs, ("gr", "gk", "gv", "gw")):
        check(t, ref[n], BF, f"{what} {n}")


def _gu_ok(gu, ref_gu, H):
    got = torch.sum(gu, 0).view(H, -1)                                 # the caller's reduction, in bf16 (src/model.py:181)
    assert gu.dtype == BF and max_norm_err(host(got), ref_gu) <= PART_BF16


@pytest.mark.parametrize("via", ["module", "torch.ops", "shim", "shim torch.ops"])
def test_wkv6_and_wkv6_bi(ops, shim, oracle, via):
    B, T, H = 3, 70, 2
    C = H * 64
    d, t = _case(11, B, T, H, with_mask=True)
    fw, bw, bfw, bbw = {
        "module": (ops.wkv6_cuda.forward, ops.wkv6_cuda.backward, ops.wkv6_bi_cuda.forward, ops.wkv6_bi_cuda.backward),
        "torch.ops": (torch.ops.wkv6.forward, torch.ops.wkv6.backward, torch.ops.wkv6bi.forward, torch.ops.wkv6bi.backward),
        "shim": (shim.wkv6.forward, shim.wkv6.backward, shim.wkv6_bi.forward, shim.wkv6_bi.backward),
        "shim torch.ops": (torch.ops.shim_wkv6.forward, torch.ops.shim_wkv6.backward, torch.ops.shim_wkv6bi.forward,
                           torch.ops.shim_wkv6bi.backward),
    }[via]
    a = (B, T, C, H)
    # wkv6
    (y,) = _empties(B, T, C, 1)
    fw(*a, t["r"], t["k"], t["v"], t["ew"], t["u"], y)
    check(y, oracle.forward(d["r"], d["k"], d["v"], d["w"], d["u"]), BF, f"wkv6 forward ({via})")
    outs, gu = _empties(B, T, C), torch.empty(B, C, device="cuda", dtype=BF)
    bw(*a, t["r"], t["k"], t["v"], t["ew"], t["u"], t["gy"], *outs, gu)
    og = oracle.backward(d["r"], d["k"], d["v"], d["w"], d["u"], d["gy"])
    _check_grads(outs, og, f"wkv6 backward ({via})")
    _gu_ok(gu, og["gu"], H)
    # wkv6_bi: int32 mask after H
    fill = torch.full((B, T, C), 7.0, device="cuda", dtype=BF)
    y = fill.clone()
    bfw(*a, t["mask"], t["r"], t["k"], t["v"], t["ew"], t["u"], y)
    check(y, oracle.bi_forward(d["mask"], d["r"], d["k"], d["v"], d["w"], d["u"]), BF, f"wkv6_bi forward ({via})")
    assert bool((y[2, 2:] == 0).all())                                 # Q2: zero-filled behind the first masked token (index 1, itself processed)
    outs, gu = [fill.clone() for _ in range(4)], torch.empty(B, C, device="cuda", dtype=BF)
    bbw(*a, t["mask"], t["r"], t["k"], t["v"], t["ew"], t["u"], t["gy"], *outs, gu)
    og = oracle.bi_backward(d["mask"], d["r"], d["k"], d["v"], d["w"], d["u"], d["gy"])
    _check_grads(outs, og, f"wkv6_bi backward ({via})")
    _gu_ok(gu, og["gu"], H)


@pytest.mark.parametrize("via", ["module", "torch.ops", "shim", "shim torch.ops"])
@pytest.mark.parametrize("flavour", ["wkv6state", "wkv6infctx"])
def test_state_flavours(ops, shim, oracle, flavour, via):
    B, T, H = 2, 90, 2
    C = H * 64
    d, t = _case(12, B, T, H)
    g = torch.Generator().manual_seed(5)
    per_batch = flavour == "wkv6infctx"
    s = (torch.randn(*((B,) if per_batch else ()), H, 64, 64, generator=g) * 0.5).to(BF)
    sd, sn = s.cuda().contiguous(), s.float().numpy()
    mod = {"module": getattr(ops, flavour + "_cuda"), "torch.ops": getattr(torch.ops, flavour),
           "shim": getattr(shim, flavour), "shim torch.ops": getattr(torch.ops, "shim_" + flavour)}[via]
    a = (B, T, C, H)
    (y,) = _empties(B, T, C, 1)
    s_in = sd.clone()
    mod.forward(*a, t["r"], t["k"], t["v"], t["w"], t["u"], s_in, y)          # raw bf16 decay (cuda/wkv6state_op.cpp:8-10)
    yo, so = oracle.forward(d["r"], d["k"], d["v"], d["w"], d["u"], sn, return_state=True)
    check(y, yo, BF, f"{flavour} forward ({via})")
    if per_batch:                                                              # infctx: s is overwritten with the final state
        check(s_in, so, BF, f"{flavour} final state ({via})")
    outs, gu = _empties(B, T, C), torch.empty(B, C, device="cuda", dtype=BF)
    gs = torch.empty(B, H, 64, 64, device="cuda", dtype=BF)
    mod.backward(*a, t["r"], t["k"], t["v"], t["w"], t["u"], sd, t["gy"], *outs, gu, gs)
    og = oracle.backward(d["r"], d["k"], d["v"], d["w"], d["u"], d["gy"], sn)
    _check_grads(outs, og, f"{flavour} backward ({via})")
    _gu_ok(gu, og["gu"], H)
    assert gs.dtype == BF
    if per_batch:
        assert max_norm_err(host(gs), og["gs_b"]) <= PART_BF16
    else:
        assert max_norm_err(host(torch.sum(gs, 0)), og["gs"]) <= PART_BF16     # src/model.py:181


def test_shim_rwkv6_every_flavour(shim, oracle):
    """cuda/rwkv6_op.cpp:12-23 through the C++ shim: forward_bf16, forward_fp16, forward_fp32 (fp32 state, decay exp(-exp(w)))."""
    B, T, H = 2, 40, 2
    C = H * 64
    g = torch.Generator().manual_seed(6)
    w = -1.0 + 0.5 * torch.randn(B, T, C, generator=g)
    eew = torch.exp(-torch.exp(w.cuda())).contiguous()
    base = [torch.randn(B, T, C, generator=g) * 0.5 for _ in range(3)] + [torch.randn(H, 64, generator=g) * 0.3]
    for dt, fn, tol in ((BF, shim.rwkv6.forward_bf16, 8e-3), (torch.float16, shim.rwkv6.forward_fp16, 1e-3),
                        (torch.float32, shim.rwkv6.forward_fp32, 1e-5), (torch.float16, torch.ops.shim_rwkv6.forward_fp16, 1e-3)):
        r, k, v, u = (x.to(dt) for x in base)
        yo, so = oracle.forward(r.float().numpy(), k.float().numpy(), v.float().numpy(), w.numpy(), u.float().numpy(),
                                np.zeros((B, H, 64, 64), np.float32), return_state=True)
        st = torch.zeros(B, H, 64, 64, device="cuda")
        y = torch.empty(B, T, C, device="cuda", dtype=dt)
        fn(B, T, C, H, st, r.cuda(), k.cuda(), v.cuda(), eew, u.cuda(), y)
        assert max_norm_err(host(y), yo) <= tol, dt
        assert max_norm_err(st.cpu().numpy(), so) <= (2e-4 if dt == BF else 1e-5), dt


def test_shim_rejects_what_the_c_abi_cannot_take(shim):
    B, T, H = 1, 8, 1
    C = 64
    z = lambda *s, dt=BF: torch.zeros(*s, device="cuda", dtype=dt)
    with pytest.raises(RuntimeError):                                          # mask of the wrong size (backward too)
        shim.wkv6_bi.backward(B, T, C, H, torch.ones(B, T + 1, dtype=torch.int32, device="cuda"), z(B, T, C), z(B, T, C), z(B, T, C),
                              z(B, T, C, dt=torch.float32), z(H, 64), z(B, T, C), z(B, T, C), z(B, T, C), z(B, T, C), z(B, T, C), z(B, C))
    with pytest.raises(RuntimeError):                                          # sizes that do not fit the ABI's int
        shim.wkv6.forward(2 ** 31, T, C, H, z(B, T, C), z(B, T, C), z(B, T, C), z(B, T, C, dt=torch.float32), z(H, 64), z(B, T, C))
