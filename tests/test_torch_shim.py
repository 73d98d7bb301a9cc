"""INTEGRATION.md level 2: the C++ torch-extension shim (csrc/torch_shim/wkv6_torch_shim.cpp) compiles against this
torch + ROCm, links librwkv6_amd.so, exposes the reference's pybind functions and registers the TORCH_LIBRARY operators
(cuda/wkv6_op.cpp:8-22, wkv6_bi_op.cpp, wkv6state_op.cpp, wkv6infctx_op.cpp, rwkv6_op.cpp).  The CPU test builds and
inspects it; the GPU test calls through it and compares with the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import max_norm_err



@pytest.fixture(scope="module")
def shim():
    from rwkv_lm_ext_amd import torch_shim
    return torch_shim.load(prefix="shim")


def test_shim_builds_and_registers_every_operator(shim):
    for sub, fns in (("wkv6", ("forward", "backward")), ("wkv6_bi", ("forward", "backward")),
                     ("wkv6state", ("forward", "backward")), ("wkv6infctx", ("forward", "backward")),
                     ("rwkv6", ("forward_bf16", "forward_fp16", "forward_fp32"))):
        for fn in fns:
            assert callable(getattr(getattr(shim, sub), fn))
    for ns, fns in (("shim_wkv6", ("forward", "backward")), ("shim_wkv6bi", ("forward", "backward")),
                    ("shim_wkv6state", ("forward", "backward")), ("shim_wkv6infctx", ("forward", "backward")),
                    ("shim_rwkv6", ("forward_bf16", "forward_fp16", "forward_fp32"))):
        for fn in fns:
            assert getattr(getattr(torch.ops, ns), fn) is not None
    # stricter than the reference shim: CPU tensors are rejected before anything is launched
    B, T, C, H = 1, 4, 64, 1
    t = torch.zeros(B, T, C, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        shim.wkv6.forward(B, T, C, H, t, t, t, t.float(), torch.zeros(H, 64, dtype=torch.bfloat16), t.clone())


@pytest.mark.gpu
def test_shim_calls_match_oracle(shim, oracle):
    bf = torch.bfloat16
    B, T, H = 2, 80, 2
    C = H * 64
    g = torch.Generator().manual_seed(3)
    f = lambda x: x.to(bf).float().numpy()
    r, k, v = (f(torch.randn(B, T, C, generator=g) * 0.5) for _ in range(3))
    w = f(-1.0 + 0.5 * torch.randn(B, T, C, generator=g))
    u = f(torch.randn(H, 64, generator=g) * 0.3)
    gy = f(torch.randn(B, T, C, generator=g))
    d = lambda x, dt=bf: torch.from_numpy(x).to("cuda", dt).contiguous()
    rd, kd, vd, ud, gyd = d(r), d(k), d(v), d(u), d(gy)
    ew = (-torch.exp(d(w).float())).contiguous()                      # src/model.py:210
    y = torch.empty(B, T, C, device="cuda", dtype=bf)
    shim.wkv6.forward(B, T, C, H, rd, kd, vd, ew, ud, y)
    assert max_norm_err(y.float().cpu().numpy(), oracle.forward(r, k, v, w, u)) <= 8e-3
    outs = [torch.empty(B, T, C, device="cuda", dtype=bf) for _ in range(4)]
    gu = torch.empty(B, C, device="cuda", dtype=bf)
    torch.ops.shim_wkv6.backward(B, T, C, H, rd, kd, vd, ew, ud, gyd, *outs, gu)
    og = oracle.backward(r, k, v, w, u, gy)
    for t, n in zip(outs, ("gr", "gk", "gv", "gw")):
        assert max_norm_err(t.float().cpu().numpy(), og[n]) <= 8e-3, n
    # state flavour through the shim: raw bf16 decay
    s = f(torch.randn(H, 64, 64, generator=g) * 0.5)
    shim.wkv6state.forward(B, T, C, H, rd, kd, vd, d(w), ud, d(s), y)
    assert max_norm_err(y.float().cpu().numpy(), oracle.forward(r, k, v, w, u, s)) <= 8e-3
    # inference kernel: fp32 state, decay already exponentiated
    st = torch.zeros(B, H, 64, 64, device="cuda")
    eew = torch.exp(-torch.exp(d(w).float())).contiguous()
    shim.rwkv6.forward_bf16(B, T, C, H, st, rd, kd, vd, eew, ud, y)
    yo, so = oracle.forward(r, k, v, w, u, np.zeros((B, H, 64, 64), np.float32), return_state=True)
    assert max_norm_err(y.float().cpu().numpy(), yo) <= 8e-3
    assert max_norm_err(st.cpu().numpy(), so) <= 2e-4


@pytest.mark.gpu
def test_rwkv6_fp16_flavour_and_torch_ops(oracle):
    """cuda/rwkv6_op.cpp:16-19, 30-34: forward_fp16 and the rwkv6 TORCH_LIBRARY names."""
    from rwkv_lm_ext_amd import wkv6_op                                # noqa: F401  (registers torch.ops.rwkv6)
    B, T, H = 1, 40, 2
    C = H * 64
    g = torch.Generator().manual_seed(5)
    h = lambda x: x.to(torch.float16)
    r, k, v = (h(torch.randn(B, T, C, generator=g) * 0.5) for _ in range(3))
    w = -1.0 + 0.5 * torch.randn(B, T, C, generator=g)
    u = h(torch.randn(H, 64, generator=g) * 0.3)
    yo, so = oracle.forward(r.float().numpy(), k.float().numpy(), v.float().numpy(), w.numpy(), u.float().numpy(),
                            np.zeros((H, 64, 64), np.float32), return_state=True)
    st = torch.zeros(H, 64, 64, device="cuda")
    y = torch.empty(B, T, C, device="cuda", dtype=torch.float16)
    eew = torch.exp(-torch.exp(w.cuda())).contiguous()
    torch.ops.rwkv6.forward_fp16(B, T, C, H, st, r.cuda(), k.cuda(), v.cuda(), eew, u.cuda(), y)
    assert max_norm_err(y.float().cpu().numpy(), yo) <= 1e-3          # fp16 output rounding
    assert max_norm_err(st.cpu().numpy(), so[0]) <= 1e-5
