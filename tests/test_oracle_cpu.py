"""CPU suite: the oracle against the reference-generated golden vectors, host-side logic, and the C ABI
(load + exported symbols + argument rejection; no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, max_norm_err

PLAIN = ["wkv6_init", "wkv6_stress", "wkv6_extreme", "wkv6_T1", "wkv6_T2", "wkv6_T3"]
TOL = 2e-5   # golden values are the reference's fp32 results; the oracle computes in fp64


@pytest.mark.parametrize("name", PLAIN)
def test_oracle_matches_reference_plain(oracle, name):
    g = load_golden(name)
    y = oracle.forward(g["r"], g["k"], g["v"], g["w"], g["u"])
    assert max_norm_err(y, g["y"]) <= TOL
    og = oracle.backward(g["r"], g["k"], g["v"], g["w"], g["u"], g["gy"])
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(og[n], g[n]) <= TOL, n


def test_oracle_matches_reference_state(oracle):
    g = load_golden("wkv6_state")
    assert max_norm_err(oracle.forward(g["r"], g["k"], g["v"], g["w"], g["u"], g["s"]), g["y"]) <= TOL
    og = oracle.backward(g["r"], g["k"], g["v"], g["w"], g["u"], g["gy"], g["s"])
    for n in ("gr", "gk", "gv", "gw", "gu", "gs"):
        assert max_norm_err(og[n], g[n]) <= TOL, n


def test_oracle_matches_reference_infctx(oracle):
    g = load_golden("wkv6_infctx")
    y, sf = oracle.forward(g["r"], g["k"], g["v"], g["w"], g["u"], g["s"], return_state=True)
    assert max_norm_err(y, g["y"]) <= TOL
    assert max_norm_err(sf, g["s_final"]) <= TOL
    og = oracle.backward(g["r"], g["k"], g["v"], g["w"], g["u"], g["gy"], g["s"])
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(og[n], g[n]) <= TOL, n
    assert max_norm_err(og["gs_b"], g["gs"]) <= TOL
    # carried state: three chunks of 16 reproduce the single 48-token call
    s = g["s"]
    ys = []
    for c in range(3):
        sl = slice(16 * c, 16 * c + 16)
        yc, s = oracle.forward(g["r"][:, sl], g["k"][:, sl], g["v"][:, sl], g["w"][:, sl], g["u"], s,
                               return_state=True)
        ys.append(yc)
    assert max_norm_err(np.concatenate(ys, 1), g["y"]) <= TOL
    assert max_norm_err(s, g["s_final"]) <= TOL


def test_oracle_matches_reference_bi(oracle):
    g = load_golden("wkv6_bi")
    y = oracle.bi_forward(g["mask"], g["r"], g["k"], g["v"], g["w"], g["u"])
    assert max_norm_err(y, g["y"]) <= TOL
    assert np.all(y[0, 31:] == 0) and np.all(y[1, 18:] == 0) and np.all(y[2, 1:] == 0)
    og = oracle.bi_backward(g["mask"], g["r"], g["k"], g["v"], g["w"], g["u"], g["gy"])
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(og[n], g[n]) <= TOL, n


def test_oracle_backward_is_gradient_of_forward(oracle):
    """Finite-difference check of the hand-written adjoint (independent of the reference)."""
    rng = np.random.default_rng(7)
    B, T, H, N = 1, 5, 1, 8
    C = H * N
    r, k, v = (rng.standard_normal((B, T, C)).astype(np.float32) * 0.5 for _ in range(3))
    w = (-1 + 0.5 * rng.standard_normal((B, T, C))).astype(np.float32)
    u = rng.standard_normal((H, N)).astype(np.float32) * 0.3
    s0 = rng.standard_normal((H, N, N)).astype(np.float32) * 0.5
    gy = rng.standard_normal((B, T, C)).astype(np.float32)
    g = oracle.backward(r, k, v, w, u, gy, s0)
    loss = lambda **kw: float((oracle.forward(kw.get("r", r), kw.get("k", k), kw.get("v", v), kw.get("w", w),
                                              kw.get("u", u), kw.get("s0", s0)).astype(np.float64) * gy).sum())
    eps = 1e-2
    for name, arr, grad in (("r", r, g["gr"]), ("k", k, g["gk"]), ("v", v, g["gv"]), ("w", w, g["gw"]),
                            ("u", u, g["gu"]), ("s0", s0, g["gs"])):
        idx = tuple(rng.integers(0, s) for s in arr.shape)
        hi, lo = arr.copy(), arr.copy()
        hi[idx] += eps
        lo[idx] -= eps
        fd = (loss(**{name: hi}) - loss(**{name: lo})) / (2 * eps)
        assert abs(fd - grad[idx]) <= 2e-3 * max(1.0, abs(fd)), (name, fd, grad[idx])


def test_torch_port_matches_reference():
    """The cpu_baseline implementation (oracle/wkv6_torch_naive.py) reproduces the golden vectors."""
    from oracle.wkv6_torch_naive import wkv6_naive, wkv6_naive_fwd_bwd
    g = load_golden("wkv6_stress")
    t = {k_: torch.from_numpy(v_) for k_, v_ in g.items()}
    y, gr = wkv6_naive_fwd_bwd(t["r"], t["k"], t["v"], t["w"], t["u"], t["gy"])
    assert max_norm_err(y, g["y"]) <= TOL
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(gr[n], g[n]) <= TOL, n
    g = load_golden("wkv6_infctx")
    t = {k_: torch.from_numpy(v_) for k_, v_ in g.items()}
    y, sf = wkv6_naive(t["r"], t["k"], t["v"], t["w"], t["u"], s0=t["s"], return_state=True)
    assert max_norm_err(y, g["y"]) <= TOL and max_norm_err(sf, g["s_final"]) <= TOL


# ---- C ABI ------------------------------------------------------------------------------------------
def _header_symbols():
    text = open(os.path.join(ROOT, "include", "wkv6_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(r?wkv6\w*)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rwkv_lm_ext_amd import _lib
    lib = ctypes.CDLL(_lib._build.build())
    syms = _header_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/wkv6_amd.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), "python binding table and header disagree"


def test_abi_rejects_bad_arguments_without_launching():
    from rwkv_lm_ext_amd import _lib
    lib = _lib.load()
    assert lib.wkv6_amd_version() == b"0.1"
    # C != H*64  (reference: assert(H*_N_ == C), cuda/wkv6_cuda.cu:231)
    assert lib.wkv6_cuda_forward(1, 4, 100, 1, 1, 1, 1, 1, 1, 1, None) == -1
    assert lib.wkv6_cuda_forward(0, 4, 64, 1, 1, 1, 1, 1, 1, 1, None) == -1
    assert lib.wkv6_cuda_forward(1, 4, 64, 1, None, 1, 1, 1, 1, 1, None) == -2
    assert lib.wkv6state_cuda_forward(1, 4, 64, 1, 1, 1, 1, 1, 1, None, 1, None) == -2
    assert lib.wkv6_backward_ex(1, 4, 64, 1, 1, 1, 1, 1, 1, None, 1, 1, 1, 1, 1, None, None, 1, 16, 1, None) == -3
    # max(scan scratch fp32 [B,T,C], one 64x64 fp32 forward state per (batch, head) and 64 tokens) -- equal: 4 B per token-channel
    assert lib.wkv6_backward_workspace_bytes(8, 4096, 2048, 32) == 8 * 32 * (4096 // 64) * 64 * 64 * 4
    assert lib.wkv6_backward_workspace_bytes(1, 16, 64, 1) == max(1 * 16 * 64 * 4, 64 * 64 * 4)
    # lens | 2 scan workspaces (one per direction) | 4 fp32 [B,T,C] side buffers, each 256-byte aligned
    assert lib.wkv6bi_workspace_bytes(2, 16, 64, 1) == 256 + 2 * (2 * 1 * 64 * 64 * 4) + 4 * (2 * 16 * 64 * 4)
    # one sequence must stay 32-bit addressable (T*C < 2^31): refused before anything is dereferenced
    assert lib.wkv6_cuda_forward(1, 1 << 20, 2048, 32, 1, 1, 1, 1, 1, 1, None) == -4


def test_python_operator_rejects_cpu_and_wrong_dtype_tensors():
    from rwkv_lm_ext_amd import wkv6_op
    from rwkv_lm_ext_amd.wkv import WKV_6, WKV_6_BI
    B, T, C, H = 1, 4, 64, 1
    x = torch.zeros(B, T, C, dtype=torch.bfloat16)
    u = torch.zeros(H, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="GPU"):
        wkv6_op.wkv6_cuda.forward(B, T, C, H, x, x, x, x.float(), u, x.clone())
    with pytest.raises(AssertionError):                       # reference: assert r.dtype == torch.bfloat16
        WKV_6.apply(B, T, C, H, x.float(), x, x, x, u)
    with pytest.raises(AssertionError):                       # reference: assert r.is_contiguous()
        WKV_6.apply(B, T, C, H, torch.zeros(B, C, T, dtype=torch.bfloat16).transpose(1, 2), x, x, x, u)
    with pytest.raises(AssertionError):                       # cuda/wkv6_bi.py:22
        WKV_6_BI.apply(B, T, C, H, torch.ones(B, T, dtype=torch.int64), x, x, x, x, u)
    assert hasattr(torch.ops.wkv6, "forward") and hasattr(torch.ops.wkv6bi, "backward")
    assert hasattr(torch.ops.wkv6state, "forward") and hasattr(torch.ops.wkv6infctx, "backward")


def test_python_constants_match_the_header():
    """The ctypes layer mirrors the header's enums by value: keep them from drifting apart."""
    import os
    import re
    from rwkv_lm_ext_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "wkv6_amd.h")) as f:
        text = f.read()
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(WKV6_[A-Z0-9_]+)\s*=\s*(-?\d+)", text)}
    pairs = dict(WKV6_W_EW_F32=_lib.W_EW_F32, WKV6_W_RAW=_lib.W_RAW, WKV6_IO_F32=_lib.IO_F32, WKV6_S0_PER_BATCH=_lib.S0_PER_BATCH,
                 WKV6_ALGO_SCAN=_lib.ALGO_SCAN, WKV6_CKPT_VALID=_lib.CKPT_VALID, WKV6_BI_KEEP_CKPT=_lib.BI_KEEP_CKPT,
                 WKV6_PARTIALS_F32=_lib.PARTIALS_F32, WKV6_REV_R=_lib.REV_R, WKV6_REV_K=_lib.REV_K, WKV6_REV_V=_lib.REV_V,
                 WKV6_REV_W=_lib.REV_W, WKV6_REV_Y=_lib.REV_Y)
    for name, value in pairs.items():
        assert enums[name] == value, name
    assert _lib.REV_ALL == _lib.REV_R | _lib.REV_K | _lib.REV_V | _lib.REV_W | _lib.REV_Y


def test_header_is_plain_c_and_the_pair_struct_matches_ctypes(tmp_path):
    """include/wkv6_amd.h compiles as C (gcc -std=c99 -pedantic), and wkv6_seq_set has the layout the ctypes mirror assumes."""
    import subprocess
    from rwkv_lm_ext_amd import _lib
    fields = [n for n, _ in _lib.SeqSet._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "wkv6_amd.h"\nint main(void) {\n'
                   '    printf("%zu\\n", sizeof(wkv6_seq_set));\n'
                   + "".join(f'    printf("%zu\\n", offsetof(wkv6_seq_set, {n}));\n' for n in fields)
                   + "    return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(_lib.SeqSet)
    assert out[1:] == [getattr(_lib.SeqSet, n).offset for n in fields]
    # no-launch argument checks of the pair entry points
    lib = _lib.load()
    sets = (_lib.SeqSet * 2)()
    assert lib.wkv6_forward_pair_ex(1, 4, 64, 1, None, sets, _lib.W_RAW, None) == -2          # u missing
    assert lib.wkv6_forward_pair_ex(1, 4, 64, 1, 1, sets, _lib.W_RAW, None) == -2             # tensors missing
    assert lib.wkv6_forward_pair_ex(1, 4, 64, 1, 1, sets, _lib.W_RAW | _lib.IO_F32, None) == -4
    assert lib.wkv6_backward_pair_ex(1, 4, 100, 1, 1, sets, _lib.W_RAW, None) == -1


def test_oracle_matches_the_medium_reference_fixtures(oracle):
    """T = 160 vectors generated from the reference's CPU recurrence (oracle/gen_golden_medium.py): plain, per-sample state
    (gs, final state), ragged wkv6_bi."""
    from conftest import load_golden_mid
    g = load_golden_mid("wkv6_mid")
    a = [g[n] for n in ("r", "k", "v", "w", "u")]
    assert max_norm_err(oracle.forward(*a), g["y"]) <= 2e-5
    og = oracle.backward(*a, g["gy"])
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(og[n], g[n]) <= 2e-5, n
    g = load_golden_mid("wkv6_mid_state")
    a = [g[n] for n in ("r", "k", "v", "w", "u")]
    y, so = oracle.forward(*a, g["s"], return_state=True)
    assert max_norm_err(y, g["y"]) <= 2e-5 and max_norm_err(so, g["s_final"]) <= 2e-5
    og = oracle.backward(*a, g["gy"], g["s"])
    for n, m in (("gr", "gr"), ("gk", "gk"), ("gv", "gv"), ("gw", "gw"), ("gu", "gu"), ("gs_b", "gs")):
        assert max_norm_err(og[n], g[m]) <= 2e-5, n
    g = load_golden_mid("wkv6_mid_bi")
    a = [g[n] for n in ("r", "k", "v", "w", "u")]
    mask = g["mask"].astype(np.int32)
    assert max_norm_err(oracle.bi_forward(mask, *a), g["y"]) <= 2e-5
    ob = oracle.bi_backward(mask, *a, g["gy"])
    for n in ("gr", "gk", "gv", "gw", "gu"):
        assert max_norm_err(ob[n], g[n]) <= 2e-5, n


def test_clock_ring_host_bookkeeping():
    """wkv6_set_clock_ring / wkv6_clock_ring_counts (include/wkv6_amd.h, measurement aids): setting the ring resets the per-kernel launch
    counts, nothing is launched or dereferenced on the host, NULL switches the probe off; the python wrapper's no-op contract for A/B
    libraries that predate the symbols rests on _lib.has_symbol."""
    import ctypes
    from rwkv_lm_ext_amd import _lib
    lib = _lib.load()
    assert all(_lib.has_symbol(s) for s in _lib.MEASUREMENT_AIDS)
    f, b = ctypes.c_long(-1), ctypes.c_long(-1)
    lib.wkv6_set_clock_ring(ctypes.c_void_p(0x1000), 64, 128)      # (a device address as far as the library knows: only handed to kernels)
    lib.wkv6_clock_ring_counts(ctypes.byref(f), ctypes.byref(b))
    assert (f.value, b.value) == (0, 0)
    lib.wkv6_set_clock_ring(None, 0, 0)
    lib.wkv6_set_clock_buffer(None, 0)
    lib.wkv6_clock_ring_counts(ctypes.byref(f), None)
    assert f.value == 0
