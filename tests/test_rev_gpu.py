"""In-kernel sequence reversal on the GPU: the wkv6_*_rev_ex operator and the reversed-stream token
shift of the ddlerp kernel against what the reference writes with torch.gather (reverse_x_idx / reverse_x,
src/model_ext.py:410-419) around the SAME kernels -- the arithmetic is identical and only the addressing differs, so the
operator must agree bit for bit (forward and every gradient); then the two bidirectional compositions of the time-mix module
with in-kernel reversal against their gather formulation (bf16 GEMMs may order rows differently: 2 bf16 ulps of the tensor scale).
"""
import pytest
import torch

from oracle import caller_weights as cw
from rwkv_lm_ext_amd import callers

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


@pytest.fixture(scope="module")
def op():
    assert torch.cuda.is_available()
    from rwkv_lm_ext_amd import wkv6_op
    return wkv6_op


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf).cuda()


def rev_idx_of(rev_n, T):
    pos = torch.arange(T, device=rev_n.device).unsqueeze(0)
    n = rev_n.long().unsqueeze(1)
    return torch.where(pos < n, n - 1 - pos, pos)


def gather(x, idx, on):
    return callers.reverse_x(x, idx).contiguous() if on else x


@pytest.mark.parametrize("mask_name", ["K|V|Y", "ALL", "R|W", "Y", "none"])
def test_rev_operator_is_the_gather_formulation_bit_for_bit(op, mask_name):
    B, T, H = 4, 150, 2
    C = 64 * H
    bits = dict(R=op.REV_R, K=op.REV_K, V=op.REV_V, W=op.REV_W, Y=op.REV_Y)
    rev_mask = op.REV_ALL if mask_name == "ALL" else 0 if mask_name == "none" else sum(bits[c] for c in mask_name.split("|"))
    r, k, v = (rnd(B, T, C, scale=0.5, seed=s) for s in (1, 2, 3))
    w = (rnd(B, T, C, scale=0.7, seed=4).float() - 2.0).to(bf)
    u = rnd(H, 64, scale=0.3, seed=5)
    gy = rnd(B, T, C, seed=6)
    rev_n = torch.tensor([0, 1, 77, T], dtype=torch.int32, device="cuda")
    idx = rev_idx_of(rev_n, T)
    on = {c: bool(rev_mask & b) for c, b in bits.items()}

    ckpt = op.new_checkpoint(B, T, C, H, r.device)
    y = op.forward_rev_ex(r, k, v, w, u, H, rev_n, rev_mask, ckpt=ckpt)
    gr, gk, gv, gw, gu = op.backward_rev_ex(r, k, v, w, u, gy, H, rev_n, rev_mask, ckpt=ckpt)
    # the backward without the forward's checkpoints (own state pass) must give the same
    gr2, gk2, gv2, gw2, gu2 = op.backward_rev_ex(r, k, v, w, u, gy, H, rev_n, rev_mask)

    rg, kg, vg, wg = gather(r, idx, on["R"]), gather(k, idx, on["K"]), gather(v, idx, on["V"]), gather(w, idx, on["W"])
    y_ref = gather(op.forward_ex(rg, kg, vg, wg, u, H), idx, on["Y"])
    g_ref = op.backward_ex(rg, kg, vg, wg, u, gather(gy, idx, on["Y"]), H)
    want = [gather(g_ref[0], idx, on["R"]), gather(g_ref[1], idx, on["K"]), gather(g_ref[2], idx, on["V"]),
            gather(g_ref[3], idx, on["W"]), g_ref[4]]
    assert torch.equal(y, y_ref), mask_name
    for name, got, got2, ref in zip("gr gk gv gw gu".split(), (gr, gk, gv, gw, gu), (gr2, gk2, gv2, gw2, gu2), want):
        assert torch.equal(got, ref), (mask_name, name)
        assert torch.equal(got2, ref), (mask_name, name, "self-contained backward")


def test_rev_operator_rejects_bad_arguments(op):
    B, T, H = 2, 64, 1
    C = 64
    r = rnd(B, T, C)
    u = rnd(H, 64)
    with pytest.raises(RuntimeError):
        op.forward_rev_ex(r, r, r, r, u, H, torch.zeros(B, dtype=torch.int64, device="cuda"), op.REV_ALL)
    with pytest.raises(RuntimeError):
        op.forward_rev_ex(r, r, r, r, u, H, torch.zeros(B, dtype=torch.int32, device="cuda"), 64)
    with pytest.raises(RuntimeError):
        op.forward_rev_ex(r, r, r, r, u, H, torch.zeros(B + 1, dtype=torch.int32, device="cuda"), op.REV_ALL)


@pytest.mark.parametrize("ns", [1, 5])
def test_ddlerp_reversed_stream_shift(ns):
    from rwkv_lm_ext_amd import mix_op
    B, T, C = 5, 41, 128
    rev_n = torch.tensor([0, 1, 2, 23, T], dtype=torch.int32, device="cuda")
    idx = rev_idx_of(rev_n, T)
    x = rnd(B, T, C, seed=1).requires_grad_(True)
    maa = rnd(ns, C, scale=0.5, seed=2).requires_grad_(True)
    m = rnd(ns, B, T, C, scale=0.3, seed=3).requires_grad_(True) if ns == 5 else None
    dout = rnd(ns, B, T, C, seed=4)

    out = mix_op.ddlerp(x, maa, m, None, rev_n)
    out.backward(dout)
    got = [out.detach(), x.grad.clone(), maa.grad.clone()] + ([m.grad.clone()] if m is not None else [])
    x.grad = maa.grad = None

    # gather formulation: reverse x (and m), plain shift, un-reverse the result
    idx4 = idx.unsqueeze(0).expand(ns, -1, -1)
    g4 = lambda t: torch.gather(t, 2, idx4.unsqueeze(-1).expand(-1, -1, -1, C))
    m2 = None if m is None else g4(m.detach()).contiguous().requires_grad_(True)
    x2 = callers.reverse_x(x.detach(), idx).contiguous().requires_grad_(True)
    maa2 = maa.detach().clone().requires_grad_(True)
    out2 = mix_op.ddlerp(x2, maa2, m2, None, None)
    out2.backward(g4(dout).contiguous())
    want = [g4(out2.detach()), callers.reverse_x(x2.grad, idx), maa2.grad] + ([g4(m2.grad)] if m is not None else [])
    assert torch.equal(got[0], want[0])
    assert torch.equal(got[1], want[1])
    if m is not None:
        assert torch.equal(got[3], want[3])
    # parameter gradient: fp32 partial sums over rows dealt to workgroups in a different order, rounded once
    err = (got[2].float() - want[2].float()).abs().max() / want[2].float().abs().max()
    assert float(err) <= 1e-2


def _tmix():
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    return tm.cuda().to(bf)


def _close(got, ref, what, ulps=2.0):
    scale = float(ref.float().abs().max())
    d = float((got.float() - ref.float()).abs().max())
    assert d <= ulps * scale * 2.0 ** -8, f"{what}: max|d| = {d:.3e} at scale {scale:.3e}"


@pytest.mark.parametrize("comp", ["b", "c"])
def test_bidirectional_compositions_without_gathers(comp):
    """Tmix_x060.forward_bi_b / forward_bi_c with in-kernel reversal against the gather formulation (forced by handing the
    module a wrapped operator, which switches the in-kernel path off), outputs and gradients."""
    B, T = 3, 48
    x0 = rnd(B, T, cw.N_EMBD, seed=21)
    mask = torch.zeros(B, T, dtype=torch.int, device="cuda")
    for b, n in enumerate([T, 17, 1]):
        mask[b, :n] = 1
    rev_idx = callers.reverse_x_idx(mask, T)
    dout = rnd(B, T, cw.N_EMBD, seed=22)
    res = []
    for in_kernel in (True, False):
        tm = _tmix()
        if not in_kernel:
            inner = tm.wkv
            tm.wkv = lambda *a: inner(*a)
        assert tm._in_kernel_reversal(x0) == in_kernel
        x = x0.clone().requires_grad_(True)
        out = tm.forward_bi_b(x, mask) if comp == "b" else tm.forward_bi_c(x, rev_idx, mask)
        out.backward(dout)
        res.append((out.detach(), x.grad, tm.time_faaaa.grad, tm.key.weight.grad, tm.time_maa_k.grad, tm.time_decay.grad))
    names = "out dx d_time_faaaa d_key d_time_maa_k d_time_decay".split()
    for name, a, b in zip(names, *res):
        _close(a, b, f"composition {comp}: {name}", ulps=2.0 if name in ("out", "dx") else 4.0)
