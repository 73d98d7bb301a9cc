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


@pytest.mark.parametrize("io", ["bf16 scan", "fp32"])
@pytest.mark.parametrize("mask_name", ["K|V|Y", "ALL", "R|W"])
def test_rev_in_the_exact_scan_kernels(op, io, mask_name):
    """The same index maps in the token-serial fp32 kernels (fp32 I/O, or bf16 I/O with algo="scan"): bit for bit the gather
    formulation around the same kernels, forward and every gradient."""
    B, T, H = 3, 45, 2
    C = 64 * H
    bits = dict(R=op.REV_R, K=op.REV_K, V=op.REV_V, W=op.REV_W, Y=op.REV_Y)
    rev_mask = op.REV_ALL if mask_name == "ALL" else sum(bits[c] for c in mask_name.split("|"))
    dt = torch.float32 if io == "fp32" else bf
    algo = None if io == "fp32" else "scan"
    r, k, v = (rnd(B, T, C, scale=0.5, seed=s).to(dt) for s in (11, 12, 13))
    w = (rnd(B, T, C, scale=0.7, seed=14).float() - 2.0).to(dt)
    u = rnd(H, 64, scale=0.3, seed=15).to(dt)
    gy = rnd(B, T, C, seed=16).to(dt)
    rev_n = torch.tensor([1, 30, T], dtype=torch.int32, device="cuda")
    idx = rev_idx_of(rev_n, T)
    on = {c: bool(rev_mask & b) for c, b in bits.items()}
    y = op.forward_rev_ex(r, k, v, w, u, H, rev_n, rev_mask, algo=algo)
    got = op.backward_rev_ex(r, k, v, w, u, gy, H, rev_n, rev_mask, algo=algo)
    rg, kg, vg, wg = gather(r, idx, on["R"]), gather(k, idx, on["K"]), gather(v, idx, on["V"]), gather(w, idx, on["W"])
    y_ref = gather(op.forward_ex(rg, kg, vg, wg, u, H, algo=algo), idx, on["Y"])
    g_ref = op.backward_ex(rg, kg, vg, wg, u, gather(gy, idx, on["Y"]), H, algo=algo)
    want = [gather(g_ref[0], idx, on["R"]), gather(g_ref[1], idx, on["K"]), gather(g_ref[2], idx, on["V"]),
            gather(g_ref[3], idx, on["W"]), g_ref[4]]
    assert y.dtype == dt and torch.equal(y, y_ref)
    for name, a_, b_ in zip("gr gk gv gw gu".split(), got, want):
        assert torch.equal(a_, b_), (io, mask_name, name)


@pytest.mark.parametrize("mask_name", ["K|V|Y", "ALL"])
@pytest.mark.parametrize("B,T,H", [(4, 150, 2), (10, 96, 32)])          # the second: more slots than CUs, 2 B H = 640 workgroups
def test_pair_launch_is_the_two_calls_bit_for_bit(op, mask_name, B, T, H):
    """wkv6_forward_pair_ex / wkv6_backward_pair_ex (row n2, second half): the forward-direction and the reversed-direction
    operator call of a bidirectional layer in one launch = the two separate calls, bit for bit."""
    C = 64 * H
    rev_mask = op.REV_ALL if mask_name == "ALL" else op.REV_K | op.REV_V | op.REV_Y
    mk = lambda s0: [rnd(B, T, C, scale=0.5, seed=s0 + i) for i in range(3)] + [(rnd(B, T, C, scale=0.7, seed=s0 + 3).float() - 2.0).to(bf)]
    p0 = mk(20)
    p1 = mk(30) if mask_name == "ALL" else p0                  # composition B feeds the same tensors to both directions
    u = rnd(H, 64, scale=0.3, seed=5)
    gy0, gy1 = rnd(B, T, C, seed=6), rnd(B, T, C, seed=7)
    rev_n = torch.randint(0, T + 1, (B,), generator=torch.Generator().manual_seed(1)).to(torch.int32).cuda()
    rev_n[0], rev_n[-1] = 0, T
    ck = [op.new_checkpoint(B, T, C, H, u.device) for _ in range(4)]
    y0 = op.forward_ex(*p0, u, H, ckpt=ck[0])
    y1 = op.forward_rev_ex(*p1, u, H, rev_n, rev_mask, ckpt=ck[1])
    g0 = op.backward_ex(*p0, u, gy0, H, ckpt=ck[0])
    g1 = op.backward_rev_ex(*p1, u, gy1, H, rev_n, rev_mask, ckpt=ck[1])
    names = ("r", "k", "v", "w")
    sets = [dict(zip(names, p0), ckpt=ck[2]), dict(zip(names, p1), ckpt=ck[3], rev_n=rev_n, rev_mask=rev_mask)]
    py0, py1 = op.forward_pair_ex(H, u, sets)
    assert torch.equal(py0, y0) and torch.equal(py1, y1)
    assert torch.equal(ck[2], ck[0]) and torch.equal(ck[3], ck[1])
    sets[0]["gy"], sets[1]["gy"] = gy0, gy1
    pg0, pg1 = op.backward_pair_ex(H, u, sets)
    for name, a_, b_ in zip("gr gk gv gw gu".split(), pg0, g0):
        assert torch.equal(a_, b_), ("forward direction", name)
    for name, a_, b_ in zip("gr gk gv gw gu".split(), pg1, g1):
        assert torch.equal(a_, b_), ("reversed direction", name)
    # a forward nobody differentiates needs no checkpoints; the backward does
    sets2 = [dict(zip(names, p0)), dict(zip(names, p1), rev_n=rev_n, rev_mask=rev_mask)]
    qy0, qy1 = op.forward_pair_ex(H, u, sets2)
    assert torch.equal(qy0, y0) and torch.equal(qy1, y1)
    sets2[0]["gy"], sets2[1]["gy"] = gy0, gy1
    with pytest.raises(RuntimeError):
        op.backward_pair_ex(H, u, sets2)


def test_pair_autograd_node_survives_a_second_backward(op):
    """WKV_6_PAIR drops its checkpoints after the first backward; a second backward through the same node (retain_graph=True) must
    take the self-contained route (each problem's own state pass) and give the same gradients."""
    from rwkv_lm_ext_amd.wkv import WKV_6_PAIR
    B, T, H = 2, 100, 2
    C = 64 * H
    leaves = [rnd(B, T, C, scale=0.5, seed=40 + i).requires_grad_(True) for i in range(3)]
    w = (rnd(B, T, C, scale=0.7, seed=44).float() - 2.0).to(bf).requires_grad_(True)
    u = rnd(H, 64, scale=0.3, seed=45).requires_grad_(True)
    rev_n = torch.tensor([T, 37], dtype=torch.int32, device="cuda")
    r, k, v = leaves
    y0, y1 = WKV_6_PAIR.apply(B, T, C, H, r, k, v, w, r, k, v, w, u, rev_n, op.REV_K | op.REV_V | op.REV_Y)
    loss = (y0.float() * 0.5 + y1.float()).sum()
    first = torch.autograd.grad(loss, [r, k, v, w, u], retain_graph=True)
    second = torch.autograd.grad(loss, [r, k, v, w, u])
    for a_, b_ in zip(first, second):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("comp", ["B", "C"])
def test_compositions_with_pair_launch_equal_two_launches(comp):
    """Tmix_x060.forward_bi_b / forward_bi_c with pair_launch on and off: same outputs and parameter gradients, bit for bit."""
    torch.manual_seed(0)
    n_embd, H = 128, 2
    B, T = 3, 70
    layer = callers.Tmix_x060(n_embd, n_embd).cuda().to(bf)
    for p_ in layer.parameters():
        torch.nn.init.normal_(p_, std=0.05)
    x = rnd(B, T, n_embd, seed=3)
    mask = torch.ones(B, T, device="cuda")
    mask[1, 40:] = 0
    mask[2, 1:] = 0
    rev_idx = callers.reverse_x_idx(mask, T)
    res = []
    for pair in (True, False):
        layer.pair_launch = pair
        layer.zero_grad()
        xx = x.clone().requires_grad_(True)
        out = layer.forward_bi_b(xx, mask) if comp == "B" else layer.forward_bi_c(xx, rev_idx, mask)
        out.float().square().sum().backward()
        res.append((out.detach(), xx.grad, [p_.grad.clone() for p_ in layer.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a_, b_ in zip(res[0][2], res[1][2]):
        assert torch.equal(a_, b_)


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
