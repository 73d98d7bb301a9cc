"""Fused elementwise neighbours of the operator (csrc/wkv6_mix.hip) on the GPU against the same
math in plain PyTorch fp32 on the same bf16 inputs, and -- through the time-mix module -- against the vectors captured from
the reference's module (tests/golden/callers.npz).

Tolerances: outputs are bf16: |out - RNE_bf16(ref)| <= 1 bf16 ulp of max(|ref|, 1e-2 max|ref|) and rel-rms <= 2e-3 (one
rounding of an fp32 result); parameter gradients (fp32 partial sums over rows, rounded once) max-normalised <= 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import bf16_report, load_golden, max_norm_err

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


@pytest.fixture(scope="module")
def mix():
    assert torch.cuda.is_available()
    from rwkv_lm_ext_amd import mix_op
    return mix_op


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf).cuda()


def close(out, ref, what, ulps=1.01, rms=2e-3):
    r, off, u = bf16_report(out.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy())
    assert r <= rms and u <= ulps, f"{what}: rel-rms {r:.2e}, {u:.2f} ulp, {off * 100:.1f}% not correctly rounded"


def ddlerp_ref(x, maa, m, first):
    x, maa = x.float(), maa.float()
    prev = F.pad(x, (0, 0, 1, -1))
    if first is not None:
        prev[:, 0] = first.float()
    delta = prev - x
    mm = 0 if m is None else m.float()
    return x + delta * (maa.view(-1, 1, 1, x.shape[-1]) + mm)


@pytest.mark.parametrize("ns,has_m,carry", [(1, False, False), (5, True, False), (5, True, True), (1, True, True), (2, False, False)])
def test_ddlerp_forward_backward(mix, ns, has_m, carry):
    B, T, C = 3, 37, 256
    x = rnd(B, T, C, seed=1).requires_grad_(True)
    maa = rnd(ns, C, scale=0.5, seed=2).requires_grad_(True)
    m = rnd(ns, B, T, C, scale=0.3, seed=3).requires_grad_(True) if has_m else None
    first = rnd(B, C, seed=4).requires_grad_(True) if carry else None
    out = mix.ddlerp(x, maa, m, first)
    assert out.shape == (ns, B, T, C) and out.dtype == bf
    # reference: the same math in fp32 on the CPU
    cpu = lambda t: None if t is None else t.detach().float().cpu().clone()
    xr, mr_, mm_ = cpu(x).requires_grad_(True), cpu(maa).requires_grad_(True), (cpu(m).requires_grad_(True) if has_m else None)
    fr = cpu(first).requires_grad_(True) if carry else None
    ref = ddlerp_ref(xr, mr_, mm_, fr)
    close(out, ref, "ddlerp out")
    dout = rnd(ns, B, T, C, seed=5)
    out.backward(dout)
    ref.backward(cpu(dout))
    close(x.grad, xr.grad, "ddlerp dx", ulps=1.5)
    if has_m:
        close(m.grad, mm_.grad, "ddlerp dm")
    assert max_norm_err(maa.grad.float().cpu().numpy(), mr_.grad.numpy()) <= 1e-2
    if carry:       # the token in front of the row (infctx carry) receives the gradient of the first token's blends
        close(first.grad, fr.grad, "ddlerp dshifted0", ulps=1.5)


def test_ddlerp_carry_gradient_on_a_reversed_stream(mix):
    """With a reversed leading span the stream starts at token rev_n - 1: that token's blends carry the gradient of the
    token in front of the row.  Against the gather formulation (reverse x and m, plain shift)."""
    ns, B, T, C = 5, 4, 29, 128
    rev_n = torch.tensor([0, 1, 13, T], dtype=torch.int32, device="cuda")
    x, maa, m = rnd(B, T, C, seed=1), rnd(ns, C, scale=0.5, seed=2), rnd(ns, B, T, C, scale=0.3, seed=3)
    first = rnd(B, C, seed=4).requires_grad_(True)
    dout = rnd(ns, B, T, C, seed=5)
    mix.ddlerp(x, maa, m, first, rev_n).backward(dout)
    ar = torch.arange(T, device="cuda").view(1, T).expand(B, T)
    n = rev_n.long().view(B, 1)
    idx = torch.where(ar < n, n - 1 - ar, ar)                                  # stream position -> token
    gat = lambda t: torch.gather(t, t.dim() - 2, idx.view(*([1] * (t.dim() - 3)), B, T, 1).expand(*t.shape[:-1], C))
    first2 = first.detach().clone().requires_grad_(True)
    mix.ddlerp(gat(x).contiguous(), maa, gat(m).contiguous(), first2, None).backward(gat(dout).contiguous())
    assert torch.equal(first.grad, first2.grad)


def test_infctx_shift_state_gradient_fused_vs_unfused():
    """time_mix through the infctx carry (src/model.py:773-782 keeps the shift state in the graph): the gradient of the incoming
    shift state from the fused HIP path against the eager path."""
    from oracle import caller_weights as cw
    from rwkv_lm_ext_amd import callers, infctx
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    tm = tm.cuda().to(bf)
    B, T, C = 2, 24, cw.N_EMBD
    x = rnd(B, T, C, seed=21)
    grads = {}
    for fused in (True, False):
        tm.fused = fused
        shift = rnd(B, C, seed=22).requires_grad_(True)
        wkv = rnd(B, tm.n_head, 64, 64, scale=0.3, seed=23)
        out, _ = infctx.tmix_forward_infctx(tm, x, infctx.TimeMixState(shift, wkv))
        out.backward(rnd(B, T, C, seed=24))
        grads[fused] = shift.grad.float()
    tm.fused = None
    err = (grads[True] - grads[False]).abs().max() / grads[False].abs().max()
    assert float(err) <= 2e-2, float(err)        # two bf16 pipelines (one rounding per blend vs several in the eager chain)


def test_group_norm_gate_forward_backward(mix):
    rows, H = 301, 4
    C = 64 * H
    eps = 1e-5 * 64
    y = rnd(rows, C, scale=2.0, seed=6).requires_grad_(True)
    g = rnd(rows, C, seed=7).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(8))).to(bf).cuda().requires_grad_(True)
    beta = rnd(C, scale=0.1, seed=9).requires_grad_(True)
    out = mix.group_norm_gate(y, g, gamma, beta, H, eps)
    # reference in fp32 on the CPU (F.group_norm's weight gradient on the GPU disagrees with the definition for [N,C] inputs)
    leaves = [t.detach().float().cpu().clone().requires_grad_(True) for t in (y, g, gamma, beta)]
    ref = F.group_norm(leaves[0], H, leaves[2], leaves[3], eps) * leaves[1]
    close(out, ref, "gn_gate out")
    dout = rnd(rows, C, seed=10)
    out.backward(dout)
    ref.backward(dout.float().cpu())
    xh = F.group_norm(leaves[0].detach(), H, None, None, eps)
    assert max_norm_err(leaves[2].grad.numpy(), (dout.float().cpu() * leaves[1].detach() * xh).sum(0).numpy()) <= 1e-5
    close(y.grad, leaves[0].grad, "gn_gate dy", ulps=1.5)
    close(g.grad, leaves[1].grad, "gn_gate dg")
    assert max_norm_err(gamma.grad.float().cpu().numpy(), leaves[2].grad.numpy()) <= 1e-2
    assert max_norm_err(beta.grad.float().cpu().numpy(), leaves[3].grad.numpy()) <= 1e-2


def test_full_width_rows(mix):
    """C = 2048 (512 threads per row), many rows: the launch shape of the 1B6 model."""
    B, T, C, H = 2, 640, 2048, 32
    x, maa, m = rnd(B, T, C, seed=11), rnd(5, C, scale=0.5, seed=12), rnd(5, B, T, C, scale=0.3, seed=13)
    close(mix.ddlerp(x, maa, m), ddlerp_ref(x.cpu(), maa.cpu(), m.cpu(), None), "ddlerp C=2048")
    y, g = rnd(B * T, C, seed=14), rnd(B * T, C, seed=15)
    gamma, beta = rnd(C, seed=16), rnd(C, seed=17)
    ref = F.group_norm(y.float().cpu(), H, gamma.float().cpu(), beta.float().cpu(), 6.4e-4) * g.float().cpu()
    close(mix.group_norm_gate(y, g, gamma, beta, H, 6.4e-4), ref, "gn_gate C=2048")


def test_time_mix_module_fused_matches_reference_vectors_and_unfused():
    """Tmix_x060 with the fused stages against the reference module's captured outputs, and gradients of the fused module
    against the unfused (plain PyTorch) module on the same bf16 weights."""
    from oracle import caller_weights as cw
    from rwkv_lm_ext_amd import callers
    gold = {k: torch.from_numpy(v) for k, v in load_golden("callers").items()}

    def make(fused):
        tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT, fused=fused)
        tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
        return tm.cuda().to(bf)

    x = gold["x"]
    outs, grads = [], []
    for fused in (True, False):        # fused: bf16 on the GPU through the HIP kernels; unfused: fp32 on the CPU, oracle WKV
        if fused:
            tm = make(True)
            xi = x.cuda().to(bf).requires_grad_(True)
        else:
            from oracle.wkv6_torch_naive import wkv6_naive
            tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT, fused=False, wkv=lambda B, T, C, H, r, k, v, w, u: wkv6_naive(r, k, v, w, u))
            tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
            xi = x.clone().requires_grad_(True)
        r, k, v, g, w = tm.jit_func(xi)
        for got, name in ((r, "r"), (k, "k"), (v, "v"), (g, "g"), (w, "w")):
            assert max_norm_err(got.detach().float().cpu().numpy(), gold[name].numpy()) <= (3e-2 if fused else 1e-4), name
        out = tm(xi)
        assert max_norm_err(out.detach().float().cpu().numpy(), gold["out"].numpy()) <= (3e-2 if fused else 1e-4)
        out.float().pow(2).sum().backward()
        outs.append(out.detach().float().cpu())
        grads.append({n: p.grad.float().cpu() for n, p in tm.named_parameters()} | {"x": xi.grad.float().cpu()})
    for n in grads[0]:
        e = max_norm_err(grads[0][n].numpy(), grads[1][n].numpy())
        assert e <= 8e-2, (n, e)          # a bf16 pipeline of ~12 ops (GEMMs, WKV, fused stages) against fp32


@pytest.mark.parametrize("shape", [(5, 200, 32), (2, 129, 64), (3, 17, 2)], ids=["B5T200H32", "B2T129H64", "B3T17H2"])
def test_group_norm_gate_fused_into_the_operator_forward(shape, monkeypatch):
    """WKV_6_GN = WKV6 -> GroupNorm(H) -> * gate in one forward kernel, against the two-kernel path
    (WKV_6 + mix_op.group_norm_gate) on the same inputs: the statistics are taken of the same bf16-rounded y, so the outputs agree
    to one bf16 ulp (the per-head sums run in a different order) and the gradients likewise; without grad no y is written."""
    from rwkv_lm_ext_amd import wkv, wkv6_op
    monkeypatch.setenv("WKV6_SPLIT", "0")                     # (few heads would otherwise fall back to the two kernels)
    B, T, H = shape
    C = 64 * H
    mk = lambda *s, scale=1.0, seed=0: rnd(*s, scale=scale, seed=seed).requires_grad_(True)
    r, k, v, g = mk(B, T, C, scale=0.5, seed=1), mk(B, T, C, scale=0.5, seed=2), mk(B, T, C, scale=0.5, seed=3), mk(B, T, C, seed=4)
    w = (rnd(B, T, C, scale=0.5, seed=5) - 1.0).detach().requires_grad_(True)
    u = mk(H, 64, scale=0.3, seed=6)
    gamma, beta = (rnd(C, scale=0.2, seed=7) + 1.0).detach().requires_grad_(True), mk(C, scale=0.1, seed=8)
    eps, dout = 64e-5, rnd(B, T, C, seed=9)
    leaves = (r, k, v, w, u, g, gamma, beta)
    out_f = wkv.RUN_CUDA_RWKV6_GN(B, T, C, H, r, k, v, w, u, g, gamma, beta, eps)
    out_f.backward(dout)
    got = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    y = wkv.RUN_CUDA_RWKV6(B, T, C, H, r, k, v, w, u)
    out_2 = mix_op_group(y, g, gamma, beta, H, eps)
    out_2.backward(dout)
    close(out_f, out_2.float(), "fused out vs two kernels", ulps=1.01, rms=1e-3)
    for n, a, t in zip(("gr", "gk", "gv", "gw", "gu", "dg", "dgamma", "dbeta"), got, leaves):
        close(a, t.grad.float(), "fused " + n, ulps=2.02 if n in ("gw", "dgamma", "dbeta", "gu") else 1.5, rms=2e-3)
    with torch.no_grad():                                       # inference: the same `out`, y is not materialised
        res = wkv6_op.forward_gn_ex(r, k, v, w, u, H, g, gamma, beta, eps, want_y=False, want_stats=False)
        assert res is not None and res[1] is None and res[2] is None and torch.equal(res[0], out_f)
    monkeypatch.setenv("WKV6_SPLIT", "1")                       # two workgroups per head: the library declines, the op falls back
    with torch.no_grad():
        assert wkv6_op.forward_gn_ex(r, k, v, w, u, H, g, gamma, beta, eps) is None
        close(wkv.RUN_CUDA_RWKV6_GN(B, T, C, H, r, k, v, w, u, g, gamma, beta, eps), out_2.float(), "fallback out", ulps=1.01, rms=1e-3)


def mix_op_group(y, g, gamma, beta, H, eps):
    from rwkv_lm_ext_amd import mix_op
    B, T, C = y.shape
    return mix_op.group_norm_gate(y.reshape(B * T, C), g.reshape(B * T, C), gamma, beta, H, eps).view(B, T, C)


def test_time_mix_module_with_the_fused_epilogue():
    """callers.Tmix_x060(fuse_epilogue=True): forward and input / parameter gradients against the default two-kernel module."""
    from oracle import caller_weights as cw
    from rwkv_lm_ext_amd import callers
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    tm = tm.cuda().to(bf)
    os_ = __import__("os")
    os_.environ["WKV6_SPLIT"] = "0"
    try:
        res = {}
        for fuse in (False, True):
            tm.fuse_epilogue = fuse
            x = rnd(3, 40, cw.N_EMBD, seed=31).requires_grad_(True)
            tm.zero_grad(set_to_none=True)
            out = tm(x)
            out.backward(rnd(3, 40, cw.N_EMBD, seed=32))
            res[fuse] = (out.detach(), x.grad.clone(), tm.ln_x.weight.grad.clone(), tm.time_faaaa.grad.clone())
    finally:
        del os_.environ["WKV6_SPLIT"]
        tm.fuse_epilogue = False
    for n, a, b in zip(("out", "dx", "d ln_x.weight", "d time_faaaa"), res[True], res[False]):
        close(a, b.float(), "fused-epilogue module " + n, ulps=2.02, rms=3e-3)



def test_channel_mix_glue_kernels(mix):
    """sqrelu and sigmoid_mul (src/model.py:640-644) forward and backward against fp32 torch on the same bf16 inputs."""
    x = rnd(5, 33, 512, scale=1.5, seed=21).requires_grad_(True)
    xr = x.detach().float().cpu().requires_grad_(True)
    out, ref = mix.sqrelu(x), torch.relu(xr) ** 2
    close(out, ref, "sqrelu out")
    d = rnd(5, 33, 512, seed=22)
    out.backward(d)
    ref.backward(d.float().cpu())
    close(x.grad, xr.grad, "sqrelu dx")
    r, kv = rnd(4, 50, 256, scale=2.0, seed=23).requires_grad_(True), rnd(4, 50, 256, seed=24).requires_grad_(True)
    rr, kr = r.detach().float().cpu().requires_grad_(True), kv.detach().float().cpu().requires_grad_(True)
    out, ref = mix.sigmoid_mul(r, kv), torch.sigmoid(rr) * kr
    close(out, ref, "sigmul out")
    d = rnd(4, 50, 256, seed=25)
    out.backward(d)
    ref.backward(d.float().cpu())
    close(r.grad, rr.grad, "sigmul dr")
    close(kv.grad, kr.grad, "sigmul dkv")


def test_channel_mix_module_fused_equals_eager():
    """CMix_x060 with the HIP glue kernels against its eager form (same bf16 parameters): the fused kernels round once where the
    eager chain rounds after every op, so the comparison is at a few bf16 ulps; gradients likewise."""
    from rwkv_lm_ext_amd import callers
    torch.manual_seed(3)
    C, F_ = 256, 896
    cm = callers.CMix_x060(C, F_).cuda().to(bf)
    with torch.no_grad():
        cm.time_maa_k.uniform_(0.1, 0.9)
        cm.time_maa_r.uniform_(0.1, 0.9)
    x = rnd(3, 40, C, seed=31)
    outs = []
    for fused in (True, False):
        cm.fused = fused
        xi = x.clone().requires_grad_(True)
        y = cm(xi)
        y.backward(rnd(3, 40, C, seed=32))
        outs.append((y.detach().float().cpu(), xi.grad.float().cpu(), cm.key.weight.grad.float().cpu().clone()))
        cm.zero_grad()
    for a_, b_, what in zip(outs[0], outs[1], ("y", "dx", "dW_key")):
        assert float((a_ - b_).abs().max()) <= 3e-2 * float(b_.abs().max()), what
    # against the fp32 module on the bf16-rounded parameters: the fused form (one rounding per kernel) is at least as close as the eager chain
    ref = callers.CMix_x060(C, F_)
    ref.load_state_dict({k: v.float().cpu() for k, v in cm.state_dict().items()})
    want = ref(x.float().cpu())
    e_fused = float((outs[0][0] - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    e_eager = float((outs[1][0] - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    assert e_fused <= 1.1 * e_eager + 1e-4 and e_fused <= 1e-2, (e_fused, e_eager)


def test_lora_linear_epilogue_add_equals_eager():
    """train_dp.LoraLinear on bf16 GPU tensors adds the low-rank term in the second GEMM's epilogue: same values as the eager
    expression to bf16 rounding, same gradients for x, A and B; the frozen weight gets none."""
    from rwkv_lm_ext_amd import train_dp
    torch.manual_seed(5)
    lin = train_dp.LoraLinear(256, 640, r=8, alpha=32.0).cuda().to(bf)
    with torch.no_grad():
        lin.lora_B.normal_(0.0, 0.05)
    x = rnd(4, 30, 256, seed=41)
    gy = rnd(4, 30, 640, seed=42)
    xi = x.clone().requires_grad_(True)
    y = lin(xi)                                        # fused path (bf16 GPU, no dropout)
    y.backward(gy)
    got = (y.detach().float(), xi.grad.float(), lin.lora_A.grad.float().clone(), lin.lora_B.grad.float().clone())
    assert lin.weight.grad is None
    lin.zero_grad()
    xf = x.float().clone().requires_grad_(True)       # reference: the eager expression in fp32
    W, A, Bm = lin.weight.float(), lin.lora_A.float().detach().requires_grad_(True), lin.lora_B.float().detach().requires_grad_(True)
    yr = F.linear(xf, W) + lin.scaling * F.linear(F.linear(xf, A), Bm)
    yr.backward(gy.float())
    for a_, b_, what in zip(got, (yr.detach(), xf.grad, A.grad, Bm.grad), ("y", "dx", "dA", "dB")):
        assert float((a_ - b_).abs().max()) <= 2e-2 * float(b_.abs().max()), what
