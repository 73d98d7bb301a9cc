"""Callers of the WKV6 operator on the GPU (bf16 modules, the HIP kernels underneath) against the vectors captured from
the reference's modules, plus one bi-encoder training step (config 3 shape family: query/positive/negative, InfoNCE)
whose gradients are checked against fp32 autograd through the pure-PyTorch port on the CPU.

Three yardsticks (measured values in brackets, MI355X, round 3):
(1) against the reference's fp32 vectors a whole bf16 module (bf16 GEMMs, bf16 activations) can only be held to
    max|d|/max|ref| <= 3e-2 -- that gap is the modules' precision, not the kernels';
(2) against the SAME bf16 module on the CPU (`bf16_cpu_twin`: eager torch, the WKV operator replaced by the pure-PyTorch port of
    the reference's CPU recurrence) the gap is as wide [time-mix 2.1e-2, encoder hidden 1.2e-2, gradients 4.0e-2]: two bf16
    pipelines that round at different points (the eager chain rounds every intermediate, the fused kernels once per blend);
(3) against the same GPU module with ONLY the operator swapped for that port (`swap_wkv`: same GEMM backend, same fused
    kernels, same rounding points) the operator is the only difference: OP_TOL [time-mix output 3.0e-3, encoder hidden 7.7e-3,
    logits 4.4e-3; round 4, with the channel-mix glue as HIP kernels in both copies: hidden 1.15e-2, logits 4.3e-3 -- the
    operator's last-bit differences travel through a differently rounded FFN; the fused FFN itself is CLOSER to fp32 than the
    eager chain, rel-rms 4.9e-3 against 5.9e-3, tools/diag_cmix.py]; parameter gradients of the training step 3e-2 [1.9e-2: the operator's bf16 gradients against fp32 autograd
    through the port, accumulated over 12 x 32 tokens]."""
import pytest
import torch

from conftest import load_golden, max_norm_err
from oracle import caller_weights as cw
from rwkv_lm_ext_amd import callers

pytestmark = pytest.mark.gpu
TOL = 3e-2
TWIN_TOL = 3e-2          # yardstick (2)
OP_TOL = 1.5e-2          # yardstick (3)
bf = torch.bfloat16


def _naive_bf16(B, T, C, H, r, k, v, w, u):
    from oracle.wkv6_torch_naive import wkv6_naive
    return wkv6_naive(r, k, v, w, u).to(bf)


def swap_wkv(module):
    """A copy of the GPU bf16 `module` whose time-mix blocks call the naive port of the reference's recurrence (torch ops on the
    GPU, fp32 inside, y rounded to bf16) instead of the HIP operator; everything else identical."""
    import copy
    twin = copy.deepcopy(module)
    for m in twin.modules():
        if isinstance(m, callers.Tmix_x060):
            m.wkv = _naive_bf16
    return twin


def bf16_cpu_twin(module):
    """`module` (a CPU fp32 callers.* module) as a bf16 CPU module whose time-mix blocks call the naive port: the arithmetic
    the GPU path is supposed to reproduce, at the GPU path's precision.  torch's CPU norms take bf16 inputs with fp32
    parameters only."""
    import copy
    twin = copy.deepcopy(module).to(bf)
    for m in twin.modules():
        if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)):
            m.float()
        if isinstance(m, callers.Tmix_x060):
            m.wkv, m.fused = _naive_bf16, False
    return twin


@pytest.fixture(scope="module")
def gold():
    assert torch.cuda.is_available()
    return {k: torch.from_numpy(v) for k, v in load_golden("callers").items()}


def f32(t):
    return t.detach().float().cpu()


def test_time_mix_and_bi_compositions_bf16(gold):
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    twin = bf16_cpu_twin(tm)
    tm = tm.cuda().to(bf)
    x = gold["x"].cuda().to(bf)
    with torch.no_grad():
        r, k, v, g, w = tm.jit_func(x)
        y = tm._run(r, k, v, w)
        assert y.dtype == bf
        # the op on the module's own bf16 r,k,v,w against the reference WKV output
        assert max_norm_err(f32(y), gold["y"]) <= TOL
        out = tm(x)
        assert max_norm_err(f32(out), gold["out"]) <= TOL
        out_bi = tm.forward_bi_c(x, gold["rev_idx"].cuda(), gold["mask"].cuda())
        assert max_norm_err(f32(out_bi), gold["out_bi"]) <= TOL
        # the same module at the same precision on the CPU
        xc = gold["x"].to(bf)
        e1 = max_norm_err(f32(out), f32(twin(xc)))
        e2 = max_norm_err(f32(out_bi), f32(twin.forward_bi_c(xc, gold["rev_idx"], gold["mask"])))
        e3 = max_norm_err(f32(out), f32(swap_wkv(tm)(x)))
        print(f"time-mix vs bf16 CPU twin: out {e1:.2e}, composition C {e2:.2e}; operator swapped on the GPU: {e3:.2e}")
        assert e1 <= TWIN_TOL and e2 <= TWIN_TOL and e3 <= OP_TOL, (e1, e2, e3)


def test_composition_b_on_gpu_vs_oracle(gold, oracle):
    """src/model_bi.py:325-350 on the HIP path: y = WKV(r,k,v,w,u) + unrev(WKV(r, rev k, rev v, w, u)) with the reversal
    over the unmasked prefix, against the same composition of the CPU oracle on the bf16 r,k,v,w the module produced
    (the composition tests/test_callers_cpu.py pins on the CPU)."""
    import numpy as np
    from conftest import bf16_report
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    tm = tm.cuda().to(bf)
    x, mask = gold["x"].cuda().to(bf), gold["mask"].cuda()
    with torch.no_grad():
        r, k, v, g, w = tm.jit_func(x)
        rev_idx = callers.reverse_x_idx(mask, x.shape[1])
        y_dev = tm._run(r, k, v, w).float() + callers.reverse_x(
            tm._run(r, callers.reverse_x(k, rev_idx), callers.reverse_x(v, rev_idx), w), rev_idx).float()
        got = tm.forward_bi_b(x, mask)
        rn, kn, vn, wn = (f32(t).numpy() for t in (r, k, v, w))
        u = f32(tm.time_faaaa).numpy()
        kr, vr = kn.copy(), vn.copy()
        lens = gold["mask"].sum(1).tolist()
        for b, n in enumerate(lens):
            kr[b, :n] = kn[b, :n][::-1]
            vr[b, :n] = vn[b, :n][::-1]
        y2 = oracle.forward(rn, kr, vr, wn, u)
        for b, n in enumerate(lens):
            y2[b, :n] = y2[b, :n][::-1].copy()
        y1 = oracle.forward(rn, kn, vn, wn, u)
        # the two WKV outputs are rounded to bf16 separately before they are added (as in the reference): compare the
        # sum with the sum of the correctly rounded halves -- each half within the suite's bf16 contract
        from conftest import bf16_round
        want_sum = bf16_round(y1).astype(np.float64) + bf16_round(y2).astype(np.float64)
        d = f32(y_dev).numpy().astype(np.float64) - want_sum
        scale = np.abs(want_sum).max()
        assert np.sqrt(np.mean(d ** 2)) <= 1e-3 * np.sqrt(np.mean(want_sum ** 2)) and np.abs(d).max() <= 2 * 2.0 ** -7 * scale
        want = tm.float().jit_func_2(torch.from_numpy(y1 + y2).cuda(), g.float())
    assert max_norm_err(f32(got), f32(want)) <= TOL


def test_encoder_bf16(gold):
    enc = callers.RwkvEncoder(cw.VOCAB, cw.N_EMBD, cw.N_LAYER, cw.DIM_ATT, cw.DIM_FFN)
    enc.load_state_dict(cw.encoder_weights(), strict=True)
    twin = bf16_cpu_twin(enc)
    enc = enc.cuda().to(bf)
    idx = gold["idx"].cuda()
    with torch.no_grad():
        logits, hidden = enc(idx, True)
        assert max_norm_err(f32(hidden), gold["hidden"]) <= TOL
        assert max_norm_err(f32(logits), gold["logits"]) <= TOL
        assert max_norm_err(f32(enc.encode_sentence(idx)), gold["sent"]) <= TOL
        lt, ht = twin(gold["idx"], True)
        e1, e2 = max_norm_err(f32(hidden), f32(ht)), max_norm_err(f32(logits), f32(lt))
        lo, ho = swap_wkv(enc)(idx, True)
        e3, e4 = max_norm_err(f32(hidden), f32(ho)), max_norm_err(f32(logits), f32(lo))
        print(f"encoder vs bf16 CPU twin: hidden {e1:.2e}, logits {e2:.2e}; operator swapped on the GPU: {e3:.2e}, {e4:.2e}")
        assert e1 <= TWIN_TOL and e2 <= TWIN_TOL and e3 <= OP_TOL and e4 <= OP_TOL, (e1, e2, e3, e4)


def test_bi_encoder_training_step_gradients():
    """query / positive / negative through the encoder, weighted-mean pooling, InfoNCE, backward through WKV_6 --
    the reference's RwkvForSequenceEmbedding.training_step (src/model_ext.py:1882-1911) on a tiny model."""
    from oracle.wkv6_torch_naive import wkv6_naive

    def make(wkv=None):
        enc = callers.RwkvEncoder(cw.VOCAB, cw.N_EMBD, cw.N_LAYER, cw.DIM_ATT, cw.DIM_FFN, wkv=wkv)
        enc.load_state_dict(cw.encoder_weights(), strict=True)
        return enc

    g = torch.Generator().manual_seed(21)
    bs, T = 4, 32
    idx = torch.randint(4, cw.VOCAB, (3 * bs, T), generator=g)
    lens = torch.randint(8, T - 1, (3 * bs,), generator=g)
    for b in range(3 * bs):
        idx[b, lens[b]] = 1
        idx[b, lens[b] + 1:] = 0

    def step(enc, idx):
        _, hidden = enc(idx, True)
        emb = callers.pooling(hidden, torch.eq(idx, 1).int().argmax(-1), "weightedmean").float()
        return callers.info_nce_loss(emb[:bs], emb[bs:2 * bs], emb[2 * bs:])

    ref = make(wkv=lambda B, T_, C, H, r, k, v, w, u: wkv6_naive(r, k, v, w, u))        # fp32, CPU, autograd
    loss_ref = step(ref, idx)
    loss_ref.backward()
    twin = bf16_cpu_twin(make())                                                        # bf16, CPU, autograd
    loss_twin = step(twin, idx)
    loss_twin.backward()
    enc = make().cuda().to(bf)
    swapped = swap_wkv(enc)                                                             # GPU, bf16, only the operator differs
    loss = step(enc, idx.cuda())
    loss.backward()
    loss_sw = step(swapped, idx.cuda())
    loss_sw.backward()
    assert abs(float(loss) - float(loss_ref)) <= 2e-2, (float(loss), float(loss_ref))     # [1.4e-3 with the eager FFN, 1.2e-2 with the fused one] a bf16 model against its fp32 self
    assert abs(float(loss) - float(loss_twin)) <= 3e-2, (float(loss), float(loss_twin))   # [9e-3 eager FFN; 2.2e-2 fused: the two bf16 pipelines sit 1.2e-2 above / 1.1e-2 below the fp32 loss]
    assert abs(float(loss) - float(loss_sw)) <= 1e-2, (float(loss), float(loss_sw))       # [6.5e-3 with the fused FFN in both copies; the InfoNCE logits are 20 x cosines]
    checked, worst, worst_ref, worst_sw = 0, 0.0, 0.0, 0.0
    for (n, p), (_, pr), (_, pt), (_, ps) in zip(enc.named_parameters(), ref.named_parameters(), twin.named_parameters(),
                                                 swapped.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        if any(s in n for s in ("time_faaaa", "time_decay", "key.weight", "value.weight", "receptance.weight")) and "ffn" not in n:
            # parameters whose gradient flows through the WKV backward kernels
            e_ref, e, e_sw = max_norm_err(f32(p.grad), pr.grad), max_norm_err(f32(p.grad), f32(pt.grad)), max_norm_err(f32(p.grad), f32(ps.grad))
            worst, worst_ref, worst_sw = max(worst, e), max(worst_ref, e_ref), max(worst_sw, e_sw)
            assert e_ref <= 6e-2, (n, e_ref)                # [3.6e-2] against fp32: the bf16 model's own precision
            assert e <= 6e-2, (n, e)                        # [4.0e-2] against the same bf16 model on the CPU: two bf16 pipelines
            assert e_sw <= 3e-2, (n, e_sw)                  # [1.9e-2] the operator's backward against autograd through the naive port
            checked += 1
    print(f"training step: worst gradient difference vs bf16 CPU twin {worst:.2e}, vs fp32 {worst_ref:.2e}, operator swapped on the "
          f"GPU {worst_sw:.2e}; loss {float(loss):.4f} / twin {float(loss_twin):.4f} / fp32 {float(loss_ref):.4f} / swapped {float(loss_sw):.4f}")
    assert checked >= 8


def test_infctx_time_mix_on_gpu(gold):
    """Chunks of 8 with the bf16 wkv state carried through BlockStateList (RUN_CUDA_RWKV6_INFCTX underneath) against
    the reference module's single-call output."""
    from rwkv_lm_ext_amd.infctx import BlockStateList, BlockState, ChannelMixState, tmix_forward_infctx
    tm = callers.Tmix_x060(cw.N_EMBD, cw.DIM_ATT)
    tm.load_state_dict(cw.tmix_weights(torch.Generator().manual_seed(11), layer_id=1), strict=True)
    tm = tm.cuda().to(bf)
    x = gold["x"].cuda().to(bf)
    B, T, C = x.shape
    states = BlockStateList.create(1, B, C, tm.n_head, x.device, bf)
    outs = []
    with torch.no_grad():
        for c in range(3):
            y, ts = tmix_forward_infctx(tm, x[:, 8 * c:8 * c + 8].contiguous(), states[0].time_mix_state)
            states[0] = BlockState(ts, ChannelMixState(states[0].channel_mix_state.shift_state))
            outs.append(y)
    assert max_norm_err(f32(torch.cat(outs, 1)), gold["out"]) <= TOL
    assert states.wkv_states.dtype == bf and states.wkv_states.abs().sum() > 0
