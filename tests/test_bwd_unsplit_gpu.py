"""GPU parity of the backward kernel (csrc/wkv6_chunk_bwd12k.hip: 64-token row-order checkpoints, K part of the stage image two
stages ahead) in its ONE-workgroup-per-(batch, head) mode, forced onto every shape of the suite: the suite's small shapes would
otherwise run its two-workgroups-per-pair mode, so WKV6_SPLIT=0 puts them on the mode the benched shapes use (the split mode is
what the rest of the suite exercises, and both are bit-identical: test_wkv6_gpu.py::test_two_workgroups_per_head_is_the_same_arithmetic).
Same bf16 contract as everything else: golden vectors generated from the reference, the oracle on random shapes, the exact scan
kernels at every block / stage / checkpoint boundary, the wkv6_bi and in-kernel-reversal store paths, rows of 0 .. 65 tokens beside
long ones in one launch (the hand-over protocol at its edges), and config 2 at full size against oracle slices and the scan kernels."""
import numpy as np
import pytest
import torch

from conftest import load_golden, max_norm_err
from test_wkv6_gpu import PART_TOL, check, dev, host, rand_inputs

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "the gpu suite needs a GPU"
    from rwkv_lm_ext_amd import wkv6_op
    assert wkv6_op.selftest() == 0
    return wkv6_op


@pytest.fixture
def two_level(monkeypatch):
    """One workgroup per (batch, head) for every shape (the library reads the switch at each call): small (batch, head) counts would
    otherwise run two workgroups per pair."""
    monkeypatch.setenv("WKV6_SPLIT", "0")


@pytest.mark.parametrize("name", ["wkv6_init", "wkv6_stress", "wkv6_extreme", "wkv6_T1", "wkv6_T2", "wkv6_T3", "wkv6_state", "wkv6_infctx"])
def test_golden_vectors(ops, two_level, name):
    g = load_golden(name)
    H = g["u"].shape[0]
    r, k, v, w, u, gy = (dev(g[n], BF) for n in ("r", "k", "v", "w", "u", "gy"))
    s = dev(g["s"], BF) if "s" in g else None
    B, T, C = g["r"].shape
    ck = ops.new_checkpoint(B, T, C, H, "cuda")
    check(ops.forward_ex(r, k, v, w, u, H, s0=s, ckpt=ck), g["y"], BF, name + " y")
    for use_ckpt in (True, False):           # checkpoints from the forward / from the backward's own state pass
        gr, gk, gv, gw, gu, gs = ops.backward_ex(r, k, v, w, u, gy, H, s0=s, want_gs=s is not None, ckpt=ck if use_ckpt else None)
        for n, t in (("gr", gr), ("gk", gk), ("gv", gv), ("gw", gw)):
            check(t, g[n], BF, f"{name} {n}")
        assert max_norm_err(host(gu).sum(0).reshape(H, 64), g["gu"]) <= PART_TOL
        if s is not None:
            gs_ref = g["gs"]
            got = host(gs).sum(0) if gs_ref.ndim == 3 else host(gs)
            assert max_norm_err(got, gs_ref) <= PART_TOL


@pytest.mark.parametrize("shape", [(2, 300, 3, "stress"), (1, 1000, 2, "init"), (3, 17, 1, "stress"), (2, 64, 2, "init"), (1, 129, 1, "stress")],
                         ids=["B2T300H3", "B1T1000H2", "B3T17H1", "B2T64H2", "B1T129H1"])
def test_random_vs_oracle_with_state(ops, oracle, two_level, shape):
    B, T, H, kind = shape
    r, k, v, w, u, gy = rand_inputs(300 + T, B, T, H, kind)
    g = torch.Generator().manual_seed(T + 1)
    s0 = (torch.randn(B, H, 64, 64, generator=g) * 0.5).to(BF).float().numpy()
    d = [dev(t, BF) for t in (r, k, v, w, u, gy)]
    og = oracle.backward(r, k, v, w, u, gy, s0)
    gr, gk, gv, gw, gu, gs = ops.backward_ex(*d, H, s0=dev(s0, BF), want_gs=True)
    for n, t in (("gr", gr), ("gk", gk), ("gv", gv), ("gw", gw), ("gs_b", gs.to(BF))):
        check(t, og[n], BF, f"{n} (two-level)")
    assert max_norm_err(host(gu), og["gu_b"]) <= PART_TOL


@pytest.mark.parametrize("T", [1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 79, 80, 81, 127, 128, 129, 191, 193])
def test_block_and_chunk_boundaries_vs_scan(ops, two_level, T):
    """Lengths around the 16-token block and the 64-token chunk boundaries, with an initial state, against the exact scan kernels."""
    B, H = 2, 2
    r, k, v, w, u, gy = rand_inputs(300 + T, B, T, H, "init")
    g = torch.Generator().manual_seed(T)
    s0 = dev((torch.randn(H, 64, 64, generator=g) * 0.5).numpy(), BF)
    d = [dev(t, BF) for t in (r, k, v, w, u, gy)]
    ck = ops.new_checkpoint(B, T, H * 64, H, "cuda")
    ops.forward_ex(*d[:5], H, s0=s0, ckpt=ck)
    oc = ops.backward_ex(*d, H, s0=s0, want_gs=True, ckpt=ck)
    osn = ops.backward_ex(*d, H, s0=s0, want_gs=True, algo="scan")
    for n, c, s_ in zip(("gr", "gk", "gv", "gw", "gu", "gs"), oc, osn):
        c, s_ = host(c), host(s_)
        scale = max(float(np.abs(s_).max()), 1e-3)
        assert float(np.abs(c - s_).max()) <= (4.0 if n in ("gw", "gu", "gs") else 2.0) * 2.0 ** -8 * scale, (n, T)


def test_extreme_decay_across_block_boundaries(ops, oracle, two_level):
    """Every channel decays by more than e^-9 per token (the clamp): frame gaps of ~200 binary orders between neighbouring
    blocks, so the factor between two frames underflows fp32 while the term it scales is of order one (the case that an
    earlier build of this kernel got wrong at exactly one element per block boundary)."""
    B, T, H = 1, 77, 1
    g = torch.Generator().manual_seed(1077)
    bf = lambda x: x.to(BF).float().numpy()
    r, k, v, gy = (bf(torch.randn(B, T, 64, generator=g) * 0.5) for _ in range(4))
    w = bf(0.5 + 2.0 * torch.rand(B, T, 64, generator=g))
    u = bf(torch.randn(H, 64, generator=g) * 0.3)
    s0 = bf(torch.randn(B, H, 64, 64, generator=g) * 0.3)
    d = [dev(t, BF) for t in (r, k, v, w, u, gy)]
    two = ops.backward_ex(*d, H, s0=dev(s0, BF), want_gs=True)
    scan = ops.backward_ex(*d, H, s0=dev(s0, BF), want_gs=True, algo="scan")
    for n, a, b in zip(("gr", "gk", "gv", "gw", "gu", "gs"), two, scan):
        a, b = host(a), host(b)
        # the chunked kernels clamp the per-token decay at e^-9: up to 1.3e-4 of the carried state
        assert float(np.abs(a - b).max()) <= 3.0 * 2.0 ** -8 * max(float(np.abs(b).max()), 1e-3), n


def test_wkv6_bi_store_paths(ops, two_level):
    """wkv6_bi (GEN instantiation: fp32 side buffers, accumulation, tail zeroing, per-row lengths, reversed second scan) on the
    golden vector and on ragged random rows against the scan kernels."""
    g = load_golden("wkv6_bi")
    H = g["u"].shape[0]
    r, k, v, w, u, gy = (dev(g[n], BF) for n in ("r", "k", "v", "w", "u", "gy"))
    mask = torch.from_numpy(g["mask"]).to("cuda", torch.int32)
    gr, gk, gv, gw, gu = ops.bi_backward_ex(mask, r, k, v, w, u, gy, H)
    for n, t in (("gr", gr), ("gk", gk), ("gv", gv), ("gw", gw)):
        check(t, g[n], BF, "bi " + n)
        assert np.all(host(t)[1, 18:] == 0)
    assert max_norm_err(host(gu).sum(0).reshape(H, 64), g["gu"]) <= PART_TOL
    rng = np.random.default_rng(7)
    for case in range(6):
        B, H, T = int(rng.integers(1, 4)), int(rng.integers(1, 3)), int(rng.choice([17, 64, 65, 130, 191]))
        ri = rand_inputs(6000 + case, B, T, H, ("init", "stress")[case % 2])
        d = [dev(t, BF) for t in ri]
        mask = torch.ones(B, T, dtype=torch.int32)
        for b in range(B):
            cut = int(rng.integers(0, T + 1))
            if cut < T:
                mask[b, cut:] = 0
        m = mask.cuda()
        ws = ops.bi_new_workspace(B, T, H * 64, H, "cuda")
        ops.bi_forward_ex(m, *d[:5], H, ws=ws)                         # keeps both directions' checkpoints (64 tokens apart)
        for got, ref in ((ops.bi_backward_ex(m, *d, H, ws=ws), ops.bi_backward_ex(m, *d, H, algo="scan")),
                         (ops.bi_backward_ex(m, *d, H), ops.bi_backward_ex(m, *d, H, algo="scan"))):
            for n, c, s_ in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
                c, s_ = host(c), host(s_)
                scale = max(float(np.abs(s_).max()), 1e-2 if n == "gw" else 1e-3)
                assert float(np.abs(c - s_).max()) <= (4.0 if n in ("gw", "gu") else 2.0) * 2.0 ** -8 * scale, (case, n)


def test_in_kernel_reversal(ops, two_level, monkeypatch):
    """wkv6_backward_rev_ex: per-tensor reversal bits and per-row spans: the chunked kernels against the exact scan kernels."""
    B, T, H = 3, 150, 2
    ri = rand_inputs(4242, B, T, H, "init")
    d = [dev(t, BF) for t in ri]
    rev_n = torch.tensor([150, 64, 1], dtype=torch.int32, device="cuda")
    for mask in (ops.REV_ALL, ops.REV_R | ops.REV_Y, ops.REV_K | ops.REV_V | ops.REV_W):
        ck = ops.new_checkpoint(B, T, H * 64, H, "cuda")
        ops.forward_rev_ex(*d[:5], H, rev_n, mask, ckpt=ck)
        got = ops.backward_rev_ex(*d, H, rev_n, mask, ckpt=ck)
        ref = ops.backward_rev_ex(*d, H, rev_n, mask, algo="scan")          # the exact kernels with the same index maps
        for n, a, b in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
            a, b = host(a), host(b)
            scale = max(float(np.abs(b).max()), 1e-3)
            assert float(np.abs(a - b).max()) <= (4.0 if n in ("gw", "gu") else 2.0) * 2.0 ** -8 * scale, (mask, n)


def test_checkpoint_path_is_the_state_pass_path(ops, two_level):
    """forward_ex(ckpt=) + backward_ex(ckpt=) == the self-contained backward (own state pass), bit for bit; and the checkpoint
    buffer the library asks for at 64-token spacing is half the 32-token one for long sequences."""
    B, T, H = 2, 333, 2
    d = [dev(t, BF) for t in rand_inputs(91, B, T, H, "stress")]
    ck = ops.new_checkpoint(B, T, H * 64, H, "cuda")
    ops.forward_ex(*d[:5], H, ckpt=ck)
    g1 = ops.backward_ex(*d, H, ckpt=ck)
    g0 = ops.backward_ex(*d, H)
    for a_, b_ in zip(g0[:5], g1[:5]):
        assert torch.equal(a_, b_)


def test_config2_full_size_vs_oracle_slices_and_scan(ops, oracle, two_level, monkeypatch):
    """BASELINE configs[1] at full size (B=8, T=4096, H=32): oracle on (batch, head) slices, and agreement with the exact scan
    backward over the whole tensors."""
    from bench import synth
    B, T, H = 8, 4096, 32
    C = H * 64
    r, k, v, w, u, gy = synth(B, T, H, torch.device("cuda", 0))
    ck = ops.new_checkpoint(B, T, C, H, r.device)
    assert ck.numel() == B * T * C * 4                            # 4 B per token-channel
    ops.forward_ex(r, k, v, w, u, H, ckpt=ck)
    gr, gk, gv, gw, gu, _ = ops.backward_ex(r, k, v, w, u, gy, H, ckpt=ck)
    for (b, h) in ((0, 0), (7, 31), (3, 16)):
        sl = (slice(b, b + 1), slice(None), slice(64 * h, 64 * h + 64))
        rs, ks, vs, ws, gys = (host(x[sl]) for x in (r, k, v, w, gy))
        og = oracle.backward(rs, ks, vs, ws, host(u[h:h + 1]), gys)
        for n, t in (("gr", gr), ("gk", gk), ("gv", gv)):
            check(t[sl], og[n], BF, f"config2 ({b},{h}) {n}")
        # gw: a suffix sum over 4096 tokens; its last bit depends on the summation order (>= 90 % exactly rounded, as for the default kernel)
        from conftest import bf16_report
        rms, off, ulps = bf16_report(host(gw[sl]), og["gw"], floor=0.1)
        assert rms <= 1e-3 and ulps <= 2.0 and off <= 0.10, (b, h, rms, off, ulps)
        assert max_norm_err(host(gu[b, 64 * h:64 * h + 64]), og["gu_b"][0]) <= 1e-3
    ref = ops.backward_ex(r, k, v, w, u, gy, H, algo="scan")          # the exact fp32-state kernels, whole tensors
    for n, a, b_ in zip(("gr", "gk", "gv", "gw"), (gr, gk, gv, gw), ref):
        a, b_ = host(a), host(b_)
        big = np.abs(b_) >= 1e-2 * np.abs(b_).max()
        same = float(np.mean(a[big] == b_[big]))
        worst = float((np.abs(a - b_)[big] / np.abs(b_[big])).max())
        # (gw: each kernel is within two bf16 ulps of the oracle on the slices above; between themselves they may be three apart)
        assert same >= (0.9 if n == "gw" else 0.97) and worst <= 2.0 ** -7 * (3.03 if n == "gw" else 2.02), (n, same, worst)


def test_ragged_rows_share_one_unsplit_launch(ops, two_level):
    """Rows of 0, 1, 31, 32, 33, 63, 64, 65 tokens beside long ones in ONE launch with one workgroup per (batch, head): the hand-over
    protocol of the 12-wave kernel at its edges -- workgroups with no stage at all, with one partial stage, with exactly one / two /
    three stages next to 7-stage neighbours (every role executes every stage of ITS row: a tag can only be waited for by a wave whose
    partner is about to write it).  wkv6_bi (first half: forward-direction scan into the fp32 side buffers; second half: reversed
    scan, accumulating) with the row lengths passed directly, and the in-kernel-reversed operator with the same spans, against the
    exact scan kernels; forward with kept checkpoints and self-contained backward."""
    B, T, H = 10, 200, 2
    lens = torch.tensor([0, 1, 31, 32, 33, 63, 64, 65, 200, 129], dtype=torch.int32, device="cuda")
    d = [dev(t, BF) for t in rand_inputs(9191, B, T, H, "init")]
    ws = ops.bi_new_workspace(B, T, H * 64, H, "cuda")
    y = ops.bi_forward_ex(None, *d[:5], H, ws=ws, lens=lens)
    ys = ops.bi_forward_ex(None, *d[:5], H, algo="scan", lens=lens)
    scale = float(np.abs(host(ys)).max())
    assert float(np.abs(host(y) - host(ys)).max()) <= 2.0 * 2.0 ** -8 * scale
    for b in range(B):
        assert np.all(host(y)[b, int(lens[b]):] == 0)
    ref = ops.bi_backward_ex(None, *d, H, algo="scan", lens=lens)
    for got in (ops.bi_backward_ex(None, *d, H, ws=ws, lens=lens), ops.bi_backward_ex(None, *d, H, lens=lens)):
        for n, c, s_ in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
            c, s_ = host(c), host(s_)
            scale = max(float(np.abs(s_).max()), 1e-2 if n == "gw" else 1e-3)
            assert float(np.abs(c - s_).max()) <= (4.0 if n in ("gw", "gu") else 2.0) * 2.0 ** -8 * scale, n
            if n != "gu":
                for b in range(B):
                    assert np.all(c[b, int(lens[b]):] == 0), (n, b)
    # the reversed operator: every row is scanned in full, the first rev_n[b] tokens in reverse order
    for mask in (ops.REV_ALL, ops.REV_K | ops.REV_V | ops.REV_Y):
        ck = ops.new_checkpoint(B, T, H * 64, H, "cuda")
        yr = ops.forward_rev_ex(*d[:5], H, lens, mask, ckpt=ck)
        yrs = ops.forward_rev_ex(*d[:5], H, lens, mask, algo="scan")
        assert float(np.abs(host(yr) - host(yrs)).max()) <= 2.0 * 2.0 ** -8 * float(np.abs(host(yrs)).max())
        got = ops.backward_rev_ex(*d, H, lens, mask, ckpt=ck)
        ref = ops.backward_rev_ex(*d, H, lens, mask, algo="scan")
        for n, c, s_ in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
            c, s_ = host(c), host(s_)
            scale = max(float(np.abs(s_).max()), 1e-3)
            assert float(np.abs(c - s_).max()) <= (4.0 if n in ("gw", "gu") else 2.0) * 2.0 ** -8 * scale, (mask, n)


def test_ragged_rows_with_the_reference_decay_kind_share_one_fused_launch(ops, two_level):
    """The same rows through the reference's decay kind (fp32 ew = -exp(w), what cuda/wkv6_bi_op.cpp:5-13 / cuda/wkv6_bi.py:31-56 pass):
    since round 6 the fp32-ew instantiations of BOTH persistent wkv6_bi launches (chunk_fwd_bi_kernel<false>,
    chunk_bwd12k_bi_kernel<false>) serve it -- one launch per pass -- where round 5's backward fell back to two launches.  Rows of
    0, 1, 31 .. 65 tokens beside long ones against the exact scan kernels on the same ew, with kept checkpoints and self-contained."""
    B, T, H = 10, 200, 2
    lens = torch.tensor([0, 1, 31, 32, 33, 63, 64, 65, 200, 129], dtype=torch.int32, device="cuda")
    d = [dev(t, BF) for t in rand_inputs(9292, B, T, H, "init")]
    d[3] = (-torch.exp(d[3].float())).contiguous()                       # ew, as WKV_6_BI.forward forms it (cuda/wkv6_bi.py:31)
    ws = ops.bi_new_workspace(B, T, H * 64, H, "cuda")
    y = ops.bi_forward_ex(None, *d[:5], H, ws=ws, lens=lens, w_is_ew=True)
    ys = ops.bi_forward_ex(None, *d[:5], H, algo="scan", lens=lens, w_is_ew=True)
    scale = float(np.abs(host(ys)).max())
    assert float(np.abs(host(y) - host(ys)).max()) <= 2.0 * 2.0 ** -8 * scale
    for b in range(B):
        assert np.all(host(y)[b, int(lens[b]):] == 0)
    ref = ops.bi_backward_ex(None, *d, H, algo="scan", lens=lens, w_is_ew=True)
    for got in (ops.bi_backward_ex(None, *d, H, ws=ws, lens=lens, w_is_ew=True), ops.bi_backward_ex(None, *d, H, lens=lens, w_is_ew=True)):
        for n, c, s_ in zip(("gr", "gk", "gv", "gw", "gu"), got, ref):
            c, s_ = host(c), host(s_)
            scale = max(float(np.abs(s_).max()), 1e-2 if n == "gw" else 1e-3)
            assert float(np.abs(c - s_).max()) <= (4.0 if n in ("gw", "gu") else 2.0) * 2.0 ** -8 * scale, n
            if n != "gu":
                for b in range(B):
                    assert np.all(c[b, int(lens[b]):] == 0), (n, b)


@pytest.mark.parametrize("ew", [False, True], ids=["raw_w", "fp32_ew"])
def test_chained_persistent_launches_equal_the_two_launch_halves_bit_for_bit(ops, two_level, monkeypatch, ew):
    """Round 6 chained the calls of the persistent wkv6_bi backward (a call's producers, column and row waves prepare the call that follows;
    the stage -> LDS slot maps rotate across the call boundary) and fused the fp32-ew kind: every wave still does the arithmetic of the plain
    two-launch kernels (WKV6_BI_FUSED=0), so forward and backward must agree BIT FOR BIT -- on rows of every chaining case: fewer than two
    stages (not chained, nor chained into), exactly two, odd and even stage counts, several rows per workgroup slot."""
    B, T, H = 12, 330, 32                                             # 384 rows on 256 slots: a second row behind half of the slots
    lens = torch.tensor([330, 0, 1, 32, 33, 64, 65, 96, 97, 128, 200, 313], dtype=torch.int32, device="cuda")
    d = [dev(t, BF) for t in rand_inputs(9393, B, T, H, "stress")]
    if ew:
        d[3] = (-torch.exp(d[3].float())).contiguous()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("WKV6_BI_FUSED", mode)
        ws = ops.bi_new_workspace(B, T, H * 64, H, "cuda")
        y = ops.bi_forward_ex(None, *d[:5], H, ws=ws, lens=lens, w_is_ew=ew)
        g = ops.bi_backward_ex(None, *d, H, ws=ws, lens=lens, w_is_ew=ew)
        outs[mode] = [host(y)] + [host(t) for t in g]
    for n, a_, b_ in zip(("y", "gr", "gk", "gv", "gw", "gu"), outs["1"], outs["0"]):
        assert np.array_equal(a_, b_), (n, float(np.abs(a_ - b_).max()))
