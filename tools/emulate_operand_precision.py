#!/usr/bin/env python3
"""What would a single-operand (fp16) MFMA path do to the outputs?  (round 5)

fp64 numpy emulation of the chunked kernels' block algebra (csrc/wkv6_chunk.hip, wkv6_chunk_bwd12k.hip: 16-token blocks, one
reference point per block, the forward state S and the adjoint state G carried exactly) in which every MFMA operand that the
kernels derive in fp32 -- Rhat, Khat, the masked scores, dA, (E8 S), (E16m8 G) -- is rounded the way a candidate arithmetic would
round it before the product, and nothing else is (products and sums in fp64, i.e. the candidate at its best):

    exact    no operand rounding: the reference (checked against the C oracle by tests/test_emulation_cpu.py)
    split    bf16 hi + bf16 lo, products hi*hi + hi*lo + lo*hi          (what the kernels do: 3 MFMAs per product)
    fp16     one operand of 11 significant bits, ANY exponent            (fp16 with an ideal per-tile power-of-two scale: 1 MFMA)
    fp16x2   fp16 hi + fp16 lo, 3 products                                (same MFMA count as split; for reference)
    bf16     one bf16 operand                                            (8 significant bits: 1 MFMA)

r, k, v, gy are bf16 inputs and stay exact in every mode.  Outputs are rounded to bf16 (RNE) and held to the suite's contract against
RNE_bf16(exact): rel-rms <= 1e-3, <= 2 bf16 ulps anywhere, >= 95 % of the significant elements the correctly rounded value
(oracle/contract.py states it; the metric is restated here because tools/ must not import the oracle).

    python tools/emulate_operand_precision.py [--T 1024] [--heads 4]      # prints the table of profiles/r05_fp16_path.md
"""
import argparse

import numpy as np

BLK = 16
LW_MIN = -9.0


def bf16_round(x):
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def round_bits(x, bits):
    """RNE to `bits` significant bits at any exponent (fp64 in, fp64 out)."""
    x = np.asarray(x, np.float64)
    m, e = np.frexp(x)
    return np.ldexp(np.rint(m * 2.0 ** bits), e - bits)


class Mode:
    """An operand-rounding policy: parts(x) -> list of operand pieces whose cross products are kept by prod()."""

    def __init__(self, name):
        self.name = name

    def pieces(self, x):
        n = self.name
        if n == "exact":
            return [np.asarray(x, np.float64)]
        if n == "bf16":
            return [round_bits(x, 8)]
        if n == "fp16":
            return [round_bits(x, 11)]
        bits = 8 if n == "split" else 11
        hi = round_bits(x, bits)
        return [hi, round_bits(np.asarray(x, np.float64) - hi, bits)]

    def mm(self, a, b, a_exact=False, b_exact=False, ca=None, cb=None):
        """a @ b with the policy applied to the inexact operands; hi*hi + hi*lo + lo*hi for two-piece operands.
        ca / cb: the operand class of a / b (OPERAND_CLASSES) -- a per-class policy (PerOperand) rounds each class its own way."""
        pa = [np.asarray(a, np.float64)] if a_exact else self.pieces_of(a, ca)
        pb = [np.asarray(b, np.float64)] if b_exact else self.pieces_of(b, cb)
        out = 0.0
        for i, x in enumerate(pa):
            for j, y in enumerate(pb):
                if i + j <= 1:                       # lo*lo is dropped, as in the kernels
                    out = out + x @ y
        return out


OPERAND_CLASSES = ("Rhat", "Khat", "scores", "dA", "E8.S", "E16m8.G")     # every fp32-derived MFMA operand of the two kernels


def _pieces_of(self, x, cls):
    return self.pieces(x)


Mode.pieces_of = _pieces_of


class PerOperand(Mode):
    """ONE operand class rounded the candidate way (`fp16`: 11 bits at any exponent, `bf16`: 8 bits), every other class split bf16 hi + lo
    as the kernels do: would that class's split instructions and two of its three MFMAs be removable without touching the contract?"""

    def __init__(self, cls, candidate):
        super().__init__(f"{cls}:{candidate}")
        self.cls, self.cand, self.rest = cls, Mode(candidate), Mode("split")

    def pieces_of(self, x, cls):
        assert cls in OPERAND_CLASSES, cls
        return (self.cand if cls == self.cls else self.rest).pieces(x)


def block_factors(w):
    """per-token clamped log-decay and the block's frames: c (exclusive cumulative), c8, c16"""
    lw_true = -np.exp(w.astype(np.float64))
    lw = np.maximum(lw_true, LW_MIN)
    c = np.concatenate([np.zeros((1, w.shape[1])), np.cumsum(lw, 0)], 0)       # c[a] = sum_{s<a} lw_s, a = 0..n
    n = w.shape[0]
    c8 = c[min(8, n)]
    return lw_true, lw, c, c8, c[n]


def forward(mode, r, k, v, w, u, s0=None):
    """One head: r, k, v, w [T, 64], u [64]; returns y [T, 64] (fp64) and the list of block-entry states S[i][j]."""
    T, N = r.shape
    S = np.zeros((N, N)) if s0 is None else np.array(s0, np.float64)            # [key i][value j]
    y = np.zeros((T, N))
    states = []
    for t0 in range(0, T, BLK):
        sl = slice(t0, min(t0 + BLK, T))
        rb, kb, vb, wb = (x[sl].astype(np.float64) for x in (r, k, v, w))
        n = rb.shape[0]
        states.append(S.copy())
        _, lw, c, c8, c16 = block_factors(wb)
        Rh = rb * np.exp(c[:n] - c8)
        Kh = kb * np.exp(c8 - c[1:n + 1])
        A = np.tril(mode.mm(Rh, Kh.T, ca="Rhat", cb="Khat"), -1) + np.diag(np.sum(rb * u * kb, 1))      # masked scores + bonus diagonal
        y[sl] = mode.mm(A, vb, b_exact=True, ca="scores") + mode.mm(Rh, np.exp(c8)[:, None] * S, ca="Rhat", cb="E8.S")
        S = np.exp(c16)[:, None] * S + np.exp(c16 - c8)[:, None] * mode.mm(Kh.T, vb, b_exact=True, ca="Khat")
    return y, states, S


def backward(mode, r, k, v, w, u, gy, states):
    """Adjoint of forward() block by block (csrc/wkv6_chunk_bwd12k.hip header); returns gr, gk, gv, gw [T, 64], gu [64]."""
    T, N = r.shape
    G = np.zeros((N, N))                                                         # dL/d(state after the block) [i][j]
    gr, gk, gv, gw = (np.zeros((T, N)) for _ in range(4))
    gu = np.zeros(N)
    Rc = np.zeros(N)                                                             # suffix sum of (a_s - b_s) over later blocks
    nb = (T + BLK - 1) // BLK
    for b in range(nb - 1, -1, -1):
        sl = slice(b * BLK, min((b + 1) * BLK, T))
        rb, kb, vb, wb, gb = (x[sl].astype(np.float64) for x in (r, k, v, w, gy))
        n = rb.shape[0]
        S = states[b]
        lw_true, lw, c, c8, c16 = block_factors(wb)
        fR = np.exp(c[:n] - c8)
        fK = np.exp(c8 - c[1:n + 1])
        Rh, Kh = rb * fR, kb * fK
        E8, E16, E16m8 = np.exp(c8), np.exp(c16), np.exp(c16 - c8)
        dA = np.tril(gb @ vb.T, -1)                                               # exact operands
        vg = np.sum(gb * vb, 1)
        A = np.tril(mode.mm(Rh, Kh.T, ca="Rhat", cb="Khat"), -1) + np.diag(np.sum(rb * u * kb, 1))
        GE = E16m8[:, None] * G
        gv[sl] = mode.mm(A.T, gb, b_exact=True, ca="scores") + mode.mm(Kh, GE, ca="Khat", cb="E16m8.G")
        dq = fR * (mode.mm(dA, Kh, ca="dA", cb="Khat") + E8 * mode.mm(gb, S.T, a_exact=True, cb="E8.S"))   # (E8 applied to the result, as the kernel does)
        dk = fK * (mode.mm(dA.T, Rh, ca="dA", cb="Rhat") + mode.mm(vb, GE.T, a_exact=True, cb="E16m8.G"))
        gr[sl] = dq + vg[:, None] * u * kb
        gk[sl] = dk + vg[:, None] * u * rb
        gu += np.sum(vg[:, None] * rb * kb, 0)
        at, bt = rb * dq, kb * dk
        dl = at - bt
        sfx = np.cumsum(dl[::-1], 0)[::-1]                                         # inclusive suffix sums inside the block
        # gw multiplier: the true lw, times d_true / d_clamped where the clamp is active
        lwn = lw_true * np.exp(np.minimum(lw_true - LW_MIN, 0.0))
        gw[sl] = (Rc + (sfx - dl) - bt) * lwn
        Rc = Rc + sfx[0]
        G = E16[:, None] * G + E8[:, None] * mode.mm(Rh.T, gb, b_exact=True, ca="Rhat")
    return gr, gk, gv, gw, gu


def report(out, ref, floor=1e-3):
    """(rel_rms, frac_not_correctly_rounded, max_ulps) against RNE_bf16(ref) -- oracle/contract.py: bf16_report, restated."""
    out = bf16_round(np.asarray(out, np.float32)).astype(np.float64)
    ref = np.asarray(ref, np.float64)
    want = bf16_round(ref.astype(np.float32)).astype(np.float64)
    d = out - want
    rms = max(np.sqrt(np.mean(ref ** 2)), floor)
    rel_rms = float(np.sqrt(np.mean(d ** 2)) / rms)
    fl = max(1e-2 * np.abs(ref).max(), floor)
    big = np.abs(ref) >= fl
    off = float(np.mean(d[big] != 0)) if big.any() else 0.0
    ulp = np.maximum(np.abs(ref), fl) * 2.0 ** -7
    return rel_rms, off, float((np.abs(d) / ulp).max())


def synth(T, heads, kind, seed=0):
    """bench.py: synth restated for [T, heads * 64] (config 2's distributions), or the suite's `stress` decays."""
    rng = np.random.default_rng(seed)
    C = heads * 64
    bf = lambda x: bf16_round(x.astype(np.float32)).astype(np.float64)
    r, k, v = (bf(rng.standard_normal((T, C)) * 0.5) for _ in range(3))
    if kind == "init":
        ramp = np.array([-6 + 5 * (n / (C - 1)) ** (0.7 + 1.3 * 0.5) for n in range(C)])
        w = bf(ramp[None, :] + 0.1 * rng.standard_normal((T, C)))
    else:
        w = bf(-1.0 + 0.5 * rng.standard_normal((T, C)))
    u = bf(rng.standard_normal(C) * 0.3)
    gy = bf(rng.standard_normal((T, C)))
    return r, k, v, w, u, gy


def run(T=1024, heads=4, kinds=("init", "stress"), modes=("split", "fp16x2", "fp16", "bf16")):
    rows = []
    for kind in kinds:
        r, k, v, w, u, gy = synth(T, heads, kind)
        outs = {}
        for name in ("exact",) + tuple(modes):
            m = name if isinstance(name, Mode) else Mode(name)
            name = m.name
            acc = {n: [] for n in ("y", "gr", "gk", "gv", "gw", "gu")}
            for h in range(heads):
                s = slice(64 * h, 64 * h + 64)
                y, states, _ = forward(m, r[:, s], k[:, s], v[:, s], w[:, s], u[s])
                # (the backward restarts from the same arithmetic's forward states, as the kernels' checkpoints + rebuilds do)
                g = backward(m, r[:, s], k[:, s], v[:, s], w[:, s], u[s], gy[:, s], states)
                for n, val in zip(acc, (y,) + g):
                    acc[n].append(val)
            outs[name] = {n: np.concatenate([np.atleast_2d(x) for x in vals], -1) for n, vals in acc.items()}
        for name in (m.name if isinstance(m, Mode) else m for m in modes):
            for n in ("y", "gr", "gk", "gv", "gw"):
                rms, off, ulps = report(outs[name][n], outs["exact"][n], floor=0.1 if n == "gw" else 1e-3)
                ok = rms <= 1e-3 and ulps <= 2.0 and off <= (0.10 if n == "gw" else 0.05)
                rows.append((kind, name, n, rms, off, ulps, ok))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--per-operand", action="store_true",
                    help="ONE operand class at a time in fp16 / plain bf16, the rest split (profiles/r06_ceiling.md)")
    args = ap.parse_args()
    modes = ("split", "fp16x2", "fp16", "bf16")
    if args.per_operand:
        modes = ("split",) + tuple(PerOperand(c, cand) for cand in ("fp16", "bf16") for c in OPERAND_CLASSES)
    rows = run(args.T, args.heads, modes=modes)
    print(f"# T = {args.T}, {args.heads} heads; contract: rel-rms <= 1e-3, <= 2 ulp, >= 95 % correctly rounded (gw: >= 90 % at long T, as the suite)")
    print(f"{'decays':8s} {'operands':14s} {'tensor':6s} {'rel-rms':>10s} {'% not correctly rounded':>24s} {'max ulp':>8s}  contract")
    for kind, name, n, rms, off, ulps, ok in rows:
        print(f"{kind:8s} {name:14s} {n:6s} {rms:10.2e} {100 * off:24.1f} {ulps:8.2f}  {'pass' if ok else 'FAIL'}")
    if args.per_operand:
        print("# summary: does the class stay inside the contract on every tensor (both decay regimes)?  worst % not correctly rounded / worst ulp")
        for m in modes[1:]:
            mine = [r for r in rows if r[1] == m.name and r[2] != "gw"]
            gw = [r for r in rows if r[1] == m.name and r[2] == "gw"]
            print(f"#   {m.name:14s} y/gr/gk/gv: {'pass' if all(r[6] for r in mine) else 'FAIL'} ({100 * max(r[4] for r in mine):.1f} %, {max(r[5] for r in mine):.2f} ulp)"
                  f"   gw: {'pass' if all(r[6] for r in gw) else 'FAIL'} ({100 * max(r[4] for r in gw):.1f} %, {max(r[5] for r in gw):.2f} ulp)")


if __name__ == "__main__":
    main()
