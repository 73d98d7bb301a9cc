"""tools/emulate_operand_precision.py (the fp64 emulation of the chunked kernels' block algebra behind profiles/r05_fp16_path.md):
its `exact` mode -- no operand rounding -- must BE the operator, i.e. agree with the C oracle to fp64 / fp32-interface noise; and the
`split` mode (what the kernels do) must meet the suite's bf16 contract on the emulation's own inputs while the single-operand modes
are what the committed table says they are."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_exact_mode_is_the_oracle(oracle):
    import emulate_operand_precision as em
    T, heads = 100, 2                                     # a partial last block
    for kind in ("init", "stress"):
        r, k, v, w, u, gy = em.synth(T, heads, kind, seed=3)
        f = lambda x: np.ascontiguousarray(x[None], np.float32)
        yo = oracle.forward(f(r), f(k), f(v), f(w), u.reshape(heads, 64).astype(np.float32))
        og = oracle.backward(f(r), f(k), f(v), f(w), u.reshape(heads, 64).astype(np.float32), f(gy))
        m = em.Mode("exact")
        for h in range(heads):
            s = slice(64 * h, 64 * h + 64)
            y, states, _ = em.forward(m, r[:, s], k[:, s], v[:, s], w[:, s], u[s])
            gr, gk, gv, gw, gu = em.backward(m, r[:, s], k[:, s], v[:, s], w[:, s], u[s], gy[:, s], states)
            for name, got, ref in (("y", y, yo[0][:, s]), ("gr", gr, og["gr"][0][:, s]), ("gk", gk, og["gk"][0][:, s]),
                                   ("gv", gv, og["gv"][0][:, s]), ("gw", gw, og["gw"][0][:, s]), ("gu", gu, og["gu_b"][0][s])):
                scale = max(float(np.abs(ref).max()), 1e-3)
                assert float(np.abs(got - ref).max()) <= 2e-6 * scale, (kind, h, name)     # the oracle returns float32


def test_split_operands_meet_the_contract_and_single_fp16_does_not():
    import emulate_operand_precision as em
    rows = em.run(T=256, heads=2, kinds=("init",), modes=("split", "fp16"))
    by = {(name, n): (rms, off, ulps, ok) for _, name, n, rms, off, ulps, ok in rows}
    assert all(by[("split", n)][3] for n in ("y", "gr", "gk", "gv", "gw"))
    # one 11-bit operand per product: rel-rms passes, the share of correctly rounded outputs does not (profiles/r05_fp16_path.md)
    assert not all(by[("fp16", n)][3] for n in ("y", "gr", "gk", "gv"))
