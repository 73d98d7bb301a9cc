#!/usr/bin/env python3
"""Medium-size fixtures from the REFERENCE's CPU recurrence (fla/ops/rwkv6/recurrent_naive.py:8-36 + autograd), T = 160: long
enough to cross every structural boundary of the HIP kernels (16-token blocks, 32-token stages, 64-token groups / chunks, the
second checkpoint) with reference-generated values, which the T <= 48 vectors of oracle/gen_golden.py do not.

Run in the build container only (needs /root/reference):   python oracle/gen_golden_medium.py
Writes tests/golden/wkv6_mid.npz (plain), wkv6_mid_state.npz (per-sample initial state, gs, final state), wkv6_mid_bi.npz
(ragged rows).  Inputs are stored as raw bf16 bits (uint16), outputs as float32.  The oracle is checked against every output
before anything is written (same 2e-5 bound as gen_golden.py)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                                                   # noqa: E402  (sets up the reference imports)
import numpy as np                                                        # noqa: E402
import torch                                                              # noqa: E402

from oracle import wkv6_oracle as orc                                     # noqa: E402

N = 64


def bits(x):
    return x.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def save(name, inputs, outputs):
    path = os.path.join(gg.OUT, name + ".npz")
    np.savez_compressed(path, **{k + "_bf16": bits(v) for k, v in inputs.items()},
                        **{k: np.asarray(v, np.float32) for k, v in outputs.items()})
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.set_grad_enabled(True)
    T = 160
    # ---- plain
    inp = gg.make_inputs(20, 2, T, 2, "stress")
    y, g = gg.ref_fwd_bwd(*inp)
    gg.check_oracle("wkv6_mid", inp, y, g)
    save("wkv6_mid", dict(zip("r k v w u gy".split(), inp)), dict(y=y, **g))
    # ---- per-sample initial state (infctx flavour): gs and the final state too
    inp = gg.make_inputs(21, 2, T, 1, "init")
    gen = torch.Generator().manual_seed(121)
    s = gg.bf16r(torch.randn(2, 1, N, N, generator=gen) * 0.5)
    y, g = gg.ref_fwd_bwd(*inp, s0=s)
    s_final = gg.ref_final_state(*inp[:5], s0=s)
    gg.check_oracle("wkv6_mid_state", inp, y, g, s0=s)
    _, so = orc.forward(*(x.numpy() for x in inp[:5]), s.numpy(), return_state=True)
    print(f"  final state: oracle vs reference(probe) = {gg.close(so, s_final, 2e-5, 'final state'):.1e}")
    save("wkv6_mid_state", dict(zip("r k v w u gy".split(), inp), s=s), dict(y=y, s_final=s_final, **g))
    # ---- wkv6_bi, ragged: forward scan + reverse scan (u = 0) over [0 .. L_b]
    B, H = 3, 1
    inp = gg.make_inputs(22, B, T, H, "stress")
    r, k, v, w, u, gy = inp
    lens = [T, 97, 33]
    mask = torch.ones(B, T, dtype=torch.int32)
    for b, L in enumerate(lens):
        mask[b, L - 1:] = 0                                    # first zero at L - 1: tokens 0 .. L - 1 are scanned
    ys = torch.zeros(B, T, H * N)
    gs_ = {n: torch.zeros_like(x) for n, x in zip(("gr", "gk", "gv", "gw"), (r, k, v, w))}
    gs_["gu"] = torch.zeros_like(u)
    for b in range(B):
        L = int((mask[b] == 0).nonzero()[0]) + 1 if (mask[b] == 0).any() else T
        assert L == lens[b]
        sl = [x[b:b + 1, :L].contiguous() for x in (r, k, v, w)]
        y1, g1 = gg.ref_fwd_bwd(*sl, u, gy[b:b + 1, :L].contiguous())
        y2, g2 = gg.ref_fwd_bwd(*sl, u, gy[b:b + 1, :L].contiguous(), reverse=True, use_u=False)
        ys[b, :L] = (y1 + y2)[0]
        for n in ("gr", "gk", "gv", "gw"):
            gs_[n][b, :L] = (g1[n] + g2[n])[0]
        gs_["gu"] += g1["gu"]
    mn = mask.numpy()
    args = [x.numpy() for x in (r, k, v, w, u)]
    print(f"wkv6_mid_bi: oracle y = {gg.close(orc.bi_forward(mn, *args), ys, 2e-5, 'bi y'):.1e}")
    ob = orc.bi_backward(mn, *args, gy.numpy())
    for n in gs_:
        print(f"  oracle {n} = {gg.close(ob[n], gs_[n], 2e-5, 'bi ' + n):.1e}")
    save("wkv6_mid_bi", dict(zip("r k v w u gy".split(), inp)), dict(mask=mask.numpy().astype(np.float32), y=ys, **gs_))


if __name__ == "__main__":
    main()
