#!/usr/bin/env python3
"""Per-launch duration and in-kernel shader clock of the config-2 forward / backward kernels from a cold GPU and after idle gaps
(VERDICT r5 item 1d).  Un-profiled: both figures come from the library's clock ring (wkv6_set_clock_ring: wave 0 of 64 workgroups
stamps {s_memtime, s_memrealtime} at its start and end; duration = max(end) - min(start), clock = d(memtime) / d(memrealtime)).

    python tools/dvfs_transient.py [--seconds 3] [--gaps 0.1,1,5,20] > profiles/r06_dvfs_transient.txt
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402  (build_workload)
from rwkv_lm_ext_amd import wkv6_op              # noqa: E402


def run(fwd, bwd, iters, dev, warm_iters=0, gap_ms=None):
    """[warm_iters back-to-back iterations, synchronize, gap_ms of idle,] then `iters` back-to-back fwd+bwd iterations; per-launch records of
    those, oldest first.  The ring is allocated first: nothing but the gap sits between the warm launches and the recorded ones."""
    with wkv6_op.ClockProbe(dev, n_slots=64, n_launches=iters) as probe:
        torch.cuda.synchronize()
        for _ in range(warm_iters):
            fwd()
            bwd()
        if gap_ms is not None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < gap_ms:     # the host spins: the device idles for `gap_ms`
                pass
        for _ in range(iters):
            fwd()
            bwd()
        torch.cuda.synchronize()
        return probe.read()


def table(rec, rows, label):
    f_us, f_g, b_us, b_g = rec["fwd_us_launches"], rec["fwd_ghz_launches"], rec["bwd_us_launches"], rec["bwd_ghz_launches"]
    t = 0.0
    print(f"# {label}")
    print("# iter   t_ms   fwd_us  fwd_GHz  fwd_cyc/grp   bwd_us  bwd_GHz  bwd_cyc/stage")
    show = set(rows)
    for i in range(len(b_us)):
        if i in show:
            print(f"{i:6d} {t:7.1f}  {f_us[i]:7.1f}  {f_g[i]:6.3f}  {f_us[i] * f_g[i] * 1e3 / 64:9.0f}    {b_us[i]:7.1f}  {b_g[i]:6.3f}  {b_us[i] * b_g[i] * 1e3 / 128:9.0f}")
        t += (f_us[i] + b_us[i]) * 1e-3


def mean(xs):
    return sum(xs) / len(xs)


def summarise(rec, label, tail=200):
    f, b = rec["fwd_us_launches"], rec["bwd_us_launches"]
    fs, bs = mean(f[-tail:]), mean(b[-tail:])
    step = [x + y for x, y in zip(f, b)]
    ss = fs + bs
    worst = max(range(len(step)), key=lambda i: step[i])
    settled = next((i for i in range(len(step)) if all(abs(s - ss) < 0.02 * ss for s in step[i:i + 32])), None)
    t_settled = sum(step[:settled]) * 1e-3 if settled is not None else None
    print(f"# {label}: steady (last {tail}) fwd {fs:.1f} us  bwd {bs:.1f} us  step {ss:.1f} us | first launch step {step[0]:.1f} us | "
          f"worst step {step[worst]:.1f} us at iter {worst} (+{100 * (step[worst] / ss - 1):.1f} %) | "
          f"within 2 % of steady for 32 iterations from iter {settled} (t = {t_settled if t_settled is None else round(t_settled, 1)} ms)")
    for w0, w1 in ((0, 20), (5, 25), (20, 40), (64, 84), (69, 89)):
        if w1 <= len(step):
            print(f"#   a 20-step window at iters {w0}..{w1 - 1} would report {mean(step[w0:w1]):.1f} us per step ({100 * (mean(step[w0:w1]) / ss - 1):+.1f} % vs steady)")
    return ss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--gaps", default="0.1,1,5,20")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fwd, bwd = bench.build_workload("wkv6", dev)[:2]
    fwd(); bwd()
    torch.cuda.synchronize()
    print(f"# {torch.cuda.get_device_name(0)}; config 2 (B=8 T=4096 C=2048 H=32), one iteration = chunk_fwd_kernel + chunk_bwd12k_kernel")
    time.sleep(3.0)                                              # cold: 3 s idle
    iters = int(args.seconds / 0.58e-3)
    rec = run(fwd, bwd, iters, dev)
    rows = list(range(0, 130)) + list(range(130, 400, 10)) + list(range(400, iters, 100))
    table(rec, rows, f"A. from a cold GPU (3 s idle), {iters} back-to-back iterations")
    steady = summarise(rec, "A")
    n_warm = int(1.0 / 0.58e-3)
    for gap in [float(g) for g in args.gaps.split(",")]:
        rec = run(fwd, bwd, 400, dev, warm_iters=n_warm, gap_ms=gap)
        table(rec, list(range(0, 100)) + list(range(100, 400, 20)), f"B. 1 s of steady launches, synchronize, {gap} ms idle, 400 iterations")
        summarise(rec, f"B gap {gap} ms")
    # what bench.py's fence itself costs: steady launches, torch.cuda.synchronize(), launch at once
    rec = run(fwd, bwd, 200, dev, warm_iters=n_warm, gap_ms=0.0)
    table(rec, list(range(0, 40)) + list(range(40, 200, 20)), "C. 1 s of steady launches, synchronize, launch at once (bench.py's fence)")
    summarise(rec, "C")
    # D. no synchronize at all between the warm launches and the recorded ones (bench.py's pre-warm -> warm-up hand-over)
    rec = run(fwd, bwd, 200, dev, warm_iters=n_warm, gap_ms=None)
    table(rec, list(range(0, 200, 10)), "D. 1 s of steady launches, 200 more without any synchronize in between")
    summarise(rec, "D")


if __name__ == "__main__":
    main()
