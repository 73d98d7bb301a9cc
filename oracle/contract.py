"""The parity contract of the suite as functions (TEST INFRASTRUCTURE, like everything under oracle/: imported by tests/,
__graft_entry__.smoke() and nothing else).

fp32 I/O:  max|out - ref| / max|ref| <= F32_TOL.
bf16 I/O:  against RNE_bf16(oracle): rms error <= BF16_RMS of rms(ref), <= BF16_ULPS bf16 ulps anywhere, >= BF16_EXACT of the
           significant elements exactly the correctly rounded value (stricter than BASELINE.json's 1e-3 bf16 / 1e-5 fp32).
"""
import numpy as np

F32_TOL = 1e-5
BF16_RMS, BF16_ULPS, BF16_EXACT = 1e-3, 2.0, 0.95


def max_norm_err(a, b, floor=1e-3):
    """max|a-b| / max(max|b|, floor)  -- the metric every fp32 tolerance in this suite is stated in.
    `floor` keeps an all-zero expectation (gw at T <= 2) from turning fp32 cancellation noise into inf;
    test tensors are O(1)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def bf16_round(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (numpy)."""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_report(out, ref, floor=1e-3):
    """Compare a bf16 kernel output (as float32) with the fp64/fp32 oracle result.

    Returns (rel_rms, frac_off, max_ulps): error against RNE_bf16(oracle), normalised by the rms of the
    oracle; fraction of the significant elements (>= 1 % of max|ref|) that are not the correctly rounded value; largest deviation in bf16 ulps
    of max(|ref|, 1e-2*max|ref|).
    """
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref, np.float64)
    want = bf16_round(ref.astype(np.float32)).astype(np.float64)
    d = out - want
    rms = max(np.sqrt(np.mean(ref ** 2)), floor)
    rel_rms = float(np.sqrt(np.mean(d ** 2)) / rms)
    floor = max(1e-2 * np.abs(ref).max(), floor)
    big = np.abs(ref) >= floor            # elements below 1 % of the tensor scale sit in fp32 noise
    frac_off = float(np.mean(d[big] != 0)) if big.any() else 0.0
    ulp = np.maximum(np.abs(ref), floor) * 2.0 ** -7
    max_ulps = float((np.abs(d) / ulp).max())
    return rel_rms, frac_off, max_ulps


def bf16_ok(out, ref, floor=1e-3, exact=BF16_EXACT):
    """(passed, message) of the bf16 contract for one tensor."""
    rms, off, ulps = bf16_report(out, ref, floor)
    ok = rms <= BF16_RMS and ulps <= BF16_ULPS and off <= 1 - exact
    return ok, f"bf16 rel-rms {rms:.2e}, max {ulps:.2f} ulp, {off * 100:.1f}% not correctly rounded"
