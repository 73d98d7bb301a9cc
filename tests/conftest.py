import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_golden_mid(name):
    """The T = 160 fixtures of oracle/gen_golden_medium.py: inputs stored as raw bf16 bits (`<name>_bf16`, uint16) are returned
    as the float32 values they stand for."""
    out = {}
    for k, v in load_golden(name).items():
        if k.endswith("_bf16"):
            out[k[:-5]] = (v.astype(np.uint32) << 16).view(np.float32)
        else:
            out[k] = v
    return out


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


@pytest.fixture(scope="session")
def oracle():
    from oracle import wkv6_oracle
    wkv6_oracle.build()
    return wkv6_oracle
