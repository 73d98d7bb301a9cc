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


from oracle.contract import max_norm_err, bf16_round, bf16_report      # noqa: E402,F401  (the contract lives beside the oracle)


@pytest.fixture(scope="session")
def oracle():
    from oracle import wkv6_oracle
    wkv6_oracle.build()
    return wkv6_oracle
