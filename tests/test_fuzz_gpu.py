"""A fixed-seed 50-case slice of tools/fuzz_gpu.py in the -m gpu suite: the chunked MFMA kernels against the exact
scan kernels over random shapes, decay regimes, amplitudes, ragged wkv6_bi masks and directly passed row lengths (0-token rows
included), partial reversals and the pair launch, in both workgroup modes of the chunked kernels (the suite's small shapes run two
workgroups per (batch, head) by default; WKV6_SPLIT=0 puts them on the one-workgroup mode of the benched shapes, whose hand-over
tags are polled without a bound in the product build)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("split", ["auto", "0"], ids=["two_workgroups_per_pair", "one_workgroup_per_pair"])
def test_fuzz_slice(monkeypatch, split):
    assert torch.cuda.is_available(), "the gpu suite needs a GPU"
    import fuzz_gpu
    if split != "auto":
        monkeypatch.setenv("WKV6_SPLIT", split)
    bad = fuzz_gpu.run(50, seed=31337, verbose=False)
    assert bad == [], bad[:5]
