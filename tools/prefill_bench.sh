#!/bin/bash
# Forward-only (inference prefill) timing of few long sequences with and without the two-level scan over T (GPU box).
for shape in "1 16384" "1 4096" "2 8192" "4 2048"; do
    set -- $shape
    for ts in 0 auto; do
        if [ $ts = auto ]; then unset WKV6_TSPLIT; else export WKV6_TSPLIT=$ts; fi
        ABL_NAME="B=$1 T=$2 H=32 tsplit=$ts" RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py --only fwd --no-ckpt --B $1 --T $2 --iters 50 2>&1 | grep -v amdgpu.ids
    done
done
