#!/bin/bash
# Same-box A/B/C/... timing of variant libraries: bash tools/abn.sh <rounds> <name>...   (build_ab/<name>/lib.so, alternating runs of tools/time_ops.py)
N=$1; shift
for i in $(seq $N); do
    for v in "$@"; do
        ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py --only both --iters 60 2>&1 | grep -v amdgpu.ids
    done
done
