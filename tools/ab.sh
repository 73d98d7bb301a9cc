#!/bin/bash
# Same-box A/B timing: bash tools/ab.sh libA.so libB.so [rounds]   (alternating runs of tools/time_ops.py)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
    ABL_NAME=A RWKV_AMD_LIB=$A RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py --only both --iters 60 2>&1 | grep -v amdgpu.ids
    ABL_NAME=B RWKV_AMD_LIB=$B RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py --only both --iters 60 2>&1 | grep -v amdgpu.ids
done
