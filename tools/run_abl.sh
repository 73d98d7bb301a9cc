#!/bin/bash
# time fwd/bwd with each prebuilt variant library: bash tools/run_abl.sh "<time_ops args>" name1 name2 ...
ARGS=$1; shift
for n in "$@"; do
    ABL_NAME=$n RWKV_AMD_LIB=build_ab/$n/lib.so RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py $ARGS 2>&1 | grep -v amdgpu.ids
done
