#!/bin/bash
# bytes-for-clock experiment: bash tools/clock_ab.sh   (needs build_ab/{head2,halfck}/lib.so: tools/build_variant.sh head2 "" wkv6_chunk;
# tools/build_exp_variant.sh halfck)
export RWKV_AMD_NO_SELFTEST=1
for i in 1 2 3; do for v in head2 halfck; do ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_CLOCKS=1 python tools/time_ops.py --iters 60 2>&1 | grep -v amdgpu.ids; done; done
