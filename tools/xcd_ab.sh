#!/bin/bash
# XCD row-remap A/B (round 6): bash tools/xcd_ab.sh   (needs build_ab/{xcd0,xcd1}/lib.so: tools/build_variant.sh xcd1 "-DWKV6_XCD_REMAP=1" wkv6_chunk wkv6_chunk_bwd12k)
export RWKV_AMD_NO_SELFTEST=1
for i in 1 2 3; do for v in xcd0 xcd1; do ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_CLOCKS=1 python tools/time_ops.py --iters 60 2>&1 | grep -v amdgpu.ids; done; done
