#!/bin/bash
# Forward role -> SIMD pairing A/B: bash tools/pairing_ab.sh > gpurun_out/.../pairing.txt
# needs build_ab/{pair0,pair1,pair0s,pair1s,pair0c,pair1c}/lib.so (tools/build_variant.sh <name> "-DWKV6_FWD_PAIR=n [-DWKV6_STAMP|-DWKV6_CLOCK]" wkv6_chunk wkv6_api)
export RWKV_AMD_NO_SELFTEST=1
echo "== HW_ID read-back, config 2 (8-wave workgroups)"
for v in pair0c pair1c; do echo "-- $v"; RWKV_AMD_LIB=build_ab/$v/lib.so python tools/hwid_map.py 2>&1 | grep -v amdgpu.ids; done
echo "== HW_ID read-back, configs[4] chunk shape B=4 T=2048 (two 6-wave workgroups per (batch, head))"
for v in pair0c pair1c; do echo "-- $v"; RWKV_AMD_LIB=build_ab/$v/lib.so python tools/hwid_map.py --B 4 --T 2048 2>&1 | grep -v amdgpu.ids; done
echo "== config 2, alternating processes (fwd_ms / bwd_ms: the kernel alone in a loop; step_*: inside fwd+bwd steps, from the clock ring)"
for i in 1 2 3; do for v in pair0 pair1; do ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_CLOCKS=1 python tools/time_ops.py --iters 60 2>&1 | grep -v amdgpu.ids; done; done
echo "== configs[4] chunk shape B=4 T=2048, alternating processes"
for i in 1 2 3; do for v in pair0 pair1; do ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_CLOCKS=1 python tools/time_ops.py --B 4 --T 2048 --iters 200 2>&1 | grep -v amdgpu.ids; done; done
echo "== per-role stamps (-DWKV6_STAMP builds), config 2"
for v in pair0s pair1s; do echo "-- $v"; ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_STAMP=1 python tools/time_ops.py --iters 20 2>&1 | grep -v amdgpu.ids | sed -n '1,12p'; done
echo "== per-role stamps, B=4 T=2048"
for v in pair0s pair1s; do echo "-- $v"; ABL_NAME=$v RWKV_AMD_LIB=build_ab/$v/lib.so WKV6_STAMP=1 python tools/time_ops.py --B 4 --T 2048 --iters 20 2>&1 | grep -v amdgpu.ids | sed -n '1,12p'; done
