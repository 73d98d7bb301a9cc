#!/bin/bash
# Build one librwkv6_amd.so per timing-only ablation switch and time forward/backward with each.
# Usage (GPU box, repo root): bash tools/ablate.sh outdir "NAME:-DFLAG1 -DFLAG2" "NAME2:..." ...
set -e
OUT=$1; shift
mkdir -p "$OUT"
SRC=rwkv_lm_ext_amd/csrc
for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    d="$OUT/$name"; mkdir -p "$d"
    for f in wkv6_scan wkv6_chunk wkv6_chunk_bwd12k wkv6_mix wkv6_api; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing $flags -c $SRC/$f.hip -o "$d/$f.o" &
    done
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/lib.so" "$d"/*.o
    ABL_NAME=$name RWKV_AMD_LIB="$d/lib.so" RWKV_AMD_NO_SELFTEST=1 python tools/time_ops.py ${TIME_ARGS:---only fwd} 2>&1 | grep -v amdgpu.ids
done
