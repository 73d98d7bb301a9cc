#!/bin/bash
# Build the library with the fully unrolled, unpinned forward consumer body (-DWKV6_FWD_UNROLL) and run the kernel parity
# tests against it (GPU box, repo root).  DESIGN.md 4.2: the wrong y this build once produced was the mixed-shape MFMA
# accumulation hazard; with one accumulator per MFMA shape it passes.
set -e
OUT=${1:-gpurun_out/unrolled}
mkdir -p "$OUT"
SRC=rwkv_lm_ext_amd/csrc
for f in wkv6_scan wkv6_chunk wkv6_chunk_bwd12 wkv6_mix wkv6_api; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -DWKV6_FWD_UNROLL -Iinclude -c $SRC/$f.hip -o "$OUT/$f.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/librwkv6_amd_unrolled.so" "$OUT"/*.o
RWKV_AMD_LIB="$OUT/librwkv6_amd_unrolled.so" python -m pytest tests/test_wkv6_gpu.py -q -m gpu -k "chunk or golden or selftest or bf16"
