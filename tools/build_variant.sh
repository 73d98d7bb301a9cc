#!/bin/bash
# Cross-compile a variant library in this container: bash tools/build_variant.sh <name> "<extra flags>"  ->  build_ab/<name>/lib.so
# (build_ab/ is git-ignored but travels to the GPU box with the snapshot; use with RWKV_AMD_LIB=build_ab/<name>/lib.so)
set -e
name=$1; flags=$2
d=build_ab/$name; mkdir -p $d
for f in wkv6_scan wkv6_chunk wkv6_chunk_bwd12k wkv6_mix wkv6_api; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing $flags -c rwkv_lm_ext_amd/csrc/$f.hip -o $d/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/lib.so $d/*.o
echo built $d/lib.so
