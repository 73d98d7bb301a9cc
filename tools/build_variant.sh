#!/bin/bash
# A/B variant of the library from the working tree: bash tools/build_variant.sh <name> ["-DFLAG ..."] [file.hip ...]
# Compiles the named sources (default: the two chunked kernels) with the extra flags, takes every other object from the cache
# build_ab/_obj (filled on first use from the working tree, refreshed with `bash tools/build_variant.sh --refresh`), links
# build_ab/<name>/lib.so.  Use with tools/abn.sh / RWKV_AMD_LIB.
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -w"
SRCS="wkv6_scan wkv6_chunk wkv6_chunk_bwd12k wkv6_mix wkv6_api"
mkdir -p build_ab/_obj
refresh() { for s in $SRCS; do hipcc $FLAGS -c rwkv_lm_ext_amd/csrc/$s.hip -o build_ab/_obj/$s.o & done; wait; }
if [ "$1" = "--refresh" ]; then refresh; rm -f build_ab/_obj/.key; exit 0; fi
name=$1; extra=$2; shift; shift || true
files=${@:-"wkv6_chunk wkv6_chunk_bwd12k"}
# the cached objects are those of ONE state of the sources: any header, any source or the flags changing invalidates all of them (a stale
# wkv6_api.o linked against kernels compiled for another ScanArgs layout passes a mismatched kernarg struct)
key=$(cat rwkv_lm_ext_amd/csrc/*.h include/wkv6_amd.h rwkv_lm_ext_amd/csrc/*.hip | sha256sum | cut -c1-16)-$(echo "$FLAGS" | sha256sum | cut -c1-8)
[ "$(cat build_ab/_obj/.key 2>/dev/null)" = "$key" ] || { refresh; echo "$key" > build_ab/_obj/.key; }
for s in $SRCS; do [ -f build_ab/_obj/$s.o ] || { refresh; echo "$key" > build_ab/_obj/.key; break; }; done
mkdir -p build_ab/$name
objs=""
for s in $SRCS; do
    if echo " $files " | grep -q " $s "; then
        hipcc $FLAGS $extra -c rwkv_lm_ext_amd/csrc/$s.hip -o build_ab/$name/$s.o &
        objs="$objs build_ab/$name/$s.o"
    else
        objs="$objs build_ab/_obj/$s.o"
    fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$name/lib.so $objs
rm -f build_ab/$name/*.o
ls -la build_ab/$name/lib.so
