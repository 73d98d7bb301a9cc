#!/bin/bash
# Cross-compile the library of ANOTHER source tree (e.g. an earlier round extracted with `git archive <rev> | tar -x -C build_ab/<rev>_tree`)
# for same-box A/B runs:  bash tools/build_tree_variant.sh <name> <tree>/rwkv_lm_ext_amd/csrc "<extra flags>"  ->  build_ab/<name>/lib.so
set -e
name=$1; src=$2; flags=$3
d=build_ab/$name; mkdir -p $d; rm -f $d/*.o
for f in $src/*.hip; do
    b=$(basename $f .hip)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing $flags -c $f -o $d/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/lib.so $d/*.o
echo built $d/lib.so
