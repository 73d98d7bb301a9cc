#!/bin/bash
# Timing-only experiment builds of the chunked kernels that do NOT live in the product sources: the sources are
# copied to build_ab/<name>_src, patched there, and compiled to build_ab/<name>/lib.so (use with RWKV_AMD_LIB; results are WRONG).
#   bash tools/build_exp_variant.sh nostore      # backward: every gradient store dropped by the hardware bounds check (zero-sized resources)
#   bash tools/build_exp_variant.sh noloads      # backward: ... and the r, k, v, w, gy loads too (zero-sized resources return 0): only checkpoints move
#   bash tools/build_exp_variant.sh fwd_nostore  # forward: y and checkpoint stores dropped
#   bash tools/build_exp_variant.sh fwd_noloads  # forward: ... and the r, k, v, w loads too: no memory traffic at all
#   bash tools/build_exp_variant.sh halfck       # both: HALF of every checkpoint's bytes stored by the forward / fetched by the backward (wrong results): what a
#                                                #   checkpoint of 2 B per token-channel (a compressed format, a 128-token spacing) could buy at most
#   bash tools/build_exp_variant.sh full         # both: the producers' past-the-end selects compiled out (right only where every stage / group is full:
#                                                #   T a multiple of 64 and no per-row lengths -- config 2): what a tail-free instantiation would buy
set -e
name=$1; flags=$2
src=build_ab/${name}_src; rm -rf $src; mkdir -p $src/rwkv_lm_ext_amd; cp -r rwkv_lm_ext_amd/csrc $src/rwkv_lm_ext_amd/; cp -r include $src/
f=$src/rwkv_lm_ext_amd/csrc/wkv6_chunk_bwd12k.hip
case $name in
  nostore) python3 - "$f" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "    const rsrc_t rs_gr = make_rsrc(ogr, nbytes), rs_gk = make_rsrc(ogk, nbytes), rs_gv = make_rsrc(ogv, nbytes), rs_gw = make_rsrc(ogw, nbytes);"
assert old in s
s = s.replace(old, "    const rsrc_t rs_gr = make_rsrc(ogr, 0u), rs_gk = make_rsrc(ogk, 0u), rs_gv = make_rsrc(ogv, 0u), rs_gw = make_rsrc(ogw, 0u);")
open(p, "w").write(s)
PY
  ;;
  noloads) python3 - "$f" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "    const unsigned nbytes = ntok > 0 ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u;"
assert old in s
s = s.replace(old, "    const unsigned nbytes = 0u;")
open(p, "w").write(s)
PY
  ;;
  fwd_nostore|fwd_noloads) python3 - "$src/rwkv_lm_ext_amd/csrc/wkv6_chunk.hip" "$name" <<'PY'
import sys
p, name = sys.argv[1], sys.argv[2]; s = open(p).read()
old = "(!STATE_ONLY && a.y && ntok > 0) ? (unsigned)(ntok - 1) * a.C * 2u + 128u : 0u);"
assert old in s
s = s.replace(old, "0u);")
old = "a.ckpt ? nst * 16384u : 0u);"
assert old in s
s = s.replace(old, "0u);")
if name == "fwd_noloads":
    assert "ntok > 0 ? span * 2 + 128 : 0" in s and "ntok > 0 ? span * 4 + 256 : 0" in s
    s = s.replace("ntok > 0 ? span * 2 + 128 : 0", "0").replace("ntok > 0 ? span * 4 + 256 : 0", "0")
open(p, "w").write(s)
PY
  ;;
  halfck) python3 - "$src/rwkv_lm_ext_amd/csrc" <<'PY'
import sys
d = sys.argv[1]
p = d + "/wkv6_chunk.hip"; s = open(p).read()
old = """                    for (int wb = 0; wb < 4; ++wb)
                        __builtin_amdgcn_raw_buffer_store_b128(ckd[wb], rs_ck, ck_off + wb * 4096, 0, 2 /* slc: streaming */);"""
assert old in s
open(p, "w").write(s.replace(old, old.replace("wb < 4", "wb < 2")))
p = d + "/wkv6_chunk_bwd12k.hip"; s = open(p).read()
old = """            for (int jt = 0; jt < 4; ++jt) {
                const float4 t = buf_load16f(rs_ck, off + jt * 1024u);
                CK[jt] = f4v{t.x, t.y, t.z, t.w};
            }"""
assert old in s
open(p, "w").write(s.replace(old, old.replace("jt < 4", "jt < 2") + "\n            CK[2] = CK[0]; CK[3] = CK[1];"))
PY
  ;;
  full) python3 - "$src/rwkv_lm_ext_amd/csrc" <<'PY'
import sys
d = sys.argv[1]
p = d + "/wkv6_chunk_bwd12k.hip"; s = open(p).read()
old = "const bool valid = sk * STG + pb * BLK + 2 * tq + tt < ntok_k;"
assert old in s
open(p, "w").write(s.replace(old, "const bool valid = true;"))
p = d + "/wkv6_chunk.hip"; s = open(p).read()
old = "const bool valid = grp * GRP + wv * BLK + 4 * tq + tt < ntok;"
assert old in s
open(p, "w").write(s.replace(old, "const bool valid = true;"))
PY
  ;;
  *) echo "unknown experiment $name"; exit 1;;
esac
d=build_ab/$name; mkdir -p $d; rm -f $d/*.o
for c in $src/rwkv_lm_ext_amd/csrc/*.hip; do
    b=$(basename $c .hip)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing $flags -c $c -o $d/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $d/lib.so $d/*.o
echo built $d/lib.so
