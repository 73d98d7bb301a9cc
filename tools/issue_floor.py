#!/usr/bin/env python3
"""Issue floor of the chunked kernels' role loops from MEASURED per-class issue costs (round 5).

    python tools/issue_floor.py [--waves 3] [--ops]

Static opcode histogram per role loop from the gfx950 ISA (tools/issue_model.py: loops_of), priced with the SIMD issue throughput
tools/microbench/issue2.hip measured on MI355X per instruction class at 2 / 3 / 4 waves per SIMD (profiles/r05_issue2_microbench.txt):
  full rate   v_fma/mul/add/sub/mov/and/or/xor/add_u32/fmac/cmp (VGPR or literal operands)        2.2 (2 or 4 waves) / 2.7 (3 waves)
  half rate   v_max/min_f32, a VALU op with an SGPR source, v_cndmask_e64, v_cvt_pk_bf16_f32, shifts, v_perm, v_ldexp, every DPP
              form, packed f32, v_dot2c_f32_bf16                                                    4.3
  quarter     v_exp/rcp/log/rsq/sqrt, v_permlane16/32_swap                                          8.1
  MFMA        either shape                                                                         16   (and only 2 VALU issue per MFMA slot)
  LDS         ds_read_b128 16, ds_read_b64(_tr_b16) 8.2, ds_write_b64 24, ds_write_b128 ~52, ds_read/write_b32 8/16 (per SIMD, all four
              SIMDs issuing = the CU's 256 B/clk; the LDS pipe is shared by the CU)
The floor of a SIMD = sum over its resident waves (one of each role).  VALU and MFMA time add (beside a saturated MFMA stream VALU
issue drops to one per 8 cycles: an MFMA leaves room for 2 VALU instructions in its 16 cycles)."""
import argparse
import collections
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import issue_model as im                                            # noqa: E402

QUARTER = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_permlane")
HALF = ("v_max_f32", "v_min_f32", "v_cndmask", "v_cvt_pk", "v_cvt_f32_bf16", "v_lshl", "v_lshr", "v_ashr", "v_perm_b32", "v_ldexp", "v_pk_", "v_dot2",
        "v_bfe", "v_and_or", "v_lshl_or", "v_lshl_add", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_readlane", "v_readfirstlane", "v_max3", "v_med3")
LDS_COST = {"ds_read_b128": 16, "ds_read_b64": 8.2, "ds_read_b64_tr_b16": 8.2, "ds_read_b32": 8, "ds_read2_b32": 16, "ds_read2_b64": 32,
            "ds_write_b64": 24, "ds_write_b128": 52, "ds_write_b32": 16, "ds_write2_b32": 24, "ds_write2_b64": 52, "ds_read_b96": 32, "ds_write_b96": 40,
            "ds_read_u16": 8, "ds_bpermute_b32": 8}


def price(op, line, full):
    """(class, cycles)"""
    if op.startswith("v_mfma"):
        return "mfma", 16.0
    if op.startswith("v_"):
        if "dpp" in line or "row_" in line or "quad_perm" in line or "sdwa" in line:
            return "half", 4.3
        if op.startswith(QUARTER):
            return "quarter", 8.1
        if op.startswith(HALF):
            return "half", 4.3
        # an SGPR / constant-bus source halves the rate (v_mul_f32 v, s, v: 4.1-4.6 at any occupancy)
        args = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
        srcs = [a.strip() for a in args.split(",")[1:]]
        if any(a.startswith(("s[", "s")) and not a.startswith("src") and a[1:2].isdigit() or a.startswith("s[") or a == "vcc" or a == "exec" for a in srcs):
            return "half", 4.3
        return "full", full
    if op.startswith("ds_"):
        return "lds", LDS_COST.get(op, 8.0)
    return None, 0.0


def loops_priced(asm, kernel_substr, full):
    funcs = __import__("re").split(r"\n(?=_Z[\w]+:)", asm)
    name, loops = im.loops_of(asm, kernel_substr)
    # re-walk with prices: loops_of keeps only histograms, so price the opcode histogram plus a second pass for operand-dependent classes
    return name, loops


def walk(asm, kernel_substr, kind, full, show_ops):
    import re
    funcs = re.split(r"\n(?=_Z[\w]+:)", asm)
    for f in funcs:
        nm = f.split(":", 1)[0]
        if kernel_substr in nm and nm.startswith("_Z"):
            break
    else:
        raise SystemExit("no function " + kernel_substr)
    parent = {}
    for m in re.finditer(r"^\.L(BB\d+_\d+):\s*;\s*Parent Loop (BB\d+_\d+) Depth=1", f, re.M):
        parent[m.group(1)] = m.group(2)
    lines = f.split("\n")
    # loops the asm printer does not annotate (irreducible regions, e.g. the forward producers' group loop): a backward branch to a
    # label outside every annotated loop makes [label, branch] a depth-1 loop of its own
    label_at, annotated = {}, set()
    for i, line in enumerate(lines):
        m = re.match(r"^\.L(BB\d+_\d+):\s*(;.*)?$", line)
        if m:
            label_at[m.group(1)] = i
            if m.group(2) and ("Loop" in m.group(2)):
                annotated.add(m.group(1))
    extra = {}
    end = next((i for i, line in enumerate(lines) if line.strip().startswith("s_endpgm")), len(lines))   # (behind it: out-of-line branch stubs)
    for i, line in enumerate(lines[:end]):
        m = re.match(r"^\ts_c?branch\S*\s+\.L(BB\d+_\d+)", line)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i and m.group(1) not in annotated:
            t = m.group(1)
            if not any(lo <= label_at[t] and i <= hi for lo, hi in extra.values()):
                extra[t] = (label_at[t], max(i, extra.get(t, (0, 0))[1]))
    span_of = {}
    for t, (lo, hi) in extra.items():
        for j in range(lo, hi + 1):
            span_of[j] = t
    loops = collections.OrderedDict()
    cur, depth2 = None, False
    for i, line in enumerate(lines):
        m = re.match(r"^\.L(BB\d+_\d+):\s*(;.*)?$", line)
        if m:
            label, c = m.group(1), m.group(2) or ""
            hdr = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
            if "Loop Header: Depth=1" in c:
                cur, depth2 = label, False
            elif label in parent:
                cur, depth2 = parent[label], True
            elif hdr:
                h, dep = hdr.group(1), int(hdr.group(2))
                cur, depth2 = (parent.get(h, h), dep >= 2)
            elif i in span_of:
                cur, depth2 = span_of[i], False
            else:
                cur = None
            continue
        if line.startswith(";") or not line.startswith("\t") or cur is None:
            continue
        body = line.strip().split(";")[0].strip()
        if not body:
            continue
        op = body.split()[0]
        if op.startswith("."):
            continue
        d = loops.setdefault(cur, {"ops": collections.Counter(), "n": collections.Counter(), "cyc": collections.Counter(), "n2": collections.Counter(),
                                   "cyc2": collections.Counter(), "opc": collections.Counter()})
        d["ops"][op.split("_e32")[0].split("_e64")[0]] += 1
        cls, cyc = price(op.split("_e32")[0].split("_e64")[0], body, full)
        if cls:
            d["n2" if depth2 else "n"][cls] += 1
            d["cyc2" if depth2 else "cyc"][cls] += cyc
            d["opc"][(cls, op.split("_e32")[0].split("_e64")[0])] += 1
    rows = {}
    for hdr, d in loops.items():
        proxy = {"ops": d["ops"], "cls": collections.Counter({"mfma": d["n"]["mfma"]}), "cls2": collections.Counter({"mfma": d["n2"]["mfma"]})}
        role = im.role_of(proxy, kind)
        if role is None or sum(d["n"].values()) + sum(d["n2"].values()) < 60 or role in rows:
            continue
        mult = 4 if (kind == "fwd" and role in ("consumer", "both")) else 1
        n = collections.Counter(d["n"])
        cyc = collections.Counter(d["cyc"])
        for k in d["n2"]:
            n[k] += mult * d["n2"][k]
            cyc[k] += mult * d["cyc2"][k]
        rows[role] = (n, cyc, d["opc"])
    return nm, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", action="store_true", help="opcode histogram per role")
    ap.add_argument("--csrc", default=im.CSRC, help="source directory (default: the tree's csrc)")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for src in ("wkv6_chunk.hip", "wkv6_chunk_bwd12k.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["hipcc"] + im.FLAGS + ["-o", out, os.path.join(args.csrc, src)], stderr=subprocess.DEVNULL)
            res[src] = open(out).read()
    for label, src, sub, kind, tokens, waves, full in (("forward", "wkv6_chunk.hip", "chunk_fwd_kernelILb1ELb0ELb0ELb0ELb1EEE", "fwd", 64, 2, 2.2),
                                                       ("backward", "wkv6_chunk_bwd12k.hip", "chunk_bwd12k_kernelILb1ELi0ELb0ELb1EEE", "bwd", 32, 3, 2.7)):
        nm, rows = walk(res[src], sub, kind, full, args.ops)
        print(f"\n== {label}: {nm[:90]}\n   per loop iteration = {tokens} tokens of one (batch, head); one SIMD hosts one wave of every role ({waves} waves per SIMD); "
              f"full-rate VALU priced at {full} cycles")
        print(f"   {'role':9s} | {'full':>5s} {'half':>5s} {'quart':>5s} = {'VALU':>5s} | {'MFMA':>4s} {'LDS':>4s} | cycles: {'VALU':>6s} {'MFMA':>6s} {'LDS(SIMD share)':>15s} | VALU + MFMA")
        tot = collections.Counter()
        totc = collections.Counter()
        for role, (n, cyc, opc) in rows.items():
            valu = n["full"] + n["half"] + n["quarter"]
            vc = cyc["full"] + cyc["half"] + cyc["quarter"]
            print(f"   {role:9s} | {n['full']:5d} {n['half']:5d} {n['quarter']:5d} = {valu:5d} | {n['mfma']:4d} {n['lds']:4d} | {'':7s} {vc:6.0f} {cyc['mfma']:6.0f} {cyc['lds']:15.0f} | {vc + cyc['mfma']:8.0f}")
            tot.update(n)
            totc.update(cyc)
            if args.ops:
                for (cls, op), k in sorted(opc.items(), key=lambda kv: (kv[0][0], -kv[1])):
                    print(f"        {cls:8s} {op:28s} {k}")
        valu = tot["full"] + tot["half"] + tot["quarter"]
        vc = totc["full"] + totc["half"] + totc["quarter"]
        print(f"   {'per SIMD':9s} | {tot['full']:5d} {tot['half']:5d} {tot['quarter']:5d} = {valu:5d} | {tot['mfma']:4d} {tot['lds']:4d} | {'':7s} {vc:6.0f} {totc['mfma']:6.0f} {totc['lds']:15.0f} | {vc + totc['mfma']:8.0f}")
        print(f"   mean VALU cost {vc / valu:.2f} cycles per instruction; floor (VALU + MFMA, perfectly packed) = {vc + totc['mfma']:.0f} cycles per {tokens} tokens; "
              f"LDS pipe (shared by the CU's 4 SIMDs, this is one SIMD's share) {totc['lds']:.0f}")


if __name__ == "__main__":
    sys.exit(main())
