#!/usr/bin/env python3
"""Static instruction mix of the chunked kernels' role loops, from the gfx950 ISA hipcc emits (round 4).

    python tools/issue_model.py            # compiles wkv6_chunk.hip and wkv6_chunk_bwd12k.hip to ISA and prints the tables

For the headline instantiations (raw bf16 decay, plain stores) every depth-1 loop of the kernel is one role's main loop (the role
branches are wave-uniform, each role runs its own copy of the `for (stage)` loop with the workgroup barrier inside).  A loop is
labelled by what it contains: global loads of the inputs + v_exp => producers; buffer stores of four gradients => row waves; etc.
Instructions are counted by class (full-rate VALU; "slow" VALU = transcendental, DPP / cross-lane, packed, shifts, integer multiply;
bf16 conversions; MFMA; LDS; vector memory; scalar).  Cycles are NOT guessed per class: profiles/r04_final_pmc.json gives the vector
ALU's busy time per executed instruction, SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU = 4.2-4.3 cycles for both kernels' mixes, and an
MFMA (either shape) occupies the matrix pipe for 16 cycles (tools/microbench); the model in profiles/r04_issue_model.md uses those.
The forward consumers' block loop is a run-time loop of 4 iterations inside the group loop: its body is counted 4 times.
Rare paths (blocks that only execute for clamped decays, tails) are inside the loops and are counted: the totals are upper bounds
by a few per cent, which the comparison with SQ_INSTS_VALU (profiles/r04_final_pmc.json) quantifies.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rwkv_lm_ext_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-S", "--cuda-device-only"]

SLOW_VALU = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_permlane", "v_pk_", "v_cvt_f32_bf16", "v_readlane", "v_readfirstlane",
             "v_perm_b32", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_and_or", "v_lshl_or", "v_lshl_add", "v_mul_lo", "v_mul_hi", "v_mad_u")


def classify(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "dpp" in line or "row_" in line or "quad_perm" in line:
            return "valu_slow"
        if op.startswith("v_cvt_pk_bf16"):
            return "valu_cvt"
        if op.startswith("v_dot2"):
            return "valu"
        if op.startswith(SLOW_VALU):
            return "valu_slow"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


VALU_CYC = 4.25       # vector-ALU busy cycles per VALU instruction (SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU, both kernels)
MFMA_CYC = 16


def loops_of(asm, kernel_substr):
    """{loop header label: Counter of classes, plus 'ops' Counter} for the first function whose name contains kernel_substr."""
    funcs = re.split(r"\n(?=_Z[\w]+:)", asm)
    for f in funcs:
        name = f.split(":", 1)[0]
        if kernel_substr in name and name.startswith("_Z"):
            break
    else:
        raise SystemExit(f"no function matching {kernel_substr}")
    # pass 1: which depth-2 loop headers sit in which depth-1 loop (asm printer comments)
    parent = {}
    for m in re.finditer(r"^\.L(BB\d+_\d+):\s*;\s*Parent Loop (BB\d+_\d+) Depth=1", f, re.M):
        parent[m.group(1)] = m.group(2)
    loops = collections.OrderedDict()
    cur, depth2 = None, False
    for line in f.split("\n"):
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
            else:
                cur = None
            continue
        if line.startswith(";") or not line.startswith("\t") or cur is None:
            continue
        op = line.strip().split()[0]
        if op.startswith(".") or op.startswith(";"):
            continue
        d = loops.setdefault(cur, {"cls": collections.Counter(), "ops": collections.Counter(), "cls2": collections.Counter()})
        d["cls2" if depth2 else "cls"][classify(op, line)] += 1
        d["ops"][op.split("_e32")[0].split("_e64")[0]] += 1
    return name, loops


def role_of(d, kind):
    ops = d["ops"]
    loads = sum(v for k, v in ops.items() if k.startswith("buffer_load"))
    stores = sum(v for k, v in ops.items() if k.startswith("buffer_store"))
    exps = ops.get("v_exp_f32", 0)
    if kind == "bwd":
        if exps >= 16 and loads >= 4:
            return "producer"
        if stores >= 3:
            return "row"
        if stores >= 1:
            return "column"
    else:
        mf = d["cls"]["mfma"] + d["cls2"]["mfma"]
        if exps >= 16 and mf >= 16:
            return "both"             # hipcc merged the two roles' group loops into one (role branch inside): counted together
        if exps >= 16:
            return "producer"
        if mf >= 8:
            return "consumer"
    return None


def table(asm, kernel_substr, kind, unit_tokens, blocks_per_iter=1):
    name, loops = loops_of(asm, kernel_substr)
    out = []
    for hdr, d in loops.items():
        role = role_of(d, kind)
        # inner (depth-2) loops: the forward consumers' block loop runs 4 times per group; the backward's are tag polls (once)
        mult = 4 if (kind == "fwd" and role in ("consumer", "both")) else 1
        c = collections.Counter(d["cls"])
        for k, v in d["cls2"].items():
            c[k] += mult * v
        if role is None or c["valu"] + c["valu_slow"] < 40:
            continue
        valu = c["valu"] + c["valu_slow"] + c["valu_cvt"]
        out.append((role, valu, c["valu"], c["valu_slow"], c["valu_cvt"], c["mfma"], c["lds"], c["vmem"], c["salu"],
                    int(valu * VALU_CYC), MFMA_CYC * c["mfma"]))
    print(f"\n{name[:100]}\n  per loop iteration = {unit_tokens} tokens of one (batch, head); one wave of the role")
    print(f"  {'role':9s} {'VALU':>5s} = {'full':>5s} + {'slow':>5s} + {'cvt':>4s} | {'MFMA':>5s} {'LDS':>5s} {'VMEM':>5s} {'SALU':>5s} | {'VALU-pipe cyc':>13s} {'MFMA-pipe cyc':>13s}")
    for r in out:
        print(f"  {r[0]:9s} {r[1]:5d} = {r[2]:5d} + {r[3]:5d} + {r[4]:4d} | {r[5]:5d} {r[6]:5d} {r[7]:5d} {r[8]:5d} | {r[9]:13d} {r[10]:13d}")
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for src in ("wkv6_chunk.hip", "wkv6_chunk_bwd12k.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["hipcc"] + FLAGS + ["-o", out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
            res[src] = open(out).read()
    fwd = table(res["wkv6_chunk.hip"], "chunk_fwd_kernelILb1ELb0ELb0ELb0ELb1EEE", "fwd", 64)
    bwd = table(res["wkv6_chunk_bwd12k.hip"], "chunk_bwd12k_kernelILb1ELi0ELb0ELb1EEE", "bwd", 32)
    # a SIMD hosts one wave of every role: what it must issue per loop iteration
    for label, rows, tokens, waves in (("forward", fwd, 64, 2), ("backward", bwd, 32, 3)):
        by = {}
        for r in rows:
            by.setdefault(r[0], r)          # first loop of each role (pair / split variants do not exist in these instantiations)
        valu = sum(r[1] for r in by.values())
        vcyc = sum(r[9] for r in by.values())
        mcyc = sum(r[10] for r in by.values())
        lds = sum(r[6] for r in by.values())
        tc = tokens * 64                    # token-channels per (batch, head) and iteration
        print(f"\n{label}: one SIMD per {tokens}-token iteration: {valu} VALU wave-instructions ({4 * valu / tc:.2f} per token-channel over the "
              f"4 SIMDs), {lds} LDS instructions, vector-ALU busy {vcyc} cycles, matrix pipe busy {mcyc} cycles")


if __name__ == "__main__":
    sys.exit(main())
