#!/usr/bin/env python3
"""Guard for the round-4 finding: no `s_waitcnt vmcnt(0)` inside the stage loops of the consuming roles.

    python tools/check_waitcnt.py          # compiles wkv6_chunk_bwd12k.hip to gfx950 ISA (~15 s) and prints the waits per loop

A loop-carried register whose first value is a load still in flight at loop entry makes hipcc place the wait at the register's first
use INSIDE the loop, sized for the first entry (vmcnt(0)); in steady state that wait then covers whatever the wave has just requested
and the acknowledgement of its last stores, every stage.  The row and column waves' loops (the ones that store gradients) must wait
only with counted vmcnt(N > 0); the producers' loop legitimately drains its queue (its oldest requests are the ones it needs).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rwkv_lm_ext_amd", "csrc", "wkv6_chunk_bwd12k.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-S", "--cuda-device-only"]
KERNELS = ("chunk_bwd12k_kernelILb1ELi0ELb0ELb1EEE", "chunk_bwd12k_kernelILb0ELi0ELb0ELb1EEE")    # raw-w / fp32-ew, plain stores, one workgroup per pair


def stage_loops(asm, kernel_substr):
    """[(header label, #buffer stores, #buffer loads, [vmcnt values waited for])] of the depth-1 loops of one kernel."""
    for f in re.split(r"\n(?=_Z[\w]+:)", asm):
        name = f.split(":", 1)[0]
        if kernel_substr in name and name.startswith("_Z"):
            break
    else:
        raise SystemExit(f"no function matching {kernel_substr}")
    parent = {m.group(1): m.group(2) for m in re.finditer(r"^\.L(BB\d+_\d+):\s*;\s*Parent Loop (BB\d+_\d+) Depth=1", f, re.M)}
    loops, cur = {}, None
    for line in f.split("\n"):
        m = re.match(r"^\.L(BB\d+_\d+):\s*(;.*)?$", line)
        if m:
            label, c = m.group(1), m.group(2) or ""
            hdr = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
            if "Loop Header: Depth=1" in c:
                cur = label
            elif label in parent:
                cur = parent[label]
            elif hdr:
                cur = parent.get(hdr.group(1), hdr.group(1))
            else:
                cur = None
            continue
        if cur is None or not line.startswith("\t"):
            continue
        d = loops.setdefault(cur, {"stores": 0, "loads": 0, "waits": []})
        op = line.strip().split()[0]
        if op.startswith("buffer_store"):
            d["stores"] += 1
        elif op.startswith("buffer_load"):
            d["loads"] += 1
        elif op == "s_waitcnt":
            w = re.search(r"vmcnt\((\d+)\)", line)
            if w:
                d["waits"].append(int(w.group(1)))
    return [(k, v["stores"], v["loads"], v["waits"]) for k, v in loops.items()]


# The row waves release the published G operand (tags GB / GD, immediate offsets 176 / 336 from the wave's tag base register: wkv6_chunk_bwd12k.hip, publish) right behind their reads of
# it, without waiting for the data: correct because the LDS serves a wave's requests in order AND the compiler keeps the tag store
# behind the eight transposed reads (it may alias them).  This lists, for every tag store, how many of those reads precede it closely.
RELEASE_TAG_OFFSETS = (176, 336)


def release_order(asm, kernel_substr, window=80):
    for f in re.split(r"\n(?=_Z[\w]+:)", asm):
        name = f.split(":", 1)[0]
        if kernel_substr in name and name.startswith("_Z"):
            break
    else:
        raise SystemExit(f"no function matching {kernel_substr}")
    lines = f.split("\n")
    out = []
    for i, line in enumerate(lines):
        m = re.search(r"ds_write_b32 .*offset:(\d+)\b", line)
        if m and int(m.group(1)) in RELEASE_TAG_OFFSETS:
            reads = sum(1 for x in lines[max(0, i - window):i] if x.strip().startswith("ds_read_b64_tr_b16"))
            out.append((int(m.group(1)), reads))
    return out


def check(verbose=False):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc"] + FLAGS + ["-o", out, SRC], stderr=subprocess.DEVNULL)
        asm = open(out).read()
    bad = []
    for kern in KERNELS + ("chunk_bwd12k_kernelILb1ELi1ELb0ELb1EEE", "chunk_bwd12k_kernelILb1ELi2ELb0ELb1EEE", "chunk_bwd12k_pair_kernelILb1EEE"):
        rel = release_order(asm, kern)
        if verbose:
            print(f"{kern}: operand releases (tag offset, transposed reads in the 80 lines before): {rel}")
        if sorted(o for o, _ in rel) != sorted(RELEASE_TAG_OFFSETS) or any(n < 8 for _, n in rel):
            bad.append((kern, "operand release not behind its eight reads", rel))
    for kern in KERNELS:
        consuming = 0
        for label, stores, loads, waits in stage_loops(asm, kern):
            if stores == 0:
                continue                                   # producers (loads only), polls, tails
            consuming += 1
            if verbose:
                print(f"{kern}: loop {label}: {stores} buffer stores, {loads} buffer loads, vmcnt waits {waits}")
            if 0 in waits:
                bad.append((kern, label, waits))
        if consuming < 2:
            bad.append((kern, "expected the row and the column waves' stage loops", consuming))
    return bad


if __name__ == "__main__":
    problems = check(verbose=True)
    for p in problems:
        print("PROBLEM:", p)
    sys.exit(1 if problems else 0)
