"""ISA-level guards that need no GPU: hipcc cross-compiles gfx950 here.

Round 4: the backward's consuming roles must not wait with `s_waitcnt vmcnt(0)` inside their stage loops -- that wait
made every stage wait for its own fresh requests and for the acknowledgement of the previous stage's stores (-9 % when removed) --
and the row waves' wait-free releases of the published G operand (tags GB / GD) must stay behind the eight transposed reads of it
in every instantiation (the LDS serves a wave's requests in order; the compiler must not move the tag store above them).
"""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_full_vector_memory_drain_in_the_backwards_stage_loops():
    import check_waitcnt
    assert check_waitcnt.check() == []


def test_the_guard_sees_a_drain_when_there_is_one():
    """The parser on a synthetic listing: a loop with stores and vmcnt(0) is reported, a loads-only loop is not considered."""
    import check_waitcnt
    asm = "\n".join([
        "_Z3fooILb1EEvv:",
        ".LBB0_1:                                ; =>This Loop Header: Depth=1",
        "\tbuffer_load_dwordx2 v[0:1], v2, s[0:3], 0 offen",
        "\ts_waitcnt vmcnt(0)",
        "\tbuffer_store_dwordx4 v[0:3], v4, s[0:3], 0 offen",
        "\ts_cbranch_scc1 .LBB0_1",
        ".LBB0_2:                                ; =>This Loop Header: Depth=1",
        "\tbuffer_load_dwordx2 v[0:1], v2, s[0:3], 0 offen",
        "\ts_waitcnt vmcnt(0)",
        "\ts_cbranch_scc1 .LBB0_2",
        "\ts_endpgm",
    ])
    loops = check_waitcnt.stage_loops(asm, "fooILb1E")
    assert ("BB0_1", 1, 1, [0]) in loops and ("BB0_2", 0, 1, [0]) in loops


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_scratch_access_inside_any_loop_and_no_spill_in_the_benched_kernels():
    """register spills.  From the gfx950 ISA of every instantiation of the chunked kernels: the instantiations the
    headline benchmark runs (plain stores, one workgroup per (batch, head), both decay kinds) spill no vector register at all, and
    no instantiation touches scratch memory inside a stage loop (the known spills -- two registers of the first wkv6_bi half, stored
    in front of the row waves' stage loop and reloaded behind it -- is off every loop: wkv6_chunk_bwd12k.hip has the note)."""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-S", "--cuda-device-only"]
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("wkv6_chunk.hip", "wkv6_chunk_bwd12k.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["hipcc"] + flags + ["-o", out, os.path.join(root, "rwkv_lm_ext_amd", "csrc", src)])
            asm = open(out).read()
            spills = dict(zip(re.findall(r"^\s+\.name:\s+(\S+)", asm, re.M), (int(x) for x in re.findall(r"^\s+\.vgpr_spill_count:\s+(\d+)", asm, re.M))))
            assert spills, src
            for name, n in spills.items():
                if "chunk_fwd_kernelILb1ELb0ELb0ELb0ELb1E" in name or "chunk_fwd_kernelILb0ELb0ELb0ELb0ELb1E" in name or \
                        "chunk_bwd12k_kernelILb1ELi0ELb0ELb1E" in name or "chunk_bwd12k_kernelILb0ELi0ELb0ELb1E" in name:
                    assert n == 0, (name, n)
                # round 6: the persistent wkv6_bi launches, both decay kinds -- no vector register in scratch at all (round
                # 5: 26 in the fused backward, 63 scratch instructions on every call boundary; the fp32-ew kind reloaded inside its stage loops)
                if "_bi_kernel" in name:
                    assert n == 0, (name, n)
            scratch = dict(zip(re.findall(r"^\s+\.name:\s+(\S+)", asm, re.M), (int(x) for x in re.findall(r"^\s+\.private_segment_fixed_size:\s+(\d+)", asm, re.M))))
            assert all(v == 0 for k_, v in scratch.items() if "_bi_kernel" in k_), {k_: v for k_, v in scratch.items() if "_bi_kernel" in k_}
            # scratch instructions must sit outside every stage / group / block loop.  The asm printer marks loop blocks "in Loop: Header=..
            # Depth=n" / "Loop Header: Depth=n"; in the persistent wkv6_bi kernels the depth-1 loop is the walk over (batch, head) rows --
            # a few scalar-register spills per ROW are reloaded there -- and the stage loops are depth 2
            in_loop, persistent = False, False
            for line in asm.split("\n"):
                m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", line)
                if m:
                    c = m.group(1) or ""
                    d = re.search(r"Depth[= ](\d+)", c)
                    in_loop = ("Loop" in c) and int(d.group(1) if d else 1) >= (2 if persistent else 1)
                elif re.match(r"^_Z\w+:", line):
                    in_loop, persistent = False, "_bi_kernel" in line
                elif in_loop and line.strip().startswith("scratch_"):
                    raise AssertionError(f"{src}: scratch access inside a loop: {line.strip()}")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_integer_multiply_in_the_benched_kernels_loops():
    """Round 5: token offsets are a hoisted lane part + a wave-uniform part formed on the scalar unit (wkv6_scan.h:
    TokAddr).  The general reversal map cost a compare, a select, a subtraction and a (half-rate) 32-bit multiply per access; the benched
    instantiations (no per-tensor reversal map) must have none of those multiplies left inside a group / stage loop."""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-S", "--cuda-device-only"]
    wanted = {"wkv6_chunk.hip": ("chunk_fwd_kernelILb1ELb0ELb0ELb0ELb1E", "chunk_fwd_kernelILb0ELb0ELb0ELb0ELb1E"),
              "wkv6_chunk_bwd12k.hip": ("chunk_bwd12k_kernelILb1ELi0ELb0ELb1E", "chunk_bwd12k_kernelILb0ELi0ELb0ELb1E")}
    with tempfile.TemporaryDirectory() as tmp:
        for src, names in wanted.items():
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["hipcc"] + flags + ["-o", out, os.path.join(root, "rwkv_lm_ext_amd", "csrc", src)])
            seen = 0
            for fn in re.split(r"\n(?=_Z[\w]+:)", open(out).read()):
                name = fn.split(":", 1)[0]
                if not any(n in name for n in names):
                    continue
                seen += 1
                in_loop = False
                for line in fn.split("\n"):
                    m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", line)
                    if m:
                        in_loop = "Loop" in (m.group(1) or "")
                    elif in_loop and re.match(r"\s+v_(mul_lo_u32|mul_hi_u32|mad_u64_u32)", line):
                        raise AssertionError(f"{name}: integer multiply inside a loop: {line.strip()}")
            assert seen == len(names), (src, seen)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_chained_wkv6_bi_backward_keeps_full_drains_out_of_its_stage_loops():
    """Round 6: the persistent wkv6_bi backward chains its calls -- the column waves' requests for the call that follows are issued
    unconditionally from a SELECTED row (as branches they made the loaded registers merges, which hipcc resolved with `s_waitcnt vmcnt(0)`
    inside the stage loop: every stage then waited for its own fresh requests and the previous stage's store acknowledgements, the round-4
    stall).  From the ISA of chunk_bwd12k_bi_kernel (both decay kinds): the stage loops (depth 2; the row walk is depth 1) of the row waves
    of both halves and of the column waves of the first half wait with counted vmcnt only.  (The second half's column loop adds the first
    half's partial to its result right behind the load: its vmcnt(0) predates the chaining.)"""
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-S", "--cuda-device-only"]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc"] + flags + ["-o", out, os.path.join(root, "rwkv_lm_ext_amd", "csrc", "wkv6_chunk_bwd12k.hip")])
        asm = open(out).read()
    seen = 0
    for fn in re.split(r"\n(?=_Z[\w]+:)", asm):
        if "chunk_bwd12k_bi_kernel" not in fn.split(":", 1)[0]:
            continue
        seen += 1
        loops, cur = {}, None
        for line in fn.split("\n"):
            m = re.match(r"^\.L(BB\d+_\d+):\s*(;.*)?$", line)
            if m:
                c = m.group(2) or ""
                h = re.search(r"Header=(BB\d+_\d+) Depth=2", c)
                cur = m.group(1) if "Loop Header: Depth=2" in c else (h.group(1) if h else None)
                continue
            if cur is None:
                continue
            d = loops.setdefault(cur, {"stores": 0, "mfma": 0, "waits": []})
            op = line.strip().split(" ")[0] if line.strip() else ""
            if op.startswith("buffer_store"):
                d["stores"] += 1
            elif op.startswith("v_mfma"):
                d["mfma"] += 1
            elif op == "s_waitcnt":
                w = re.search(r"vmcnt\((\d+)\)", line)
                if w:
                    d["waits"].append(int(w.group(1)))
        consuming = [v for v in loops.values() if v["stores"] > 0 and v["mfma"] > 0]
        assert len(consuming) == 4, {k: (v["stores"], v["mfma"]) for k, v in loops.items()}     # row + column waves of both halves
        second_half_column = min(consuming, key=lambda v: v["stores"])                           # one merged gv store per stage
        for v in consuming:
            if v is not second_half_column:
                assert 0 not in v["waits"], v
    assert seen == 2
