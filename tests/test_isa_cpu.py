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
