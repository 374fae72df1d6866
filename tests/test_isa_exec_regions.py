"""Build-time property of the shipped device code (no GPU needed): between the two exec writes with which march_p2_kernel
switches a packet's idle lanes off for its eight corner loads there is nothing but those loads -- scheduler fences do not stop
the register allocator from placing a copy or a spill there, which would run with the idle lanes off (tools/check_exec_regions.py,
csrc/vr_p2.h: p2_request)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_exec_regions as cer  # noqa: E402

LIB = os.path.join(ROOT, "volumerendering_amd", "libvr_hip.so")


@pytest.mark.skipif(not os.path.exists(cer.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_nothing_but_loads_between_the_exec_writes():
    assert os.path.exists(LIB), "libvr_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    objs = cer.code_objects(LIB)
    assert len(objs) >= 2  # separately rounded and fused multiply-adds: two translation units
    regions, bad = 0, []
    for o in objs:
        r, b = cer.check(cer.disassemble(o))
        regions += r
        bad += b
    assert regions >= 32, regions      # every skipping instantiation has its four requests (two at the start, two per trip)
    assert not bad, bad[:10]


def test_the_checker_sees_a_foreign_instruction():
    text = """
0000000000001000 <_ZN2vr15march_p2_kernelILi1ELb1ELb0ELb0EEEvNS_10MarchBatchENS_7PwQueueE>:
	s_mov_b64 s[16:17], exec
	s_and_b64 exec, exec, s[10:11]
	buffer_load_dwordx4 v[8:11], v13, s[84:87], 0 idxen
	s_nop 0
	v_mov_b32_e32 v1, v2
	buffer_load_dwordx4 v[36:39], v14, s[84:87], 0 idxen
	s_mov_b64 exec, s[16:17]
	v_add_f32_e32 v1, v2, v3
"""
    regions, bad = cer.check(text)
    assert regions == 1 and len(bad) == 1 and "v_mov_b32" in bad[0][1]
