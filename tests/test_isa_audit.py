"""Build-time safety net for the hand-counted register ring of gemv_chain_kernel: hipcc must not touch a VGPR that
still has an inline-asm global load in flight (tools/isa_audit.py explains the analysis)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402

GOOD = """
	v_mov_b32_e32 v9, 0
.LBB0_1:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v1, s[2:3] nt ; RING_LOAD
	;;#ASMEND
	v_add_u32_e32 v1, 64, v1
	s_cbranch_scc1 .LBB0_3
	;;#ASMSTART
	s_waitcnt vmcnt(0) ; RING_RETIRE v[4:7]
	;;#ASMEND
	v_lshlrev_b32_e32 v8, 16, v4
	s_branch .LBB0_1
.LBB0_3:
	;;#ASMSTART
	s_waitcnt vmcnt(0) ; RING_RETIRE_ALL
	;;#ASMEND
	s_endpgm
"""
BAD_COPY = GOOD.replace("\tv_add_u32_e32 v1, 64, v1\n", "\tv_add_u32_e32 v1, 64, v1\n\tv_mov_b32_e32 v20, v5\n")
BAD_BACKEDGE = GOOD.replace("\tv_lshlrev_b32_e32 v8, 16, v4\n", "\tv_lshlrev_b32_e32 v8, 16, v4\n").replace(
    "\ts_waitcnt vmcnt(0) ; RING_RETIRE v[4:7]\n", "\ts_waitcnt vmcnt(0) ; RING_RETIRE v[4:5]\n")


def _lines(txt):
    return list(enumerate(txt.split("\n"), 1))


def test_auditor_accepts_a_correct_ring():
    assert isa_audit.audit_function(_lines(GOOD)) == []


def test_auditor_catches_a_copy_of_an_in_flight_register():
    v = isa_audit.audit_function(_lines(BAD_COPY))
    assert v and any("v_mov_b32_e32 v20, v5" in t for _, t, _ in v)


def test_auditor_tracks_the_loop_back_edge():
    # v[6:7] are never retired: the next iteration's address arithmetic is fine, but reading v6 would not be
    bad = BAD_BACKEDGE.replace("\tv_lshlrev_b32_e32 v8, 16, v4\n", "\tv_lshlrev_b32_e32 v8, 16, v6\n")
    assert isa_audit.audit_function(_lines(bad))


def test_compiled_kernels_keep_their_hands_off_the_ring():
    assert isa_audit.main() == 0
