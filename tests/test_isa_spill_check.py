"""tools/check_isa_spills.py — the build-time guard against spill code placed before the exec restore of a control-flow join block
(docs/design/k_step_air.md, "A compiler bug the build now guards against") — on hand-written assembly fragments: the patterns the two observed
miscompiles had must be reported, the harmless look-alikes must not."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_isa_spills", os.path.join(ROOT, "tools", "check_isa_spills.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

KERNEL = "_ZN3fbd6k_demoEv:\n\ts_load_dwordx2 s[0:1], s[4:5], 0x0\n"


def _scan(tmp_path, body):
    p = tmp_path / "frag.s"
    p.write_text(KERNEL + body)
    return chk.scan(str(p))


def test_spill_at_the_head_of_a_join_block_is_reported(tmp_path):     # round 1: k_step<NED>
    bad = _scan(tmp_path, ".LBB0_7:\n\tv_accvgpr_write_b32 a12, v40\n\ts_or_b64 exec, exec, s[10:11]\n\tv_add_f64 v[2:3], v[2:3], v[4:5]\n")
    assert len(bad) == 1 and bad[0][0].startswith("_ZN3fbd6k_demo") and bad[0][1] == ".LBB0_7"


def test_reload_inside_a_join_block_is_reported(tmp_path):            # round 2: k_step_air<WA, Xv2, GROUND>
    bad = _scan(tmp_path, ".LBB0_9:\n\tv_cmp_ne_u32_e32 vcc, 0, v99\n\tscratch_load_dword v3, off, off offset:268 ; 4-byte Folded Reload\n"
                          "\ts_andn2_b64 s[6:7], s[6:7], exec\n\ts_or_b64 exec, exec, s[52:53]\n\tv_cndmask_b32_e32 v4, 0, v3, vcc\n")
    assert len(bad) == 1 and "Folded Reload" in bad[0][2][0][1]


def test_scratch_spill_anywhere_before_the_exec_restore_is_reported(tmp_path):
    bad = _scan(tmp_path, ".LBB0_3:\n\tv_mul_f64 v[0:1], v[2:3], v[4:5]\n\tscratch_store_dwordx2 off, v[0:1], off offset:16 ; 8-byte Folded Spill\n"
                          "\ts_or_b64 exec, exec, s[2:3]\n")
    assert len(bad) == 1


def test_spill_code_behind_the_exec_restore_is_fine(tmp_path):
    assert _scan(tmp_path, ".LBB0_4:\n\ts_or_b64 exec, exec, s[2:3]\n\tscratch_load_dword v3, off, off offset:8 ; 4-byte Folded Reload\n"
                           "\tv_accvgpr_write_b32 a1, v3\n\ts_branch .LBB0_5\n") == []


def test_reload_that_only_feeds_a_store_of_the_active_lanes_is_fine(tmp_path):   # epilogue address reloads
    assert _scan(tmp_path, ".LBB0_5:\n\tscratch_load_dwordx2 v[8:9], off, off offset:56 ; 8-byte Folded Reload\n\ts_waitcnt vmcnt(0)\n"
                           "\tglobal_store_dword v[8:9], v2, off\n\ts_or_b64 exec, exec, s[2:3]\n") == []


def test_blocks_without_an_exec_restore_are_not_join_blocks(tmp_path):
    assert _scan(tmp_path, ".LBB0_6:\n\tscratch_load_dword v3, off, off offset:8 ; 4-byte Folded Reload\n\tv_add_u32_e32 v3, v3, v4\n\ts_cbranch_execz .LBB0_8\n") == []


def test_agpr_move_in_the_middle_of_a_block_is_not_spill_placement(tmp_path):    # (only head-of-block AGPR writes count)
    assert _scan(tmp_path, ".LBB0_2:\n\tv_fma_f64 v[0:1], v[2:3], v[4:5], v[0:1]\n\tv_accvgpr_write_b32 a3, v0\n\ts_or_b64 exec, exec, s[2:3]\n") == []


def test_the_other_ways_of_re_enabling_lanes_count_as_exec_restores(tmp_path):
    """`s_or_saveexec_b64` / `s_xor_b64 exec, exec` (the else point of an if-else) and `s_mov_b64 exec, s[..]` (loop exits) bring lanes
    back just like `s_or_b64 exec, exec`: spill code in front of them is lane-incomplete in the same way."""
    spill = "\tscratch_store_dword off, v7, off offset:24 ; 4-byte Folded Spill\n"
    for restore in ("s_or_saveexec_b64 s[8:9], s[8:9]", "s_xor_b64 exec, exec, s[8:9]", "s_mov_b64 exec, s[8:9]"):
        bad = _scan(tmp_path, ".LBB0_11:\n" + spill + "\t" + restore + "\n\tv_mov_b32_e32 v1, v2\n")
        assert len(bad) == 1, restore
    # an instruction that only takes lanes away is not a restore
    assert _scan(tmp_path, ".LBB0_12:\n" + spill + "\ts_and_saveexec_b64 s[8:9], vcc\n\ts_cbranch_execz .LBB0_13\n") == []
