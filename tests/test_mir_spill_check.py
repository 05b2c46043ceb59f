"""tools/check_mir_spills.py — the gate of the build (machine IR after register allocation: nothing that touches a vector register may
stand in front of the instruction that re-enables lanes at the head of a join block) — on hand-written dumps in the format of
`-mllvm -print-after=virtregrewriter`: the fault as it looked in round 2's tree must be reported, what the allocator legitimately puts
there must not, and the last dump of a function is the one that counts."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_mir_spills", os.path.join(ROOT, "tools", "check_mir_spills.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

BANNER = "# *** IR Dump After Virtual Register Rewriter (virtregrewriter) ***:\n"
OR_EXEC = "$exec = S_OR_B64 $exec, killed renamable $sgpr6_sgpr7, implicit-def $scc"
FMA = "renamable $vgpr2_vgpr3 = nofpexcept V_FMA_F64_e64 0, $vgpr4_vgpr5, 0, $vgpr6_vgpr7, 0, $vgpr2_vgpr3, 0, 0, implicit $mode, implicit $exec"


def dump(name, *blocks, props="NoPHIs, TracksLiveness, NoVRegs, TiedOpsRewritten"):
    """blocks: lists of instruction strings; slot indexes are made up the way the real dump numbers them"""
    out, slot = [BANNER, f"# Machine code for function {name}: {props}\n\n"], 0
    for k, ins in enumerate(blocks):
        out.append(f"{slot}B\tbb.{k}:\n\t  successors: %bb.{k + 1}(0x80000000); %bb.{k + 1}(100.00%)\n\n")
        slot += 16
        for i in ins:
            out.append(f"{slot}B\t  {i}\n")
            slot += 16
        out.append("\n")
    out.append(f"# End machine code for function {name}.\n\n")
    return "".join(out)


def scan(tmp_path, text):
    p = tmp_path / "dump.mir"
    p.write_text(text)
    return chk.scan(str(p))


def test_split_copy_into_an_agpr_ahead_of_the_exec_restore_is_reported(tmp_path):     # round 2: k_step_air<WA, Xv2, GROUND>, bb.56
    reports, points, nf = scan(tmp_path, dump("_ZN3fbd6k_demoEv", [FMA, "S_CBRANCH_EXECZ %bb.1, implicit $exec"],
                                              ["renamable $agpr100_agpr101 = COPY killed renamable $vgpr54_vgpr55", "renamable $sgpr30 = S_MOV_B32 1413754136", OR_EXEC, FMA]))
    assert nf == 1 and points == 1 and len(reports) == 1
    name, bb, widen, vec = reports[0]
    assert name == "_ZN3fbd6k_demoEv" and bb == 1 and "S_OR_B64" in widen and "agpr100" in vec[0]


def test_spill_and_reload_pseudos_ahead_of_the_exec_restore_are_reported(tmp_path):
    for pseudo in ("SI_SPILL_AV64_SAVE killed $vgpr2_vgpr3, %stack.53, $sgpr32, 0, implicit $exec :: (store (s64) into %stack.53, align 4, addrspace 5)",
                   "renamable $vgpr8_vgpr9 = SI_SPILL_AV64_RESTORE %stack.27, $sgpr32, 0, implicit $exec :: (load (s64) from %stack.27, align 4, addrspace 5)"):
        for restore in (OR_EXEC, "renamable $sgpr0_sgpr1 = S_OR_SAVEEXEC_B64 killed renamable $sgpr0_sgpr1, implicit-def $exec, implicit-def dead $scc, implicit $exec",
                        "$exec = S_XOR_B64 $exec, renamable $sgpr8_sgpr9, implicit-def $scc", "$exec = S_MOV_B64 killed renamable $sgpr8_sgpr9"):
            reports, _, _ = scan(tmp_path, dump("f", [pseudo, restore, FMA]))
            assert len(reports) == 1, (pseudo, restore)


def test_what_legitimately_stands_there_is_not_reported(tmp_path):
    ok = dump("f",
              # scalar code and the lane moves of an SGPR spill (v_readlane / v_writelane ignore exec) ahead of the restore
              ["$sgpr6 = SI_RESTORE_S32_FROM_VGPR $vgpr255, 23, implicit-def $sgpr6_sgpr7", "$sgpr7 = SI_RESTORE_S32_FROM_VGPR $vgpr255, 24",
               "$vgpr254 = SI_SPILL_S32_TO_VGPR $sgpr56, 3, killed $vgpr254(tied-def 0)", "$vgpr255 = IMPLICIT_DEF", OR_EXEC,
               "renamable $vgpr8_vgpr9 = SI_SPILL_AV64_RESTORE %stack.27, $sgpr32, 0, implicit $exec :: (load (s64) from %stack.27, align 4, addrspace 5)", FMA],
              # the else branch: real work, a reload folded into the phi copy at its END, and the restore at the head of the NEXT block
              [FMA, "renamable $vgpr3 = SI_SPILL_AV32_RESTORE %stack.153, $sgpr32, 0, implicit $exec :: (load (s32) from %stack.153, addrspace 5)"],
              [OR_EXEC, FMA],
              # narrowing the mask at a block's end is not a restore
              [FMA, "renamable $sgpr4_sgpr5 = S_AND_B64 renamable $sgpr2_sgpr3, $exec, implicit-def $scc", "$exec = S_MOV_B64_term killed renamable $sgpr4_sgpr5",
               "S_CBRANCH_EXECZ %bb.5, implicit $exec"],
              # the whole-wave bracket around an SGPR spill through memory
              [FMA, "$sgpr100_sgpr101 = S_OR_SAVEEXEC_B64 -1, implicit-def $exec, implicit-def dead $scc, implicit $exec",
               "SCRATCH_STORE_DWORD_SADDR killed $vgpr254, $sgpr32, 0, 0, implicit $exec, implicit $flat_scr", "$exec = S_MOV_B64 killed $sgpr100_sgpr101", FMA])
    reports, points, nf = scan(tmp_path, ok)
    assert reports == [] and points == 2 and nf == 1


def test_the_last_dump_of_a_function_counts(tmp_path):
    """The allocator runs once per register class (SGPRs, whole-wave registers, VGPRs) and every run is followed by a rewriter dump: in
    the earlier ones the vector registers are still virtual, and a fault of the last run shows only in the last."""
    early = dump("f", [OR_EXEC, "%5:vreg_64_align2 = COPY %9:vreg_64_align2", FMA], props="NoPHIs, TracksLiveness, TiedOpsRewritten")
    late_bad = dump("f", ["renamable $agpr4_agpr5 = COPY killed renamable $vgpr174_vgpr175", OR_EXEC, FMA])
    late_ok = dump("g", [OR_EXEC, FMA])
    reports, points, nf = scan(tmp_path, early + dump("g", [FMA]) + late_bad + late_ok)
    assert nf == 2 and [r[0] for r in reports] == ["f"]


def test_an_input_without_machine_ir_is_an_error_not_a_pass(tmp_path):
    import subprocess
    import sys
    p = tmp_path / "empty.mir"
    p.write_text("warning: something\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mir_spills.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 2 and "no machine IR" in r.stdout
