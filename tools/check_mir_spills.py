#!/usr/bin/env python3
"""Build-time guard, at the level where the fault is made: the machine IR right after register allocation.

The fault (this ROCm's LLVM, gfx950, kernels at the register limit): the register allocator places code of its own — a spill
(SI_SPILL_*_SAVE), a reload (SI_SPILL_*_RESTORE) or the COPY of a live-range split into an AGPR — at the head of a control-flow JOIN
block, in front of the `$exec = S_OR_B64 $exec, ...` that re-enables the lanes of the other branch. Only the lanes of one branch then
save (or get back) the value; the others later work with whatever the register or the scratch slot held, and results change from run to
run (round 1: k_step<NED>, lanes with negative longitude lost their lambda state; round 2: the ground-capable Cessna172Xv2 instance,
scripted landings touched down differently from run to run).

Why the machine IR and not the assembly (tools/check_isa_spills.py, the first guard): when the allocator runs, SILowerControlFlow has
already put every exec restore at the HEAD of a machine basic block of its own (it splits the block where needed), so in the IR dumped
after `virtregrewriter` the rule is exact —
    in a machine basic block, nothing that touches a VGPR or an AGPR may stand in front of the first instruction that re-enables lanes
    ($exec = S_OR_B64 $exec, x | S_OR_SAVEEXEC_B64 x | $exec = S_XOR_B64 $exec, x | $exec = S_MOV_B64 x; the `_term` forms at block
    ends narrow the mask and do not count; the S_OR_SAVEEXEC_B64 -1 ... S_MOV_B64 bracket of a whole-wave SGPR spill does not count;
    v_readlane / v_writelane of SGPR spills ignore exec and do not count)
— while in the final assembly the blocks have been merged again, and a reload that the allocator folded into a phi copy at the END of
the else branch (legitimate: it defines the register for that branch's lanes only, the other branch defined it for its own) is
indistinguishable from a reload placed too early in the join block. The assembly scan therefore refused correct kernels (every
measured refusal of round 3 — the Xv2 record written back only where changed, the ECEF instance of the ground-capable Xv2 pass — shows
no fault here), and it is kept as a diagnostic only. Checked on history: the tree of round 2 that misbehaved (fd0bf76^) fails this
check in exactly the kernel that misbehaved (three join blocks of k_step_air<WA, Xv2, GROUND>: AGPR split copies ahead of S_OR_B64),
every tree shipped since passes.

Usage: check_mir_spills.py dump.mir   (the stderr of `hipcc ... -mllvm -print-after=virtregrewriter`; the LAST dump of each function
is the one that counts: the allocator runs once per register class)  -> exit 1 and a report if any function contains the pattern."""
import re
import sys

DUMP = "# *** IR Dump After Virtual Register Rewriter (virtregrewriter) ***:"
FUNC = re.compile(r"# Machine code for function (\S+):")
BLOCK = re.compile(r"\n(?=\d+B\tbb\.\d+)")
HEAD = re.compile(r"\d+B\tbb\.(\d+)")
INSTR = re.compile(r"\d+B\t  \S")
WIDEN = re.compile(r"\$exec = (S_OR_B64 \$exec|S_XOR_B64 \$exec|S_MOV_B64 |COPY )|S_OR_SAVEEXEC_B64 (?!-1)")
VECTOR = re.compile(r"\$[av]gpr\d")
LANE_OPS = ("SI_RESTORE_S32_FROM_VGPR", "SI_SPILL_S32_TO_VGPR",     # v_readlane / v_writelane: exec does not matter
            "= IMPLICIT_DEF", "= KILL ", "DBG_VALUE")                  # no code


def functions(text):
    """{name: last dump of that function}"""
    out = {}
    for d in text.split(DUMP)[1:]:
        m = FUNC.search(d)
        if m:
            out[m.group(1)] = d
    return out


def scan_function(dump):
    """[(bb number, widening instruction, [offending instructions])], and the number of widening points examined"""
    bad, seen = [], 0
    for b in BLOCK.split(dump):
        h = HEAD.match(b)
        if not h:
            continue
        ins = [ln.split("\t", 1)[1].strip() for ln in b.split("\n")[1:] if INSTR.match(ln)]
        pre, wwm = [], False
        for body in ins:
            if "S_OR_SAVEEXEC_B64 -1" in body:
                wwm = True
            if WIDEN.search(body):
                if wwm and "S_MOV_B64" in body:          # the end of a whole-wave bracket, not a join
                    wwm = False
                    pre.append(body)
                    continue
                seen += 1
                vec = [p for p in pre if VECTOR.search(p) and not any(op in p for op in LANE_OPS)]
                if vec:
                    bad.append((int(h.group(1)), body, vec))
                break
            pre.append(body)
    return bad, seen


def scan(path):
    fs = functions(open(path, errors="replace").read())
    reports, points = [], 0
    for name, dump in fs.items():
        bad, seen = scan_function(dump)
        points += seen
        reports += [(name, *b) for b in bad]
    return reports, points, len(fs)


if __name__ == "__main__":
    reports, points, nf = scan(sys.argv[1])
    for name, bb, widen, vec in reports:
        print(f"{name}: bb.{bb}: {len(vec)} vector-register instruction(s) ahead of `{widen[:70]}`: {vec[0][:110]}")
    print(f"{len(reports)} faulty block(s); {points} exec-restore points in {nf} functions examined")
    if nf == 0 or points == 0:
        print("no machine IR found in the input: was the compiler run with -mllvm -print-after=virtregrewriter ?")
        sys.exit(2)
    sys.exit(1 if reports else 0)
