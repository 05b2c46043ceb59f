#!/usr/bin/env python3
"""DIAGNOSTIC since round 3 (the build's gate is tools/check_mir_spills.py, which sees the blocks as the register allocator saw them
and has no false positives; this scan of the final assembly cannot tell a reload folded into a phi copy at the end of an else branch
from one placed too early in the join block, and refused correct kernels). Kept because it points at the place in the assembly.

The first guard against a register-allocator bug seen with this ROCm's LLVM on gfx950 at full register pressure:
VGPR spill code (a scratch_store "Folded Spill" / v_accvgpr_write, or a scratch_load "Folded Reload") gets placed in a control-flow
JOIN block BEFORE the `s_or_b64 exec, exec, s[..]` that re-enables the lanes of the other branch, so only the lanes of one branch
save (or get back) their value and the others later work with garbage — results then depend on what the scratch memory held,
i.e. they change from run to run.
  * round 1: k_step<NED>, a spill at the head of the join block: lanes with negative longitude lost their lambda state
    (the earlier "garbage status word" failure had the same signature);
  * round 2: k_step_air<WA, Xv2, GROUND>, a RELOAD two instructions into the join block: the scripted crosswind landing gave
    different touchdowns from run to run (tools/det_check.py).

The scan: from every block label up to the first branch / next label, collect the spill-code instructions that precede an
`s_or_b64 exec, exec, ...`. A reload whose register is consumed again before that exec restore by a store (the value is only
needed by the lanes that are active there) is harmless and not reported.

Usage: check_isa_spills.py file.s  -> exit 1 and a report if any kernel contains the pattern."""
import re
import sys

SPILL = re.compile(r"^\s*(v_accvgpr_write_b32\s+a\d+,\s*v\d+|scratch_store_dword\w*\s.*Spill)")
RELOAD = re.compile(r"^\s*scratch_load_dword\w*\s+(v\[?\d+)(?::\d+\])?,.*Reload")
# every way this compiler re-enables lanes at a join / else point: `s_or_b64 exec, exec, s[..]` (end of an if), `s_or_saveexec_b64 s[..], s[..]`
# and `s_xor_b64 exec, exec, s[..]` (the else point: the other branch's lanes come on), a plain `s_mov_b64 exec, s[..]` (loop exits)
EXEC_RESTORE = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64\s|s_xor_b64\s+exec,\s*exec,|s_mov_b64\s+exec,\s*s\[)")
LABEL = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
BRANCH = re.compile(r"^\s*s_c?branch")
IGNORE = re.compile(r"^\s*(;|$|\.)")


def scan(path):
    bad = []
    kernel = None
    lines = open(path).read().split("\n")
    for i, line in enumerate(lines):
        m = LABEL.match(line)
        if not m:
            continue
        if m.group(1).startswith("_Z"):
            kernel = m.group(1)
            continue
        head = True            # still at the head of the block (only spill code / nops / waits so far)
        found = []
        for j in range(i + 1, min(i + 80, len(lines))):
            ln = lines[j]
            if IGNORE.match(ln):
                continue
            if LABEL.match(ln) or BRANCH.match(ln):
                break
            if EXEC_RESTORE.match(ln):
                if found:
                    bad.append((kernel, m.group(1), found, j + 1))
                break
            r = RELOAD.match(ln)
            if r:
                # harmless only if the reloaded register is consumed (as the address or data of a memory instruction) before the exec
                # restore: then only the lanes active here ever needed it (epilogue address reloads). Anything else is reported.
                reg = r.group(1).replace("[", "")
                consumed = False
                for k in range(j + 1, min(j + 80, len(lines))):
                    t = lines[k]
                    if EXEC_RESTORE.match(t) or LABEL.match(t) or BRANCH.match(t):
                        break
                    code = t.split(";")[0]
                    if re.match(r"^\s*(global_|flat_|scratch_store|ds_write)", code) and re.search(r"\b" + reg + r"\b|\[" + reg[1:] + r":", code):
                        consumed = True
                        break
                    if re.match(r"^\s*v_lshl_add_u64\s+" + reg.replace("v", r"v\[") , code):   # address arithmetic in place, then the store
                        continue
                if not consumed:
                    found.append((j + 1, ln.strip()))
                head = False
                continue
            if SPILL.match(ln):
                if head or "Spill" in ln:
                    found.append((j + 1, ln.strip()))
                continue
            if not re.match(r"^\s*(s_nop|s_waitcnt)", ln):
                head = False
    return bad


if __name__ == "__main__":
    bad = scan(sys.argv[1])
    for kernel, label, spills, line in bad:
        print(f"{kernel}: block {label}: {len(spills)} spill-code instruction(s) before the exec restore at line {line}: {spills[0][1]} (line {spills[0][0]})")
    print(f"{len(bad)} suspicious block(s)")
    sys.exit(1 if bad else 0)
