#!/usr/bin/env python3
"""Build-time guard against a register-allocator bug seen with this ROCm's LLVM on gfx950 at full register pressure:
a VGPR spill (v_accvgpr_write / scratch_store) gets placed at the head of a control-flow JOIN block BEFORE the
`s_or_b64 exec, exec, s[..]` that re-enables the lanes of the other branch, so only the lanes of one branch save
their value and the others later reload garbage. (Found in k_step<NED>: lanes with negative longitude lost their λ state;
the earlier "garbage status word" failure had the same signature.)

Usage: check_isa_spills.py file.s  -> exit 1 and a report if any kernel contains the pattern."""
import re
import sys

SPILL = re.compile(r"^\s*(v_accvgpr_write_b32\s+a\d+,\s*v\d+|scratch_store_dword\w*\s.*Spill)")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
LABEL = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
IGNORE = re.compile(r"^\s*(;|$|\.|s_nop|s_waitcnt)")


def scan(path):
    bad = []
    kernel = None
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = LABEL.match(lines[i])
        if m:
            if m.group(1).startswith("_Z"):
                kernel = m.group(1)
            # walk the head of the block: spills seen before an exec restore are lane-incomplete
            j = i + 1
            spills = []
            while j < len(lines):
                ln = lines[j]
                if IGNORE.match(ln):
                    j += 1
                    continue
                if SPILL.match(ln):
                    spills.append((j + 1, ln.strip()))
                    j += 1
                    continue
                if EXEC_RESTORE.match(ln) and spills:
                    bad.append((kernel, m.group(1), spills, j + 1))
                break
        i += 1
    return bad


if __name__ == "__main__":
    bad = scan(sys.argv[1])
    for kernel, label, spills, line in bad:
        print(f"{kernel}: block {label}: {len(spills)} spill(s) before the exec restore at line {line}: {spills[0][1]} (line {spills[0][0]})")
    print(f"{len(bad)} suspicious block(s)")
    sys.exit(1 if bad else 0)
