#!/usr/bin/env python3
"""Static instruction mix per RHS phase of the airborne stepper, from a device assembly built with -DFB_PHASE_MARK
(hipcc ... -DFB_PHASE_MARK --cuda-device-only -S): counts the instructions between the "; FBPHASE k" markers inside the main loop."""
import collections
import re
import sys
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2] if len(sys.argv) > 2 else "_ZN3fbd10k_step_airILi0ELb0ELb0EEEvNS_5KArgsEi"
start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
body = lines[start:end]
lo = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "Inner" not in l][-1]
phase = "pre"
mix = collections.OrderedDict()
for l in body[lo:]:
    m = re.search(r"; FBPHASE (\d+)", l)
    if m:
        phase = m.group(1)
        continue
    if not l.startswith("\t"):
        continue
    s = l.strip()
    if s.startswith(".") or s.startswith(";"):
        continue
    op = s.split()[0]
    c = mix.setdefault(phase, collections.Counter())
    if op.startswith("v_") and "f64" in op: c["f64"] += 1
    elif op.startswith("v_accvgpr") or op in ("v_readlane_b32", "v_writelane_b32"): c["spill"] += 1
    elif op.startswith("v_"): c["valu32"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_mov"): c["smov"] += 1
    elif op.startswith("s_load"): c["smem"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    else: c["vmem"] += 1
names = {"pre": "loop head / emit setup", "0": "attitude, n_e, lat/lon atan2, geoid", "11": "kinematics rest + 9 emits", "1": "air data", "2": "aero: angles, filters, locate",
         "12": "aero: lookups, coefficients, wrench", "3": "gear", "4": "gear", "5": "propeller", "9": "engine head", "10": "engine chain + emit", "6": "fuel", "7": "mass", "8": "dynamics + 6 emits", "20": "loop tail, f_step!"}
keys = ["f64", "valu32", "spill", "lds", "smov", "salu", "smem", "wait", "vmem"]
print("%-44s %s  total" % ("phase (instructions AFTER marker k)", " ".join("%6s" % k for k in keys)))
tot = collections.Counter()
for ph, c in mix.items():
    print("%-4s%-40s %s  %5d" % (ph, names.get(ph, ""), " ".join("%6d" % c[k] for k in keys), sum(c.values())))
    tot.update(c)
print("%-44s %s  %5d" % ("all", " ".join("%6d" % tot[k] for k in keys), sum(tot.values())))
