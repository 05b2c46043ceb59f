#!/usr/bin/env python3
"""Where ONE-step launches of the ground-capable Cessna172Xv2 pass spend their cycles (diagnostic build; wave 0 of workgroup 0):
    python __graft_entry__.py --diagnostic-variant stamp -DFB_STAMP
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_stamp.so python tools/stamp_ground_launch.py [n] [steps per launch]
Every stamp drains the wave's outstanding memory operations (tools/stamp_profile.py), so a phase is charged the latency of what it requested."""
import ctypes as C
import os
import sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tools"))
argv = sys.argv[1:]
import ground_launch_anatomy as gla   # (parked(); its own sweep runs only as __main__)
fb = gla.fb
n = int(argv[0]) if argv else 65536
k = int(argv[1]) if len(argv) > 1 else 1
w = gla.parked(n)
sim = fb.Simulation(w, dt=0.02, Δt=0.02, save_on=False, steps_per_launch=k)
fb.step(sim, 1.0); w.sync()
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
fb.lib.fb_debug_stamps(None, None, 1)
fb.step(sim, 2.0); w.sync()
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
launches = max(cnt[31], 1)
names = {13: "(entry: the gap since the previous launch's end — not this launch's time)", 14: "tables staged into LDS (two barriers)", 15: "x_n: 27 rows from memory into the LDS panel",
         30: "inputs, actuator commands, payload sums, the carried k1 (27 rows) -> loop entry", 31: "behind the last in-loop stamp: k1 and state written back, brakes, status",
         0: "loop tail: f_step!, stage machine", 26: "control laws (x2_periodic, all of it)"}
inloop = [s for s in range(32) if cnt[s] and s not in (13, 14, 15, 30, 31)]
tot = sum(acc[s] for s in range(32) if s != 13)
print(f"n = {n}, {k} step(s) per launch, {launches} launches stamped; cycles per LAUNCH of wave 0 / workgroup 0 (shader-clock cycles, s_memtime)")
for s in (14, 15, 30):
    print(f"  {s:2d} {names[s]:95s} {acc[s] / launches:9.0f} cycles  {100 * acc[s] / tot:5.1f} %")
ev = sum(acc[s] for s in inloop if s not in (0, 21, 22, 23, 24, 25, 26, 27, 28, 29))
ctl = sum(acc[s] for s in (21, 22, 23, 24, 25, 26, 27, 28, 29))
print(f"     {'evaluations of f_ode! (all phases)':95s} {ev / launches:9.0f} cycles  {100 * ev / tot:5.1f} %   ({cnt[11] / launches:.1f} per launch)")
print(f"     {'control laws (slots 21-29)':95s} {ctl / launches:9.0f} cycles  {100 * ctl / tot:5.1f} %")
print(f"   0 {names[0]:95s} {acc[0] / launches:9.0f} cycles  {100 * acc[0] / tot:5.1f} %")
print(f"  31 {names[31]:95s} {acc[31] / launches:9.0f} cycles  {100 * acc[31] / tot:5.1f} %")
print(f"  total {tot / launches:.0f} cycles per launch; slot 13 (between launches) {acc[13] / launches:.0f}")
print("  raw:", {s: (int(acc[s] // launches), int(cnt[s] // launches)) for s in range(32) if cnt[s]})
