#!/usr/bin/env python3
"""A two-row Gantt chart of one SIMD of the wave-specialised stepper (diagnostic build):
    python __graft_entry__.py --diagnostic-variant timeline -DFB_STAMP -DFB_DUO_TIMELINE
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_timeline.so python tools/duo_timeline.py
When — in shader cycles from the moment the wave leaves the barrier at the top of an evaluation, averaged over the evaluations of a
launch — wave 0 (role P) and wave 4 (role D) of workgroup 0 pass the marked points of rhs_duo() (c172_duo_device.hpp, DUO_MARK)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
n = 1 << 18
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=EAS[:n], h_e=h[:n], ψ_nb=psi[:n]))
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 0.5); w.sync()
fb.lib.fb_debug_stamps(None, None, 1)   # (the marks of the first evaluation after this reset are measured from a stale t0: 1 in 800)
fb.lib.fb_timing_begin(w._h)
fb.step(sim, 2.0); w.sync()
ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
P = {1: "geoid height", 2: "ISA atmosphere (puts rho, h_o next)", 11: "rho, h_o put; arrives at barrier A", 3: "left barrier A", 4: "propeller: coefficients (16 loads) and angles", 5: "wrench formed (puts next)",
     6: "arrives at barrier B (engine head done)", 7: "left barrier B", 8: "engine lookups", 9: "engine done", 10: "fuel row emitted: end", 15: "arrives at the top barrier"}
D = {1: "head: attitude, wind-relative velocity, put v at the propeller", 2: "airflow angles (2 atan2)", 3: "knot locations", 4: "lookups; arrives at barrier A", 5: "left barrier A",
     6: "kinematics rows emitted (9)", 7: "mass properties, gravity, Earth rate", 8: "aerodynamic coefficients, wrench; arrives at barrier B", 9: "left barrier B",
     10: "rigid-body dynamics", 11: "velocity rows emitted (6): end", 15: "arrives at the top barrier"}
print("launch: %.3f ms per 50 steps of %d aircraft" % (ms.value / nl.value, n))
for role, names, base in (("P (wave 0)", P, 0), ("D (wave 4)", D, 16)):
    prev = 0.0
    for k in ([1, 2, 11, 3, 4, 5, 6, 7, 8, 9, 10, 15] if base == 0 else sorted(names)):
        if cnt[base + k]:
            t = acc[base + k] / cnt[base + k]
            print("%-11s %8.0f  (+%6.0f)  %s" % (role, t, t - prev, names[k]))
            prev = t
