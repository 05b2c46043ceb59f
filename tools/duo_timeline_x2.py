#!/usr/bin/env python3
"""The two-row Gantt chart of tools/duo_timeline.py for the Cessna172Xv2 instance of the wave-specialised stepper (k_step_duo<KIN, true>),
bench.py's configs[3] scenario (autopilot at Δt = 2 dt), with the phases of a control update:
    python __graft_entry__.py --diagnostic-variant timeline -DFB_STAMP -DFB_DUO_TIMELINE
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_timeline.so python tools/duo_timeline_x2.py
Cycles from the top of an evaluation (role P: behind its wait for T), averaged over the evaluations of the launches (the update marks over
the tapped evaluations only: one in eight)."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
n = 1 << 18
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
fb.init(sim, fb.TrimParameters())
w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
fb.step(sim, 1.0); w.sync()
fb.lib.fb_debug_stamps(None, None, 1)
fb.lib.fb_timing_begin(w._h)
fb.step(sim, 2.0); w.sync()
ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
P = {12: "stage positions, aerodynamic sums stored", 1: "geoid height (R published before it)", 2: "ISA atmosphere", 11: "rho, h_o put: at A", 3: "past V",
     4: "propeller coefficients and angles", 5: "wrench formed", 6: "engine head done", 8: "engine lookups", 9: "engine done: at X", 10: "fuel row emitted: end of the evaluation",
     13: "[update] arrives at U", 14: "[update] past U", 7: "[update] longitudinal half done", 15: "arrives at the top (T)"}
D = {1: "head: attitude, wind-relative velocity, V", 2: "airflow angles (2 atan2)", 3: "R waited, sums fetched, knots located", 4: "lookups", 5: "fuel row read",
     6: "kinematics rows emitted (9)", 7: "mass properties, gravity, Earth rate: at A", 8: "aerodynamics, propeller-free dynamics: at W", 9: "past W",
     10: "rigid-body dynamics", 11: "velocity rows emitted (6): end of the evaluation", 12: "[update] f_step! done, flags written: at U", 13: "[update] lateral half done",
     14: "[update] past F", 15: "arrives at the top (T)"}
print("launch: %.3f ms per 50 steps of %d aircraft (%d launches)" % (ms.value / nl.value, n, nl.value))
for role, names, base, order in (("P (wave 0)", P, 0, [12, 1, 2, 11, 3, 4, 5, 6, 8, 9, 10, 13, 14, 7, 15]), ("D (wave 4)", D, 16, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])):
    prev = 0.0
    for k in order:
        if cnt[base + k]:
            t = acc[base + k] / cnt[base + k]
            print("%-11s %8.0f  (+%6.0f)  x%-6d %s" % (role, t, t - prev, cnt[base + k], names[k]))
            prev = t
