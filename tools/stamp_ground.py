#!/usr/bin/env python3
"""Per-phase cycle profile of ONE wave of the ground-capable pass on a batch that sits on the ground, engine off (diagnostic build):
    python __graft_entry__.py --diagnostic-variant stamp -DFB_STAMP
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_stamp.so python tools/stamp_ground.py
Same fences as tools/stamp_profile.py; every stamp drains the wave's outstanding memory operations, so the cycles are an upper bound
per phase and the split is what is of interest."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402
n = 1 << 16
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=np.full(n, 45.0), h_e=np.full(n, 1000.0)))
x = w.x; s = w.s; K = fb.K
x[21:27] = 0; x[12:16] = np.array([1.0, 0, 0, 0])[:, None]
fb.f_ode(w)
x[20] += 1.85 - w.y[K["FB_Y_KIN"] + 21]
x[9] = 0; s[1] = 0
w.set_state(x, s)
u = w.u; u[0] = 0.0; w.u = u
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 1.0); w.sync()
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
fb.lib.fb_debug_stamps(None, None, 1)
fb.step(sim, 2.0); w.sync()
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
names = {0: "loop tail: emits' bookkeeping, f_step!, stage machine (20 -> 0)", 11: "kinematics head: attitude, n_e, lat/lon atan2, geoid", 1: "kinematics rest + emits",
         2: "air data", 12: "aero: airflow angles, filters, knot location", 3: "aero: lookups, coefficients, wrench", 4: "gear units (x3, per unit)",
         5: "gear tail", 9: "propeller", 10: "engine head", 6: "engine chain", 7: "fuel", 8: "mass properties", 20: "dynamics + emits"}
order = [0, 11, 1, 2, 12, 3, 4, 5, 9, 10, 6, 7, 8, 20]
evals = max(cnt[11], 1)
tot = sum(acc[k] for k in order)
for k in order:
    print("%3d %-62s %9.1f cycles/eval %5.1f %%  (x %d)" % (k, names[k], acc[k] / evals, 100.0 * acc[k] / max(tot, 1), cnt[k]))
print("total %.1f cycles per evaluation over %d evaluations; on the ground: %.3f" % (tot / evals, evals,
      ((w.y[K["FB_Y_LDG"] + 1] + w.y[K["FB_Y_LDG"] + 12] + w.y[K["FB_Y_LDG"] + 23]) > 0).mean() if (fb.f_ode(w) or True) else 0))
