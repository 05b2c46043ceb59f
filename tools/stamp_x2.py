#!/usr/bin/env python3
"""Per-phase cycle profile of ONE wave's in-kernel control update (Cessna172Xv2; diagnostic build with -DFB_STAMP):
    tools/build_variant.sh stamp -DFB_STAMP;  FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_stamp.so python tools/stamp_x2.py"""
import ctypes as C, os, sys, types
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
orig_step = fb.step
state = {"armed": False}
def step(sim, *a, **k):
    if not state["armed"]:
        state["armed"] = True
        fb.lib.fb_debug_stamps(None, None, 1)
    return orig_step(sim, *a, **k)
fb.step = step
args = types.SimpleNamespace(x2_inner=50)
bench.extra_x2(fb, C, args, bench.time_x2(fb, None, None, C, args))
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
names = {21: "end of evaluation -> entry of x2_periodic (bookkeeping, arguments, call)", 22: "record burst (94 rows) issued", 23: "guidance", 27: "lon: inputs, mode logic",
         28: "lon: outer loops (3 PID lookups + runs, integrator, cos / tan)", 29: "lon: LQR gain lookup", 24: "lon: LQR run, stores", 25: "lat", 26: "return, reload commands"}
for k in (21, 22, 23, 27, 28, 29, 24, 25, 26):
    if cnt[k]: print("%2d %-78s %9.1f cycles (x %d)" % (k, names[k], acc[k] / cnt[k], cnt[k]))
# the evaluations of the stepping loop between the updates, phase by phase (same fences as tools/stamp_profile.py)
rhs_names = {0: "loop tail + emit setup (20 -> 0)", 11: "kinematics head", 1: "kinematics rest + emits", 2: "air data", 12: "aero: angles, filters, knots",
             3: "aero: lookups, coefficients, wrench", 4: "gear unit (x3)", 5: "gear tail", 9: "propeller", 10: "engine head", 6: "engine chain",
             7: "fuel", 8: "mass properties", 20: "dynamics + emits"}
order = [0, 11, 1, 2, 12, 3, 4, 5, 9, 10, 6, 7, 8, 20]
evals = max(cnt[11], 1)
tot = sum(acc[k] for k in order)
for k in order:
    print("%3d %-50s %9.1f cycles/eval %5.1f %%" % (k, rhs_names[k], acc[k] / evals, 100.0 * acc[k] / max(tot, 1)))
print("total %.1f cycles per evaluation over %d evaluations (the interval 20 -> 0 contains the bookkeeping between evaluations: actuators, f_step!, the update's call)" % (tot / evals, evals))
