#!/usr/bin/env python3
"""How long each role of the wave-specialised stepper waits for the other (diagnostic build):
    python __graft_entry__.py --diagnostic-variant waitprof -DFB_STAMP -DFB_DUO_WAITPROF
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_waitprof.so python tools/duo_waitprof.py
Wave 0 (role P) and wave 4 (role D) of workgroup 0: cycles spent in each of the three waits of an evaluation (duo_wait, c172_kernels.hpp)."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
n = 1 << 18
w = fb.BatchedWorld(n)
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.init(sim, fb.TrimParameters())
fb.step(sim, 0.5); w.sync()
fb.lib.fb_debug_stamps(None, None, 1)
fb.lib.fb_timing_begin(w._h)
fb.step(sim, 2.0); w.sync()
ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
names_p = ["T: role D's control words, previous evaluation's rows", "V: velocity at the propeller", "X: role D has read wrench and fuel row"]
names_d = ["R: role P has read its state rows (previous evaluation complete)", "A: density, orthometric altitude", "W: propeller wrench"]
evals = cnt[0]
print("launch: %.3f ms per 50 steps of %d aircraft; evaluations profiled: %d" % (ms.value / nl.value, n, evals))
for role, base, names in (("P (wave 0)", 0, names_p), ("D (wave 4)", 8, names_d)):
    tot = 0.0
    for k in range(3):
        if cnt[base + k]:
            per = acc[base + k] / evals
            tot += per
            print("%-12s wait %-66s %8.1f cycles per evaluation (x %d)" % (role, names[k], per, cnt[base + k]))
    print("%-12s total wait %8.1f cycles per evaluation" % (role, tot))
