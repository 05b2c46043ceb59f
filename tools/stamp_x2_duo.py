#!/usr/bin/env python3
"""Per-phase cycle profile of the two halves of an in-kernel control update of the wave-specialised Cessna172Xv2 stepper (diagnostic build):
    python __graft_entry__.py --diagnostic-variant stamp -DFB_STAMP;  FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_stamp.so python tools/stamp_x2_duo.py
Wave 0 (role P: the longitudinal half) and wave 4 (role D: the lateral half) of workgroup 0; every fence drains the wave's memory traffic."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
n = 1 << 18
spl = int(sys.argv[1]) if len(sys.argv) > 1 else 50   # steps per launch (1: the update of a short launch, profiles/r06_duo_x2_phases.txt)
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=spl)
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
print("launch: %.3f ms per %d step(s) of %d aircraft" % (ms.value / nl.value, spl, n))
names = {22: "lon: entry, arguments, burst of loads", 23: "lon: guidance", 27: "lon: inputs, mode logic", 28: "lon: outer loops (PID lookups + runs)", 29: "lon: LQR gain lookup",
         24: "lon: LQR run, stores", 16: "lat: entry, arguments, burst of loads", 17: "lat: guidance", 18: "lat: gains gathered (LQR + PID)", 19: "lat: laws, stores"}
for k in (22, 23, 27, 28, 29, 24, 16, 17, 18, 19):
    if cnt[k]: print("%2d %-50s %9.1f cycles (x %d)" % (k, names[k], acc[k] / cnt[k], cnt[k]))
