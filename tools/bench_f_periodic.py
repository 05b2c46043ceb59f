#!/usr/bin/env python3
"""Time of the stand-alone fb_f_periodic verb (k_x2_ctl: guidance + control laws for every aircraft) on 524 288 Cessna172Xv2."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb
n = 1 << 19
w = fb.Cessna172Xv2World(n)
sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
fb.init(sim, fb.TrimParameters())
w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β)
for _ in range(3): fb.f_periodic(w)
w.sync()
t0 = time.perf_counter()
for _ in range(50): fb.f_periodic(w)
w.sync()
print("fb_f_periodic: %.3f ms per call of %d aircraft" % ((time.perf_counter() - t0) / 50 * 1e3, n))
