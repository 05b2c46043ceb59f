#!/usr/bin/env python3
"""Separates the per-step from the per-control-update cost of the Cessna172Xv2 stepper: 524 288 aircraft, 50 steps per launch, control
period Δt = ratio x dt for ratio in argv (default 1 2 5 10 50). Not a config of BASELINE.json (that is ratio 2): a diagnostic."""
import ctypes as C
import os
import sys
import time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402

n = int(os.environ.get("FB_BENCH_N", 1 << 19))
for ratio in ([int(a) for a in sys.argv[1:]] or [1, 2, 5, 10, 50]):
    w = fb.Cessna172Xv2World(n)
    w.set_params(wind_ned=(1.0, 0.5, 0.0))
    sim = fb.Simulation(w, dt=0.01, Δt=0.01 * ratio, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters())
    w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
    fb.step(sim, 1.0); w.sync()
    fb.lib.fb_timing_begin(w._h)
    t0 = time.perf_counter(); fb.step(sim, 3.0); w.sync(); el = time.perf_counter() - t0
    ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    print(f"ratio {ratio:3d}: {n * 300 / el:.4e} aircraft-steps/s, {ms.value / nl.value:.3f} ms per 50-step launch, terminated {int((w.status != 0).sum())}", flush=True)
    w.close()
