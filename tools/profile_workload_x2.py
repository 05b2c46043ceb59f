#!/usr/bin/env python3
"""Workload for the Cessna172Xv2 PMC passes (tools/collect_profile_x2.sh): bench.py's extra.x2 configuration — 524 288 aircraft,
README example 2 scenario, dt = 0.01, Δt = 0.02, 50 steps per launch — without torch and without the oracle leg."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb
from bench import N_TOTAL, DT
inner = int(sys.argv[1]) if len(sys.argv) > 1 else 50
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = N_TOTAL // 2
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=inner)
fb.init(sim, fb.TrimParameters())
w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
for _ in range(launches):
    fb.step(sim, inner * DT)
w.sync()
print("done", (w.status != 0).sum())
