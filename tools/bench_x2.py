#!/usr/bin/env python3
"""Config 4 of BASELINE.json on ONE GPU's share: 524 288 Cessna172Xv2 (N = 4 194 304 over 8 GPUs), dt = 0.01, Δt = 0.02, the
scenario of the reference's README example 2 (default trim, wind N = 1, E = 0.5 m/s, lon EAS_clm with clm_ref = 2 m/s, lat φ_β with
φ_ref = 30°). Prints one JSON line in bench.py's format (metric aircraft-steps/s). Not the driver's bench: run by hand /
under rocprofv3 (`python3 tools/bench_x2.py [n] [seconds_per_timed_block] [blocks]`)."""
import ctypes as C
import json
import os
import sys
import time
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
block_s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 20
DT, RATIO = 0.01, int(os.environ.get("FB_X2_RATIO", "2"))   # (the config is RATIO = 2; other values only to separate per-launch from per-step cost)
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=DT, Δt=DT * RATIO, save_on=False, steps_per_launch=RATIO)
fb.init(sim, fb.TrimParameters())
assert w.trim_success.all()
w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = np.deg2rad(30.0)
fb.step(sim, 2 * block_s); w.sync()       # warm-up
fb.lib.fb_timing_begin(w._h)
t0 = time.perf_counter()
for _ in range(blocks):
    fb.step(sim, block_s)
w.sync()
el = time.perf_counter() - t0
ms = C.c_float(); nl = C.c_int64()
fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
steps = int(round(block_s / DT)) * blocks
bad = int((w.status != 0).sum())
fb.f_ode(w)
y = w.y
line = {"metric": "aircraft-steps/sec", "value": n * steps / el, "unit": "aircraft-steps/s", "n_gpus": 1, "steps": blocks,
        "ms_per_step": el / blocks * 1e3, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"N={n} Cessna172Xv2, default trim, README example 2 scenario, dt=0.01, Δt=0.02 (BASELINE.json configs[3], one GPU's share)",
                   "rk4_steps_per_launch": RATIO, "terminated_aircraft": bad},
        "stream_ms_per_rk4_step": ms.value / steps, "stepping_launches": nl.value,
        "final": {"climb_rate": float(-y[fb.K["FB_Y_KIN"] + 36].mean()), "phi_deg": float(np.rad2deg(y[2].mean())), "EAS": float(y[fb.K["FB_Y_AIR"] + 20].mean())},
        "roofline": {"bound": "hbm", "achieved": 756.0 * n * steps / el / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "note": "algorithmic bytes = 756 B per aircraft-step for C172Xv2 (SURVEY §8d)"}}
line["roofline"]["frac"] = line["roofline"]["achieved"] / 8000.0
print(json.dumps(line))
