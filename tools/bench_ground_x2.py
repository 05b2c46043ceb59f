#!/usr/bin/env python3
"""A batch of Cessna172Xv2 standing on the runway (parked, engine off, brakes set — the start of examples/traffic_pattern.py): every lane in the
ground-capable pass k_step_air<0, true, true, PERENV>. Steps per launch 1 / 10 / 50 (what a launch costs besides its steps), with and without
per-aircraft environment rows, control laws at Δt = dt (as the scripted scenarios run them) and Δt = 2 dt.   python3 tools/bench_ground_x2.py [n]"""
import ctypes as C
import os
import sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, os.path.join(R, "examples"))
import flightbatch as fb
import traffic_pattern as tpat
K = fb.K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for env_rows in (False, True):
    for ratio in (1, 2):
        w = fb.Cessna172Xv2World(n)
        w.set_params(h_terrain=tpat.H_ORTH)
        if env_rows:
            w.set_env(wind_ned=np.stack([np.zeros(n), np.linspace(0, 6, n), np.zeros(n)]))
        sim = fb.Simulation(w, dt=0.02, Δt=0.02 * ratio, save_on=False, steps_per_launch=1)
        LOC, PSI = tpat.LOC, tpat.PSI
        n_e = np.array([np.cos(LOC[0]) * np.cos(LOC[1]), np.cos(LOC[0]) * np.sin(LOC[1]), np.sin(LOC[0])])
        fb.init(sim, fb.TrimParameters(n_e=n_e, h_e=1000.0)); fb.f_ode(w)
        geoid = float((w.y[K["FB_Y_KIN"] + 20] - w.y[K["FB_Y_KIN"] + 21])[0])
        x = np.zeros((K["FB_X2_NX"], n)); x[K["FB_X_FUEL"]] = 0.5
        kq = K["FB_X2_KIN"]
        x[kq:kq + 4] = np.array([np.cos(PSI / 2), 0, 0, np.sin(PSI / 2)])[:, None]
        a = -(LOC[0] + np.pi / 2)
        qz = np.array([np.cos(LOC[1] / 2), 0, 0, np.sin(LOC[1] / 2)]); qy = np.array([np.cos(a / 2), 0, np.sin(a / 2), 0])
        x[kq + 4:kq + 8] = np.array([qz[0] * qy[0], -qz[3] * qy[2], qz[0] * qy[2], qz[3] * qy[0]])[:, None]
        x[kq + 8] = tpat.H_ORTH + geoid + 1.81
        u = np.zeros((K["FB_NU"], n)); u[K["FB_U_MIXTURE"]] = 0.5; u[K["FB_U_M_PILOT"]] = 75; u[K["FB_U_BRAKE_LEFT"]] = 1; u[K["FB_U_BRAKE_RIGHT"]] = 1
        w.set_state(x, np.zeros((2, n), dtype=np.int32)); w.u = u
        w.ui = np.full(n, K["FB_UI_MIXTURE_AUTO"] | K["FB_UI_STEERING_ENGAGED"], dtype=np.int32)
        fb.f_init(w, None)
        for k in (1, 10, 50):
            sim = fb.Simulation(w, dt=0.02, Δt=0.02 * ratio, save_on=False, steps_per_launch=k)
            fb.step(sim, 2.0); w.sync()
            fb.lib.fb_timing_begin(w._h)
            fb.step(sim, 2.0); w.sync()
            ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
            print(f"n={n} env rows {env_rows!s:5} Δt = {ratio} dt, {k:2d} steps per launch: {ms.value / nl.value:8.3f} ms per launch, {ms.value / 100:7.3f} ms per step, {n * 100 / (ms.value * 1e-3):.3e} aircraft-steps/s", flush=True)
        fb.f_ode(w)
        y = w.y
        print("   on ground:", float(((y[K["FB_Y_LDG"] + 1] + y[K["FB_Y_LDG"] + 12] + y[K["FB_Y_LDG"] + 23]) > 0).mean()), "terminated:", int((w.status != 0).sum()))
        w.close()
