#!/usr/bin/env python3
"""The headline batch (bench.lattice(0), 1 048 576 Cessna172Sv0, 50 steps per launch) with a per-aircraft environment (fb_set_env: every
aircraft in its own wind, sea-level T / p and terrain elevation) against the batch-wide block; the Cessna172Xv2 share of configs[3] likewise.
    python tools/bench_env.py [WA|ECEF|NED]           (FLIGHTBATCH_DUO=0: the one-wave kernels)"""
import ctypes as C
import os
import sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb
from bench import lattice, N_TOTAL, DT


def env_rows(n, seed=3):
    K = fb.K
    rng = np.random.default_rng(seed)
    e = np.zeros((K["FB_NENV"], n))
    e[K["FB_ENV_WIND_N"]] = rng.uniform(-10, 10, n); e[K["FB_ENV_WIND_E"]] = rng.uniform(-10, 10, n); e[K["FB_ENV_WIND_D"]] = rng.uniform(-1, 1, n)
    e[K["FB_ENV_T_SL"]] = rng.uniform(263.0, 308.0, n); e[K["FB_ENV_P_SL"]] = rng.uniform(98000.0, 103500.0, n)
    e[K["FB_ENV_H_TERRAIN"]] = rng.uniform(0.0, 100.0, n)
    return e


def timed(w, sim, launches=10):
    for _ in range(3):
        fb.step(sim, 50 * DT)
    w.sync()
    fb.lib.fb_timing_begin(w._h)
    for _ in range(launches):
        fb.step(sim, 50 * DT)
    w.sync()
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    return ms.value / max(nl.value, 1)


KIN = sys.argv[1] if len(sys.argv) > 1 else "WA"
print("mechanisation", KIN, "duo" if os.environ.get("FLIGHTBATCH_DUO", "1") != "0" else "one-wave kernels")
for per_aircraft in (False, True):
    n = N_TOTAL
    EAS, h, psi, _ = lattice(0, n)
    w = fb.BatchedWorld(n, kinematics=KIN)
    if per_aircraft:
        w.env = env_rows(n)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h + 200.0, ψ_nb=psi))
    ok = float(w.trim_success.mean())
    sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
    ms = timed(w, sim)
    print(f"Cessna172Sv0 n={n} {'per-aircraft environment' if per_aircraft else 'batch-wide environment   '}: {ms:7.3f} ms per launch, "
          f"{n * 50 / (ms * 1e-3):.4e} aircraft-steps/s (trim success {ok:.4f}, terminated {int((w.status != 0).sum())})", flush=True)
    w.close()
    n = N_TOTAL // 2
    w = fb.Cessna172Xv2World(n, kinematics=KIN)
    if per_aircraft:
        w.env = env_rows(n, 5)
    else:
        w.set_params(wind_ned=(1.0, 0.5, 0.0))
    sim = fb.Simulation(w, dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters(h_e=1250.0))
    w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
    ms = timed(w, sim, 6)
    print(f"Cessna172Xv2 n={n} {'per-aircraft environment' if per_aircraft else 'batch-wide environment   '}: {ms:7.3f} ms per launch, "
          f"{n * 50 / (ms * 1e-3):.4e} aircraft-steps/s (terminated {int((w.status != 0).sum())})", flush=True)
    w.close()
