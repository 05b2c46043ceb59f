#!/usr/bin/env python3
"""Throughput of the three kinematic mechanisations of Cessna172Sv0 on bench.py's lattice (1 048 576 aircraft, fp64, dt = 0.01, 50 RK4
steps per launch), each on the wave-specialised k_step_duo (the default) and on the one-wave-per-SIMD k_step_air (FLIGHTBATCH_DUO=0)."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
n = bench.N_TOTAL
EAS, h, psi, _ = bench.lattice(0)
for kin, duo in (("WA", "1"), ("ECEF", "1"), ("NED", "1"), ("WA", "0"), ("ECEF", "0"), ("NED", "0")):
    os.environ["FLIGHTBATCH_DUO"] = duo
    w = fb.BatchedWorld(n, kinematics=kin)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 1.0); w.sync()
    fb.lib.fb_timing_begin(w._h)
    fb.step(sim, 5.0); w.sync()
    ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    per = ms.value / nl.value
    print("%-4s %-28s %7.3f ms per 50-step launch  %.3e aircraft-steps/s  (terminated: %d)" % (
        kin, "k_step_duo" if duo == "1" else "k_step_air (one wave / SIMD)", per, n * 50 / (per * 1e-3), int((w.status != 0).sum())))
    w.close()
