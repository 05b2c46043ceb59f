#!/usr/bin/env python3
"""Throughput of the headline stepper against the batch size on one GPU (bench.py's lattice cut to N, fp64, dt = 0.01, 50 RK4 steps per
launch): what one GPU's shard delivers when BASELINE's 1 048 576 aircraft are cut over 1, 2, 4, 8 GPUs (strong scaling) — and below."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
EAS, h, psi, _ = bench.lattice(0)
for n in (1 << 20, 1 << 19, 1 << 18, 1 << 17, 1 << 16, 1 << 15, 1 << 14):
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=EAS[:n], h_e=h[:n], ψ_nb=psi[:n]))
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 1.0); w.sync()
    fb.lib.fb_timing_begin(w._h)
    fb.step(sim, 10.0); w.sync()
    ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    per = ms.value / nl.value
    wgs = (n + 255) // 256
    print("N = %8d (%5d workgroups of 512 threads, %.1f per CU): %7.3f ms per 50-step launch, %.3e aircraft-steps/s, %.1f %% of the N = 1 M rate per aircraft"
          % (n, wgs, wgs / 256, per, n * 50 / (per * 1e-3), 100 * (n * 50 / (per * 1e-3)) / 3.65e9))
    w.close()
