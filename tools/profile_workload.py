#!/usr/bin/env python3
"""Workload for the PMC passes: the bench configuration (N = 1 048 576 lattice, inner = 50) without torch,
plus one fb_f_ode launch whose byte counts are known exactly (used to calibrate FETCH_SIZE / WRITE_SIZE for
this access pattern, as MI355X_MICROARCH.md §HBM prescribes)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb
from bench import lattice, N_TOTAL as N_PER_GPU, DT
inner = int(sys.argv[1]) if len(sys.argv) > 1 else 50
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
w = fb.BatchedWorld(N_PER_GPU, dtype=dtype)
EAS, h, psi, _ = lattice(0)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
xd = np.zeros((27, N_PER_GPU)); fb.f_ode(w, xd)          # calibration launch (k_f_ode with xdot)
sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=inner)
for _ in range(launches):
    fb.step(sim, inner * DT)
w.sync()
print("done", (w.status != 0).sum())
