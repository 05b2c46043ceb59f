import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb
from bench import lattice, N_PER_GPU, DT
n = N_PER_GPU
w = fb.BatchedWorld(n)
EAS, h, psi = lattice(0)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
print("trim ok", w.trim_success.mean(), "status after trim", np.unique(w.status, return_counts=True))
x0 = w.x
sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
fb.step(sim, 0.5); w.sync()
st = w.status
print("status after 50 steps", np.unique(st, return_counts=True))
x1 = w.x
bad = np.nonzero(st)[0]
if len(bad):
    i = bad[0]; print("first bad", i, "x0", x0[:, i], "x1", x1[:, i], "EAS,h,psi", EAS[i], h[i], psi[i], "trim ok?", w.trim_success[i])
print("nan count in x1", np.isnan(x1).sum(), "by state", np.isnan(x1).sum(axis=1))
