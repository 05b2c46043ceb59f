import sys, os, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb, bench
n = 1 << 20
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
for k in (1, 2, 5):
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
    fb.step(sim, 0.2); w.sync()
