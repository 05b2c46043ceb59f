"""The workload of the launch-fixed-cost measurement: 0.2 s of flight of 1,048,576 Cessna172Sv0 at 1, 2 and 5 RK4 steps per launch.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k1 -- python3 tools/profile_small_launches.py
The kernel times of k_step_duo at the three launch sizes give the cost of a launch besides its steps (profiles/r04_ab_start_fence.txt)."""
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
