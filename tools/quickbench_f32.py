"""The fp32 stepper at 1, 10 and 50 RK4 steps per launch (1,048,576 aircraft on the bench lattice): ms per launch and aircraft-steps/s.
    python tools/quickbench_f32.py"""
import sys, time, numpy as np, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb, ctypes as C, bench
n = 1 << 20
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n, dtype="f32")
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
for k in (1, 10, 50):
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
    fb.step(sim, 0.5); w.sync()
    fb.lib.fb_timing_begin(w._h)
    fb.step(sim, 1.0); w.sync()
    ms=C.c_float(); nl=C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    print(f"f32 k={k}: {ms.value/nl.value:.3f} ms per launch -> {n*100/(ms.value*1e-3):.3e}/s")
