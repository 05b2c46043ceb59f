import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "flight.jl_amd")); 
import flightbatch as fb, ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1<<20
t0=time.time(); w = fb.BatchedWorld(n); print("create+tables", time.time()-t0)
t0=time.time(); fb.f_init(w, fb.TrimParameters(EAS=np.linspace(35,55,n), h_e=np.linspace(200,3000,n), ψ_nb=np.linspace(-3,3,n))); print("trim", time.time()-t0, w.trim_success.mean())
for k in (1, 10, 50):
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
    fb.step(sim, 0.5); w.sync()
    fb.lib.fb_timing_begin(w._h)
    t0=time.time(); fb.step(sim, 1.0); w.sync(); dt=time.time()-t0
    ms=C.c_float(); nl=C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    print(f"k={k}: {n*100/dt:.3e} aircraft-steps/s wall; events {ms.value:.2f} ms over {nl.value} launches -> {n*100/(ms.value*1e-3):.3e}/s")
print("status nonzero:", (w.status!=0).sum())
