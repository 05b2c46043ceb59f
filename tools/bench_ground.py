import sys, os, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flight.jl_amd"))
import flightbatch as fb
n = 262144
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=np.full(n, 45.0), h_e=np.full(n, 1000.0)))
x = w.x; s = w.s
x[21:27] = 0; x[12:16] = np.array([1.0, 0, 0, 0])[:, None]
fb.f_ode(w); K = fb.K
# put on the ground: h such that wheels slightly compressed, terrain at 0
y = w.y
x[20] += 1.85 - y[K["FB_Y_KIN"] + 21]
x[9] = 0; s[1] = 0
w.set_state(x, s)
u = w.u; u[0] = 0.0; w.u = u
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 1.0); w.sync()
fb.f_ode(w); y = w.y
print("on ground frac", ((y[K["FB_Y_LDG"]+1] + y[K["FB_Y_LDG"]+12] + y[K["FB_Y_LDG"]+23]) > 0).mean(), "status", (w.status != 0).sum())
t0 = time.time(); fb.step(sim, 2.0); w.sync(); dt = time.time() - t0
print(f"ground: {n*200/dt:.3e} aircraft-steps/s")
if "short" in sys.argv[1:]:   # what a launch costs besides its steps (per-launch HIP events: both passes of a launch)
    for k in (1, 2, 4, 8, 50):
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
        fb.step(sim, 0.8); w.sync()
        nl = 80 // k
        fb.lib.fb_timing_begin_per_launch(w._h, nl)
        fb.step(sim, 0.01 * k * nl); w.sync()
        tot = C.c_float(); cnt = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(tot), C.byref(cnt))
        ms = (C.c_float * nl)(); got = C.c_int64()
        fb._lib.check(fb.lib.fb_timing_launches(w._h, ms, nl, C.byref(got)))
        m = float(np.median(np.array(ms[:got.value])))
        print(f"ground, {k:2d} step(s) per launch: {m * 1e3:8.1f} us per launch, {m / k * 1e3:7.1f} us per step, {n * k / (m * 1e-3):.3e} aircraft-steps/s")
