"""A/B of the two airborne fp64 steppers on the GPU: the one-wave-per-SIMD k_step_air (FLIGHTBATCH_DUO=0) against the
wave-specialised k_step_duo (FLIGHTBATCH_DUO=1), same initial condition: largest difference of the states after `--steps` steps,
status words, and the time per launch of each.   python tools/duo_check.py [--n 65536] [--steps 200] [--time-n 1048576]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "flight.jl_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--inner", type=int, default=50)
ap.add_argument("--time-n", type=int, default=1048576)
ap.add_argument("--launches", type=int, default=10)
args = ap.parse_args()
import flightbatch as fb

def world(n, duo):
    os.environ["FLIGHTBATCH_DUO"] = "1" if duo else "0"
    return fb.BatchedWorld(n)

rng = np.random.default_rng(5)
n = args.n
EAS = rng.uniform(35.0, 60.0, n); h = rng.uniform(200.0, 3000.0, n); psi = rng.uniform(-np.pi, np.pi, n)
ref = world(n, False)
fb.f_init(ref, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
x0, s0, u0, ui0 = ref.x, ref.s, ref.u, ref.ui
# perturb the controls so that the trajectory is not a steady state
u0 = u0.copy(); u0[fb.K["FB_U_ELEVATOR"]] += rng.uniform(-0.05, 0.05, n); u0[fb.K["FB_U_AILERON"]] += rng.uniform(-0.05, 0.05, n)
out = {}
for duo in (False, True):
    w = world(n, duo)
    w.set_state(x0, s0); w.u = u0; w.ui = ui0
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=args.inner)
    fb.step(sim, args.steps * 0.01)
    w.sync()
    out[duo] = (w.x, w.s, w.status)
xa, sa, sta = out[False]; xb, sb, stb = out[True]
scale = np.maximum(np.abs(xa), 1e-3)
err = np.abs(xa - xb) / scale
print("n %d steps %d: max scaled |x_duo - x_air| = %.3e (row %d), status differs on %d, s differs on %d, status!=0: %d / %d, nonfinite %d" % (
    n, args.steps, np.nanmax(err), int(np.nanargmax(err.max(axis=1))), int((sta != stb).sum()), int((sa != sb).sum()), int((sta != 0).sum()), int((stb != 0).sum()),
    int((~np.isfinite(xb)).sum())), flush=True)
# timing
n = args.time_n
EAS = np.tile(EAS, n // len(EAS) + 1)[:n]; h = np.tile(h, n // len(h) + 1)[:n]; psi = np.tile(psi, n // len(psi) + 1)[:n]
import ctypes as C
for duo in (False, True, False, True):
    w = world(n, duo)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=args.inner)
    for _ in range(2): fb.step(sim, args.inner * 0.01)
    w.sync()
    fb.lib.fb_timing_begin(w._h)
    for _ in range(args.launches): fb.step(sim, args.inner * 0.01)
    w.sync()
    ms = C.c_float(); nl = C.c_int64()
    fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    per = ms.value / max(nl.value, 1)
    print("duo=%d  n %d: %.3f ms per launch of %d steps = %.3e aircraft-steps/s, status!=0 %d" % (duo, n, per, args.inner, n * args.inner / (per * 1e-3), int((w.status != 0).sum())), flush=True)
    del sim, w
