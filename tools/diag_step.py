import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import flightbatch as fb
from oracle_binding import Oracle
from test_gpu_parity import lattice_trim_params, state_scale
o = Oracle()
n = 4096
w = fb.BatchedWorld(n)
fb.f_init(w, lattice_trim_params(fb, n, seed=11))
rng = np.random.default_rng(2)
x = w.x
x[21:24] += rng.normal(0, 0.02, (3, n)); x[24:27] += rng.normal(0, 1.0, (3, n))
w.set_state(x, w.s)
x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
for nst in (1, 2, 10, 100):
    w.set_state(x0, s0)
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=nst)
    fb.step(sim, nst * 0.01); w.sync()
    xg = w.x
    xo, so, st = o.step(x0, u0, ui0, s0, o.default_env(), 0.01, nst)
    e = np.abs(xg - xo) / state_scale(xo)
    k, i = np.unravel_index(e.argmax(), e.shape)
    print(f"nsteps={nst}: max scaled err {e.max():.3e} at state {k} aircraft {i}; abs {abs(xg[k,i]-xo[k,i]):.3e} value {xo[k,i]:.6g}; per-state max:", " ".join(f"{v:.1e}" for v in e.max(axis=1)))
    print("   s equal:", (w.s == so).all(), " worst aircraft x0:", x0[:, i][[0,1,9,20,21,22,23,24,25,26]])

print("---- single RHS at x0 ----")
w.set_state(x0, s0)
xd = np.zeros((27, n)); fb.f_ode(w, xd); y = w.y
xdo, yo, sto = o.f_ode(x0, u0, ui0, s0, o.default_env())
d = np.abs(xd - xdo)
print("xdot abs err max per state:", " ".join(f"{v:.1e}" for v in d.max(axis=1)))
names = {"kin":(0,40),"air":(40,62),"aero":(62,78),"ldgL":(78,89),"pwp":(111,133),"fuel":(133,134),"dyn":(134,174)}
for k,(a,b) in names.items():
    e = np.abs(y[a:b]-yo[a:b]); r = e/np.maximum(np.abs(yo[a:b]),1e-30)
    print(k, "abs:", " ".join(f"{v:.1e}" for v in e.max(axis=1)))
    print(k, "rel:", " ".join(f"{v:.1e}" for v in np.where(np.abs(yo[a:b]).max(axis=1)>0, (e/np.maximum(np.abs(yo[a:b]),1e-300)).max(axis=1), 0)))

print("---- RHS at stage-2 state x0 + dt/2 k1 ----")
x2 = x0 + 0.005 * xdo
w.set_state(x2, s0)
xd2 = np.zeros((27, n)); fb.f_ode(w, xd2)
xdo2, yo2, _ = o.f_ode(x2, u0, ui0, s0, o.default_env())
print("xdot abs err max per state:", " ".join(f"{v:.1e}" for v in np.abs(xd2 - xdo2).max(axis=1)))
print("---- host-composed RK4 step using GPU f_ode vs oracle step vs k_step ----")
def rk4(f, x):
    k1 = f(x); k2 = f(x + 0.005*k1); k3 = f(x + 0.005*k2); k4 = f(x + 0.01*k3)
    return x + (0.01/6)*(2*(k2+k3) + (k1+k4))
def fg(x):
    w.set_state(x, s0); out = np.zeros((27, n)); fb.f_ode(w, out); return out
def fo(x):
    return o.f_ode(x, u0, ui0, s0, o.default_env())[0]
xg_host = rk4(fg, x0); xo_host = rk4(fo, x0)
w.set_state(x0, s0); sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=1); fb.step(sim, 0.01); w.sync(); xk = w.x
xo1, _, _ = o.step(x0, u0, ui0, s0, o.default_env(), 0.01, 1)
for nm, a, b in (("gpu_host vs oracle_host", xg_host, xo_host), ("k_step vs gpu_host", xk, xg_host), ("oracle_step vs oracle_host", xo1, xo_host)):
    print(nm, " ".join(f"{v:.1e}" for v in np.abs(a-b).max(axis=1)))
