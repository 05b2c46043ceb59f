import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flightbatch as fb
from oracle_binding import Oracle
oracle = Oracle()
n = 65536
rng = np.random.default_rng(2026)
x = np.zeros((27, n))
x[0] = rng.uniform(-0.3, 0.5, n); x[1] = rng.uniform(-0.3, 0.3, n)
x[8] = rng.uniform(0, 1, n)
x[9] = rng.uniform(0, 320, n); x[10] = rng.uniform(-0.5, 0.5, n); x[11] = rng.uniform(-1, 1, n)
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0); x[12:16] = q * (1 + rng.uniform(-sc, sc, n))
qe = rng.normal(size=(4, n)); qe /= np.linalg.norm(qe, axis=0); x[16:20] = qe * (1 + rng.uniform(-sc, sc, n))
x[20] = np.where(rng.random(n) < 0.2, rng.uniform(11000, 25000, n), rng.uniform(-500, 11000, n))
x[21:24] = rng.normal(0, 0.5, (3, n))
x[24:27] = rng.normal(0, 1, (3, n)); x[24:27] *= rng.uniform(0, 120, n) / np.linalg.norm(x[24:27], axis=0)
s = np.stack([rng.integers(0, 2, n), rng.integers(0, 3, n)]).astype(np.int32)
u = np.zeros((16, n))
u[0] = rng.uniform(-0.2, 1.2, n); u[1] = rng.uniform(-0.2, 1.2, n); u[2:8] = rng.uniform(-1.3, 1.3, (6, n)); u[8] = rng.uniform(-0.2, 1.2, n)
u[9:11] = rng.uniform(0, 1, (2, n)); u[11:16] = rng.uniform(-10, 120, (5, n))
ui = rng.integers(0, 16, n).astype(np.int32)
env = oracle.default_env(T_sl=300.0, p_sl=99000.0, wind=(12.0, -7.0, 1.5), h_trn=-600.0)
w = fb.BatchedWorld(n)
w.set_params(T_sl=300.0, p_sl=99000.0, wind_ned=(12.0, -7.0, 1.5), h_terrain=-600.0)
w.set_state(x, s); w.u = u; w.ui = ui
xd = np.zeros((27, n)); fb.f_ode(w, xd)
xdo, yo, sto = oracle.f_ode(x, u, ui, s, env)
ok = sto == 0
d = np.abs(xd - xdo)[:, ok]
print("scale", sc, "max abs err: vdot", d[24:27].max(), "wdot", d[21:24].max(), "qdot", d[12:20].max(), "hdot", d[20].max(), "others", d[:12].max())
y = w.y; K = fb.K
print("g_c_c out err", np.abs(y[K["FB_Y_DYN"]+37:K["FB_Y_DYN"]+40] - yo[K["FB_Y_DYN"]+37:K["FB_Y_DYN"]+40])[:, ok].max())
