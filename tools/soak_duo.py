"""One-off long run of the default airborne fp64 stepper against the CPU oracle: 100 s of flight (10 000 RK4 steps) for a stratified
sample of bench.py's lattice, plus invariants over a larger batch.   python tools/soak_duo.py [n_oracle=1024] [n_batch=65536] [WA|ECEF|NED]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import flightbatch as fb
import bench
from oracle_binding import Oracle
n_or = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
kin = sys.argv[3] if len(sys.argv) > 3 else "WA"
nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
EAS, h, psi, cell = bench.lattice(0)
EAS, h, psi = EAS[:n], h[:n], psi[:n]
w = fb.BatchedWorld(n, kinematics=kin)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
x0, s0, u0, ui0 = w.x, w.s, w.u.copy(), w.ui
rng = np.random.default_rng(3)
u0[fb.K["FB_U_ELEVATOR"]] += rng.uniform(-0.02, 0.02, n); u0[fb.K["FB_U_AILERON"]] += rng.uniform(-0.02, 0.02, n)   # phugoid + roll
w.u = u0
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
t0 = time.time(); fb.step(sim, 100.0); w.sync(); print("GPU (%s): %d aircraft x 10000 steps in %.2f s" % (kin, n, time.time() - t0), flush=True)
xg, sg, stg = w.x, w.s, w.status
idx = np.sort(rng.choice(n, n_or, replace=False))
orc = Oracle()
t0 = time.time()
orc.lib.fo_set_kinematics(fb.K["FB_KIN_" + kin])      # the oracle keeps 27 rows for every mechanisation (unused kinematic rows zero)
xo27 = np.zeros((27, n_or)); xo27[:12 + nk] = x0[:12 + nk, idx]; xo27[21:] = x0[12 + nk:, idx]
xo27, so, sto, tso, two = orc.step_term(xo27, u0[:, idx], ui0[idx], s0[:, idx], orc.default_env(), 0.01, 10000)
orc.lib.fo_set_kinematics(fb.K["FB_KIN_WA"])
xo = np.vstack([xo27[:12 + nk], xo27[21:]])
print("oracle: %d aircraft x 10000 steps in %.1f s" % (n_or, time.time() - t0), flush=True)
# the oracle stops an aircraft where the reference stops it (FC/sim.jl:561-570); the GPU must agree on the status WORD, on the step and
# the place of the termination, and on the frozen state — terminated aircraft are compared like the rest
tsg, twg = w.termination
mis = np.nonzero(stg[idx] != sto)[0]
print("status mismatches: %d; GPU bits %s, oracle bits %s" % (len(mis), stg[idx][mis][:10].tolist(), sto[mis][:10].tolist()))
print("termination step mismatches: %d; place mismatches: %d" % (int((tsg[idx] != tso).sum()), int((twg[idx] != two).sum())))
print("status histogram GPU (all): %s" % dict(zip(*np.unique(stg, return_counts=True))))
term = sto != 0
err = np.abs(xg[:, idx] - xo) / np.maximum(np.abs(xo), 1.0)
print("terminated: GPU %d of %d, oracle %d of %d (places %s)" % (int((stg != 0).sum()), n, int(term.sum()), n_or, dict(zip(*np.unique(two[term], return_counts=True)))))
print("max scaled |x_gpu - x_oracle| after 10000 steps: %.3e over the %d aircraft still flying (row %d), %.3e over the %d terminated ones; discrete states equal: %s"
      % (err[:, ~term].max(), int((~term).sum()), int(err[:, ~term].max(axis=1).argmax()), err[:, term].max() if term.any() else 0.0, int(term.sum()), np.array_equal(sg[:, idx], so)))
if kin != "NED":
    q = xg[12:16]; print("max | |q| - 1 | of the attitude quaternion = %.2e" % np.abs(np.sqrt((q ** 2).sum(0)) - 1).max())
print("non-finite states: %d" % int((~np.isfinite(xg)).sum()))
if term.any():
    et = err[:, term]
    r_, c_ = np.unravel_index(et.argmax(), et.shape)
    lane = np.nonzero(term)[0][c_]
    per = et.max(0)
    print("terminated aircraft: per-aircraft max error quantiles 50 / 90 / 100 %%: %.2e %.2e %.2e; the worst: row %d, ended at step %d, gpu %.9e oracle %.9e"
          % (*np.quantile(per, [0.5, 0.9, 1.0]), r_, int(tso[lane]), xg[r_, idx[lane]], xo[r_, lane]))
