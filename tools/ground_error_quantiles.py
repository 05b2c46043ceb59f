#!/usr/bin/env python3
"""Distribution (not only the maximum) of the GPU-vs-oracle error over aircraft in ground contact: descending approaches that touch down
and roll (the scenario of tests/test_gpu_parity.py::test_approach_crosses_the_air_ground_handover, 4096 aircraft, 8 s). Contact amplifies
rounding, so the maximum over a batch moves with any change of rounding in the contact code; the quantiles are what shows whether a change
of the arithmetic changed the accuracy. FLIGHTBATCH_LIB selects the library (A/B of two builds, e.g. the stepping kernels' own forms of the
contact branch against the reference's operations in their place: python __graft_entry__.py --variant gref -DFB_GROUND_REFERENCE_FORMS)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import flightbatch as fb  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
oracle = Oracle()
n = 4096
kin = sys.argv[1] if len(sys.argv) > 1 else "WA"     # python tools/ground_error_quantiles.py [WA|ECEF|NED]
nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
rng = np.random.default_rng(17)
h_trn = 300.0
tp = fb.TrimParameters(EAS=rng.uniform(33, 40, n), h_e=h_trn + rng.uniform(14, 40, n), γ_wb_n=-np.deg2rad(rng.uniform(2, 5, n)), flaps=1.0, ψ_nb=rng.uniform(-3, 3, n))
env = oracle.default_env(h_trn=h_trn)
w = fb.BatchedWorld(n, kinematics=kin)
w.set_params(h_terrain=h_trn)
fb.f_init(w, tp)
x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
ok = w.trim_success
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 8.0); w.sync()
xg, stg = w.x, w.status
oracle.lib.fo_set_kinematics(fb.K["FB_KIN_" + kin])     # (the oracle keeps 27 rows for every mechanisation, unused kinematic rows zero)
x27 = np.zeros((27, n)); x27[:12 + nk] = x0[:12 + nk]; x27[21:] = x0[12 + nk:]
xo27, so, sto = oracle.step(x27, u0, ui0, s0, env, 0.01, 800, threads=16)
_, yo, _ = oracle.f_ode(xo27, u0, ui0, so, env)
oracle.lib.fo_set_kinematics(fb.K["FB_KIN_WA"])
xo = np.vstack([xo27[:12 + nk], xo27[21:]])
agl = yo[fb.K["FB_Y_KIN"] + 21] - h_trn
live = ok & (sto == 0) & (stg == 0)
sc = np.maximum(np.abs(xo), 1.0)
err = np.abs(xg - xo) / sc
near = live & (agl < 2.5)
per = err[:, near].max(0)
q = np.quantile(per, [0.5, 0.9, 0.99, 0.999, 1.0])
print(kin, "| library:", os.environ.get("FLIGHTBATCH_LIB", "default"), "| status equal:", np.array_equal(stg[ok], sto[ok]), "| aircraft on or near the ground:", int(near.sum()))
print("per-aircraft max scaled error, quantiles 50 / 90 / 99 / 99.9 / 100 %%: %.2e %.2e %.2e %.2e %.2e" % tuple(q))
print("airborne lanes: max %.2e" % err[:, live & ~near].max())
lane = np.nonzero(near)[0][per.argmax()]
top = np.argsort(err[:, lane])[::-1][:6]
print("worst aircraft", lane, "rows", top, "errors", err[top, lane], "\n gpu", xg[top, lane], "\n oracle", xo[top, lane])
print(" its v_eb_b", xo[-3:, lane], "w_eb_b", xo[-6:-3, lane], "regulators", xo[2:8, lane], "agl", agl[lane], "wow L/R/N", yo[fb.K["FB_Y_LDG"] + 1, lane], yo[fb.K["FB_Y_LDG"] + 12, lane], yo[fb.K["FB_Y_LDG"] + 23, lane])
