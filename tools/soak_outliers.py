"""Follow-up of tools/soak_duo.py at full size: the aircraft of the 1,048,576-aircraft soak (10 000 steps, perturbed controls) whose final
attitude quaternion is off unit length, or whose status is not 0 / GroundCrash — every one of them against the oracle.
    python tools/soak_outliers.py [WA|ECEF|NED]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import flightbatch as fb
import bench
from oracle_binding import Oracle
kin = sys.argv[1] if len(sys.argv) > 1 else "WA"
nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
n = 1 << 20
EAS, h, psi, cell = bench.lattice(0)
w = fb.BatchedWorld(n, kinematics=kin)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
x0, s0, u0, ui0 = w.x, w.s, w.u.copy(), w.ui
rng = np.random.default_rng(3)
u0[fb.K["FB_U_ELEVATOR"]] += rng.uniform(-0.02, 0.02, n); u0[fb.K["FB_U_AILERON"]] += rng.uniform(-0.02, 0.02, n)
w.u = u0
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 100.0); w.sync()
xg, sg, stg = w.x, w.s, w.status
tsg, twg = w.termination
qn = np.sqrt((xg[12:16] ** 2).sum(0)) if kin != "NED" else np.ones(n)
odd = np.nonzero((np.abs(qn - 1) > 1e-6) | ((stg != 0) & (stg != 4)))[0]
print("%s: %d of %d aircraft with |q| off 1 by more than 1e-6 or a status other than 0 / GroundCrash" % (kin, len(odd), n))
if len(odd):
    idx = odd[:256]
    orc = Oracle()
    orc.lib.fo_set_kinematics(fb.K["FB_KIN_" + kin])
    xo27 = np.zeros((27, len(idx))); xo27[:12 + nk] = x0[:12 + nk, idx]; xo27[21:] = x0[12 + nk:, idx]
    xo27, so, sto, tso, two = orc.step_term(xo27, u0[:, idx], ui0[idx], s0[:, idx], orc.default_env(), 0.01, 10000)
    orc.lib.fo_set_kinematics(fb.K["FB_KIN_WA"])
    xo = np.vstack([xo27[:12 + nk], xo27[21:]])
    for j, i in enumerate(idx):
        err = np.abs(xg[:, i] - xo[:, j]) / np.maximum(np.abs(xo[:, j]), 1.0)
        qo = np.sqrt((xo[12:16, j] ** 2).sum())
        print("aircraft %7d: status gpu %d oracle %d; ended at step gpu %d oracle %d, place %d / %d; |q| gpu %.6g oracle %.6g; max scaled state difference %.3e (row %d)"
              % (i, stg[i], sto[j], tsg[i], tso[j], twg[i], two[j], qn[i], qo, err.max(), int(err.argmax())))
