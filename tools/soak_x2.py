"""One-off long run of the Cessna172Xv2 stepper (control laws inside the stepping kernel) against the CPU oracle: 100 s of closed-loop
flight (10 000 RK4 steps, 5 000 control updates) for aircraft on randomised trims, every one in its own pair of control modes with its
own references — in both forms of the gain lookup (shared cell / per-lookup headers).   python tools/soak_x2.py [n=1024] [WA|ECEF|NED]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import flightbatch as fb
import bench
from oracle_binding import OracleX
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kin = sys.argv[2] if len(sys.argv) > 2 else "WA"
K = fb.K
DT = 0.01
rng = np.random.default_rng(41)
tp = fb.TrimParameters(h_e=rng.uniform(300.0, 2800.0, n), EAS=rng.uniform(38.0, 52.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n))
gains = fb.ctl_gains.ctl_gains_blob()
orc = bench._oracle()
X = OracleX(orc, gains)
env = orc.default_env()
perm = np.array([k if k < K["FB_X2_ACT"] else (27 + k - K["FB_X2_ACT"] if k < K["FB_X2_KIN"] else k - K["FB_NACT"]) for k in range(34)])
perm = np.array([r for r in perm if r not in {"WA": (), "ECEF": (20,), "NED": (18, 19, 20)}[kin]])   # C ABI row -> oracle / device row; the mechanisation's unused rows dropped
orc.lib.fo_set_kinematics(K["FB_KIN_" + kin])
modes_lon = rng.integers(0, 9, n); modes_lat = rng.integers(0, 5, n)
dref = dict(EAS=rng.uniform(-3, 3, n), CLM=rng.uniform(-1.5, 1.5, n), PHI=rng.uniform(-0.3, 0.3, n), CHI=rng.uniform(-0.5, 0.5, n))
ref = None
for same_grid in ("1", "0"):
    os.environ["FLIGHTBATCH_CTL_SAME_GRID"] = same_grid
    w = fb.Cessna172Xv2World(n, gains=gains, kinematics=kin)
    sim = fb.Simulation(w, dt=DT, Δt=2 * DT, save_on=False, steps_per_launch=50)
    fb.init(sim, tp)
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = modes_lon; cu[K["FB_CU_LAT_MODE_REQ"]] = modes_lat
    cu[K["FB_CU_EAS_REF"]] += dref["EAS"]; cu[K["FB_CU_CLM_REF"]] += dref["CLM"]; cu[K["FB_CU_PHI_REF"]] += dref["PHI"]; cu[K["FB_CU_CHI_REF"]] += dref["CHI"]
    w.cu = cu
    if ref is None:
        o = X.trim_init(tp.pack(n), fb.TrimState(n), env, 2 * DT)
        o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
        o["cu"] = np.ascontiguousarray(o["cu"]); o["cu"][:] = cu
        o["x"][:] = 0; o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
        t0 = time.time(); X.step_term(o, env, DT, 2, 10000, threads=min(orc.max_threads(), bench.usable_cores()))
        print("oracle: %d aircraft x 10000 closed-loop steps in %.1f s" % (n, time.time() - t0), flush=True)
        ref = o
    t0 = time.time(); fb.step(sim, 100.0); w.sync()
    print("GPU (%s, same_grid=%s): %d aircraft x 10000 steps in %.2f s" % (kin, same_grid, n, time.time() - t0), flush=True)
    term = ref["status"] != 0      # terminated aircraft are compared like the rest: frozen where the reference stops (FC/sim.jl:561-570)
    tsg, twg = w.termination
    sc = np.ones_like(ref["x"]); sc[:27] = bench.state_floor(ref["x"][:27])
    err = np.abs(w.x - ref["x"][perm]) / sc[perm]
    cerr = np.abs(w.cs - ref["cs"]) / np.maximum(np.abs(ref["cs"]), 1.0)
    print("  status mismatches: %d (terminated: GPU %d, oracle %d of %d); termination step / place mismatches: %d / %d; max scaled |x_gpu - x_oracle| after 10000 steps: "
          "%.3e over %d flying (row %d), %.3e over the terminated; control-law record: %.3e; modes equal: %s"
          % (int((w.status != ref["status"]).sum()), int((w.status != 0).sum()), int(term.sum()), n, int((tsg != ref["term_step"]).sum()), int((twg != ref["term_where"]).sum()),
             err[:, ~term].max(), int((~term).sum()), int(err[:, ~term].max(axis=1).argmax()), err[:, term].max() if term.any() else 0.0, cerr.max(),
             np.array_equal(w.cs[K["FB_CS_LON_MODE"]], ref["cs"][K["FB_CS_LON_MODE"]])), flush=True)
    w.close()
orc.lib.fo_set_kinematics(K["FB_KIN_WA"])
