"""The conditioning of a long ground roll, measured on the CPU oracle alone (tests/conditioning.py is what the GPU tests hold the HIP
path to): autopilot descents of Cessna172Xv2 onto a runway, the oracle against itself with v_eb_b nudged once at touchdown. What this
pins WITHOUT a GPU: (1) the claim the GPU tests' tolerance rests on — rounding-level differences at touchdown are amplified by the
friction regulators (landinggear.jl:411-476) by many orders of magnitude on SOME aircraft while most stay at rounding level — as
numbers in the log; (2) that the envelope check accepts a further oracle run (a stand-in for a correct GPU path) and refuses a run
that carries a real defect (one state moved by 1e-5 relative at touchdown)."""
import numpy as np
import pytest

import conditioning
from oracle_binding import OracleX, header_enums
from test_oracle_c172x import default_trim_params, default_trim_state

K = header_enums()


@pytest.fixture(scope="module")
def gains():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flight.jl_amd", "flightbatch"))
    import ctl_gains
    return ctl_gains.ctl_gains_blob()


def scaled(x):
    sc = np.ones_like(x)
    sc[:27] = np.maximum(np.abs(x[:27]), 1.0)
    sc[16:20] = 1.0
    return sc


def test_oracle_against_itself_on_long_ground_rolls(oracle, gains):
    n = 384
    rng = np.random.default_rng(61)
    n_e = np.array([1.0, 0.0, 0.0])
    N0 = oracle.lib.fo_geoid_height(n_e.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
    tp = default_trim_params(n, FB_TP_EAS=rng.uniform(42.0, 55.0, n), FB_TP_H_E=N0 + 2.0 + rng.uniform(5.0, 20.0, n),
                             FB_TP_PSI_NB=rng.uniform(-np.pi, np.pi, n), FB_TP_GAMMA_WB_N=-0.05)
    X = OracleX(oracle, gains)
    env = oracle.default_env()
    o = X.trim_init(tp, default_trim_state(n), env, 0.02, threads=8)
    assert o["ok"].all()
    o["cu"][K["FB_CU_LON_MODE_REQ"]] = float(K["FB_LON_EAS_CLM"])
    o["cu"][K["FB_CU_LAT_MODE_REQ"]] = float(K["FB_LAT_PHI_BETA"])
    o["cu"][K["FB_CU_CLM_REF"]] = -rng.uniform(2.0, 3.5, n)          # gentle enough to survive the touchdown and roll
    start = {k: np.array(v, copy=True) for k, v in o.items() if isinstance(v, np.ndarray)}
    nsteps = 1000
    nom = {k: np.array(v, copy=True) for k, v in start.items()}
    nom["status"] = np.zeros(n, np.int32); nom["nstep"] = 0
    X.step_term(nom, env, 0.01, 2, nsteps, threads=8)
    rolling = (nom["status"] == 0) & (nom["x"][20] - N0 < 3.0)      # on their wheels at the end
    assert rolling.sum() >= 100, int(rolling.sum())

    def lane_err(p):
        return np.maximum((np.abs(p["x"] - nom["x"]) / scaled(nom["x"])).max(0),
                          (np.abs(p["cs"] - nom["cs"]) / np.maximum(np.abs(nom["cs"]), 1.0)).max(0))[rolling]
    ulp = conditioning.x2_perturbed_runs(X, start, env, nsteps, 20, N0, None, K=2, seed=1, threads=8)
    rel = conditioning.x2_perturbed_runs(X, start, env, nsteps, 20, N0, 1e-12, K=5, jitter=conditioning.ULP_R, seed=2, threads=8)
    assert all(p["nudged"][rolling].all() for p in ulp + rel)
    E_ulp = np.stack([lane_err(p) for p in ulp]); E = np.stack([lane_err(p) for p in rel])
    q = [0.5, 0.9, 0.99, 1.0]
    print(f"{int(rolling.sum())} aircraft rolling after {nsteps} steps; oracle vs oracle' per-aircraft error quantiles 50/90/99/100 %: "
          f"one ulp on v_eb_b at touchdown {np.quantile(E_ulp.ravel(), q)}; 1e-12 relative + one ulp of the geocentric radius on h_e per step {np.quantile(E.ravel(), q)}")
    # (1) one ulp (1.1e-16 relative) at touchdown does not stay one ulp: the median aircraft is still at rounding level, the worst is
    # amplified by at least six orders of magnitude
    assert np.median(E_ulp) < 1e-9 and E_ulp.max() > 1e-10
    # (2) the check: a fifth run of the same kind passes as "the GPU" against the other four ...
    conditioning.check_against_envelope(E[4], E[:4], "oracle run 5 against runs 1-4")
    # ... and a run with a defect (1e-5 relative on v_eb_b at touchdown: seven orders above rounding) does not
    bad = conditioning.x2_perturbed_runs(X, start, env, nsteps, 20, N0, 1e-5, K=1, jitter=conditioning.ULP_R, seed=9, threads=8)
    with pytest.raises(AssertionError):
        conditioning.check_against_envelope(lane_err(bad[0]), E[:4], "a defective run")
