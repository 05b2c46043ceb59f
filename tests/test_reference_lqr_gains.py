"""The Jacobian of the flight-dynamics model — oracle AND HIP path — pinned by the reference's own LQR gain tables.

The reference linearises Cessna172Xv0(NED) by finite differences at each of its 28 design points and stores the LQR gains it
designs on that linearisation: five lookups x 28 nodes of K_fbk (2 x 8 / 2 x 9), K_fwd, K_int (tests/reference_lqr.py has the
recipe, line-cited). K = R⁻¹ Bᵀ P(A, B, Q, R) depends on every entry of ∂f/∂x and ∂f/∂u of the aerodynamics, propulsion, engine,
atmosphere, rigid-body and kinematics models in the longitudinal / lateral subsystem; reproducing the stored matrices from OUR
f_ode! pins those derivatives against reference-held numbers. scipy's CARE solver stands in for ControlSystems.jl's `lqr`.

Tolerance: measured, round 5 (the log of every run prints the table): <= 6.6e-7 of each matrix's largest entry over all five
lookups and 28 nodes (te2te 4.9e-7, tv2te 6.6e-7, vh2te 1.9e-7, ar2ar 1.3e-8, φβ2ar 6.8e-9); the reference's side carries
FiniteDiff's forward-difference noise (step sqrt(eps) max(|x|, 1): ~1e-8 relative per Jacobian entry, amplified by the Riccati
solve's conditioning). Asserted at 5e-6."""
import ctypes as C

import numpy as np
import pytest

import reference_fixtures as rf
import reference_lqr as rl
from test_reference_trim_points import _trim_parameters_packed

TOL = 5e-6
pytest.importorskip("scipy.linalg")


def _oracle_ned(oracle):
    class _Scope:
        def __enter__(self):
            assert oracle.lib.fo_set_kinematics(2) == 0
        def __exit__(self, *a):
            oracle.lib.fo_set_kinematics(0)
    return _Scope()


def _trim_ned(oracle):
    tp = _trim_parameters_packed()
    env = oracle.default_env()                       # still air, ISA sea level: linearize()'s SimpleAtmosphere(wind = NoWind())
    ts0 = np.tile(np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])[:, None], (1, 28))
    r = oracle.trim(tp, ts0, env)
    assert r["ok"].all()
    return r, env


def test_oracle_jacobian_reproduces_the_references_lqr_gains(oracle, capsys):
    with _oracle_ned(oracle):
        r, env = _trim_ned(oracle)

        def f_ode(X, U):
            # columns are [point, aircraft] flattened point-major: ui / s of aircraft k repeat with period 28
            m = X.shape[1]
            xd, y, st = oracle.f_ode(X, U, np.tile(r["ui"], m // 28), np.tile(r["s"], (1, m // 28)), env)
            assert (st == 0).all()
            return xd, y
        A, B, Cy = rl.linearize(f_ode, r["x"], r["u"])
    with capsys.disabled():
        dev = rl.compare_all(A, B, Cy, log=lambda s_: print("\n[oracle] " + s_, end=""))
    worst = max(v for d in dev.values() for v in d.values())
    assert worst <= TOL, dev


@pytest.mark.gpu
def test_device_jacobian_reproduces_the_references_lqr_gains(fb, capsys):
    """the same through the C ABI: fb_trim + fb_f_ode of Cessna172Sv0(NED) on the device (24-row state: the NED block has 6 rows)"""
    EAS, h, flaps = rf.design_nodes()
    npts = 41
    n = 28 * npts
    w0 = fb.BatchedWorld(28, kinematics="NED")
    fb.f_init(w0, fb.TrimParameters(h_e=h, EAS=EAS, flaps=flaps))
    assert w0.trim_success.all()
    x24, u0, ui0, s0 = w0.x, w0.u, w0.ui, w0.s
    w0.close()
    x27 = np.zeros((27, 28)); x27[:18] = x24[:18]; x27[21:] = x24[18:]
    w = fb.BatchedWorld(n, kinematics="NED")

    def f_ode(X, U):
        assert X.shape[1] == n
        x = np.vstack([X[:18], X[21:]])
        w.set_state(x, np.tile(s0, (1, npts)))
        w.u = U; w.ui = np.tile(ui0, npts)
        xd = np.zeros((24, n)); fb.f_ode(w, xd)
        assert (w.status == 0).all()
        xd27 = np.zeros((27, n)); xd27[:18] = xd[:18]; xd27[21:] = xd[18:]
        return xd27, w.y
    A, B, Cy = rl.linearize(f_ode, x27, u0)
    w.close()
    with capsys.disabled():
        dev = rl.compare_all(A, B, Cy, log=lambda s_: print("\n[HIP] " + s_, end=""))
    worst = max(v for d in dev.values() for v in d.values())
    assert worst <= TOL, dev
