"""Robot2D.Vehicle's continuous dynamics — oracle AND HIP path — against numbers the REFERENCE printed and stored.

(1) The reference's design notebook linearises `Robot2D.Vehicle` about rest and prints A (4 x 4), B, C, D to 16 digits
    (lib/FlightApps/design/robot2d/robot2d_design.ipynb cell 2; `linearize`, lib/FlightApps/src/robot2d/robot2d.jl:315-341:
    a finite-difference Jacobian of `f_ode!`'s ẋ and y = (ω, v, θ, η, u_m, τ_m) in x and u). Fixture:
    tests/golden/robot2d_linearization.json (data: the printed matrices). Here: the same Jacobian by Richardson-extrapolated
    central differences of the oracle's `r2_f_ode` and of `fb_f_ode` on a Robot2D handle, held to 1e-9 relative to each
    matrix's largest entry (measured: 1e-12, the noise floor of the difference quotient).
(2) The same notebook's cell 6 designs the velocity loop — LQR with integral action on the η-free model — and its result is
    the reference's robot2d.h5 (K_fbk, K_fwd, K_int; shipped byte-identical, hash-checked). Repeating that design on OUR
    Jacobian (scipy's CARE solver) must give the stored gains, and the printed open- and closed-loop poles."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import reference_fixtures as rf

FIX = json.load(open(os.path.join(rf.ROOT, "tests", "golden", "robot2d_linearization.json")))
A_REF, B_REF, C_REF, D_REF = (np.array(FIX[k]) for k in "ABCD")
DEFAULT_VP = np.array([0.15, 0.05, 1.0, 0.1, -1.0, -1.0, 0.32, 0.0189, 0.0014])   # Vehicle(), robot2d.jl:20-30
TOL = 1e-9
_D = C.POINTER(C.c_double)


def difference_points(h):
    """records [10, 20]: ±h and ±h/2 about x = 0, u_m = 0 in each of the five directions (ω, v, θ, η | u_m = record row 4)"""
    r = np.zeros((10, 20))
    for j in range(5):
        for k, d in enumerate((h, -h, h / 2, -h / 2)):
            r[j, 4 * j + k] = d
    return r


def jacobian(xd, y, h):
    """[ẋ; y] columns -> A, B, C, D by (4 D(h/2) − D(h)) / 3 (error O(h⁴))"""
    f = np.vstack([xd, y])
    J = np.zeros((f.shape[0], 5))
    for j in range(5):
        c = f[:, 4 * j:4 * j + 4]
        J[:, j] = (4.0 * (c[:, 2] - c[:, 3]) / h - (c[:, 0] - c[:, 1]) / (2 * h)) / 3.0
    return J[:4, :4], J[:4, 4:5], J[4:, :4], J[4:, 4:5]


def hold(name, got, want, log):
    scale = np.abs(want).max()
    err = np.abs(got - want).max() / scale
    log(f"{name}: max |ours − reference| / max|reference| = {err:.2e}")
    assert err <= TOL, f"{name} deviates by {err:.3e}"
    assert np.array_equal(got[want == 0.0] == 0.0, np.ones((want == 0.0).sum(), bool)) or np.abs(got[want == 0.0]).max() <= TOL * scale
    return err


def check_jacobian_and_design(A, B, Cm, Dm, log):
    for name, got, want in (("A", A, A_REF), ("B", B, B_REF), ("C", Cm, C_REF), ("D", Dm, D_REF)):
        hold(name, got, want, log)
    # cell 3: open-loop poles as printed (3 significant digits)
    poles = np.sort(np.linalg.eigvals(A).real)
    want = np.sort(np.array(FIX["open_loop_poles_3_digits"]))
    assert np.allclose(poles, want, rtol=5e-3, atol=1e-9), (poles, want)
    # cell 6: the velocity-loop design on the reduced (η-free) model
    scipy_linalg = pytest.importorskip("scipy.linalg")
    Ar, Br = A[:3, :3], B[:3]
    Cz, Dz = Cm[1:2, :3], Dm[1:2]                               # z = v
    A_aug = np.block([[Ar, np.zeros((3, 1))], [Cz, np.zeros((1, 1))]])
    B_aug = np.vstack([Br, Dz])
    q = FIX["velocity_loop_design"]["Q_diag"]
    Q = np.diag([q["ω"], q["v"], q["θ"], q["ξ_v"]]); R = np.array([[FIX["velocity_loop_design"]["R_diag"]["m"]]])
    P = scipy_linalg.solve_continuous_are(A_aug, B_aug, Q, R)
    K_aug = np.linalg.solve(R, B_aug.T @ P)
    M = np.linalg.inv(np.block([[Ar, Br], [Cz, Dz]]))
    K_fbk, K_int = K_aug[:, :3], K_aug[:, 3:]
    K_fwd = M[3:, 3:] + K_fbk @ M[:3, 3:]
    import sys
    sys.path.insert(0, os.path.join(rf.ROOT, "flight.jl_amd", "flightbatch"))
    import hdf5_min
    rf.assert_shipped_copy_is_the_references("flight.jl_amd/data/robot2d.h5")
    g = hdf5_min.read_all(os.path.join(rf.ROOT, "flight.jl_amd", "data", "robot2d.h5"))
    for name, got in (("K_fbk", K_fbk), ("K_fwd", K_fwd), ("K_int", K_int)):
        want = g[name].reshape(got.shape)
        err = np.abs(got - want).max() / np.abs(want).max()
        log(f"{name} redesigned from our Jacobian vs robot2d.h5: {err:.2e}")
        assert err <= 1e-10, (name, got, want)
    # cell 7: closed-loop poles of P_v as printed (the plant's η integrator stays at 0)
    Acl = A_aug - B_aug @ K_aug
    poles = np.sort(np.concatenate([[0.0], np.linalg.eigvals(Acl).real]))
    want = np.sort(np.array(FIX["velocity_loop_design"]["closed_loop_poles_3_digits"]))
    assert np.allclose(poles, want, rtol=5e-3, atol=1e-9), (poles, want)
    assert np.abs(np.linalg.eigvals(Acl).imag).max() < 1e-6        # dampreport prints damping ratio 1 for all of them


def test_oracle_robot2d_jacobian_is_the_references(oracle, capsys):
    h = 1e-3
    r = difference_points(h)
    xd = np.zeros((4, 20)); y = np.zeros((6, 20))
    oracle.lib.fo_robot2d_f_ode_y(C.c_int64(20), DEFAULT_VP.ctypes.data_as(_D), r.ctypes.data_as(_D), xd.ctypes.data_as(_D), y.ctypes.data_as(_D))
    with capsys.disabled():
        check_jacobian_and_design(*jacobian(xd, y, h), log=lambda s: print("\n[oracle] " + s, end=""))


@pytest.mark.gpu
def test_device_robot2d_jacobian_is_the_references(fb, capsys):
    """fb_f_ode on a Robot2D handle (k_r2_f_ode<double>) through the C ABI."""
    h = 1e-3
    w = fb.Robot2DWorld(20)
    w.set_state(difference_points(h))
    xd = np.zeros((4, 20))
    fb.f_ode(w, xd)
    y = w.y        # VehicleY: [ω, v, θ, η, u_m, τ_m, ω_dot, v_dot]
    assert np.array_equal(y[6:8], xd[0:2])
    with capsys.disabled():
        check_jacobian_and_design(*jacobian(xd, y[:6], h), log=lambda s: print("\n[HIP] " + s, end=""))
    w.close()
