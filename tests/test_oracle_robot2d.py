"""Pins the Robot2D oracle against the reference's own tests: lib/FlightApps/test/robot2d/test_robot2d.jl:19-64 (vehicle)
and :70-102 (closed loop with the LQR/PID controller and the gains of robot2d.h5)."""
import ctypes as C
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd", "flightbatch"))
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int32)


def dp(a):
    return a.ctypes.data_as(_D)


DEFAULT_VP = np.array([0.15, 0.05, 1.0, 0.1, -1.0, -1.0, 0.32, 0.0189, 0.0014])


def gains_from_h5():
    import hdf5_min
    d = hdf5_min.read_all(os.path.join(ROOT, "flight.jl_amd", "data", "robot2d.h5"))
    return np.concatenate([d["K_fbk"].ravel(), d["K_fwd"].ravel(), d["K_int"].ravel(), d["x_trim"].ravel(), d["u_trim"].ravel(),
                           d["z_trim"].ravel(), [0.6, 0.0, 0.0, 0.01]]).astype(np.float64)


def run(L, vp, gp, r, u, dt, ratio, ctl, step0, nsteps):
    st = np.zeros(r.shape[1], np.int32)
    L.fo_robot2d_step(C.c_int64(r.shape[1]), dp(vp), dp(gp), C.c_double(dt), ratio, ctl, dp(u), dp(r), C.c_int64(step0), C.c_int64(nsteps),
                      st.ctypes.data_as(_I))
    return st


def test_hdf5_gains_file():
    g = gains_from_h5()
    assert g.shape == (14,) and np.all(np.isfinite(g))
    assert g[0] < 0 and g[1] < 0 and g[2] < 0 and g[3] < 0 and g[4] < 0   # K_fbk, K_fwd, K_int of the balancing design


def test_vehicle_open_loop(oracle):
    """test_robot2d.jl:19-64 — 20 s at dt = 0.01, no controller."""
    L = oracle.lib
    vp = DEFAULT_VP.copy(); gp = gains_from_h5(); u = np.zeros((4, 1))
    def sim(u_m=0.0, w=0.0, eta=0.0):
        r = np.zeros((10, 1))
        L.fo_robot2d_init(C.c_int64(1), dp(vp), dp(np.array([[u_m], [w], [eta]], dtype=np.float64)), dp(r))
        run(L, vp, gp, r, u, 0.01, 1, 0, 0, 2000)
        return r[:4, 0]
    x = sim()
    assert np.all(np.abs(x) < 1e-3)
    x = sim(u_m=0.7)
    assert np.isclose(x[1], 0.32 * 0.7 * 0.05 / 0.0189, rtol=1.5e-8) and abs(x[0]) < 1e-3 and abs(x[2]) < 1e-3 and x[3] > 0
    x = sim(w=1e-3)
    assert abs(x[0]) < 1e-3 and abs(x[1]) < 1e-3 and abs(x[2] - np.pi) < 1e-3 and x[3] > 0
    x = sim(w=-1e-3)
    assert abs(x[0]) < 1e-3 and abs(x[1]) < 1e-3 and abs(x[2] + np.pi) < 1e-3 and x[3] < 0


def test_controller_closed_loop(oracle):
    """test_robot2d.jl:70-102 — Robot(vehicle = Vehicle(L = 0.1, R = 0.08, m_b = 0.5)), dt = 0.01 (Δt = dt)."""
    L = oracle.lib
    vp = DEFAULT_VP.copy(); vp[0] = 0.1; vp[1] = 0.08; vp[2] = 0.5
    gp = gains_from_h5()
    r = np.zeros((10, 1))
    L.fo_robot2d_init(C.c_int64(1), dp(vp), dp(np.zeros((3, 1))), dp(r))
    u = np.zeros((4, 1)); step = 0
    u[:, 0] = [0, 0.1, 0, 0]                       # mode_m, m_ref = 0.1
    st = run(L, vp, gp, r, u, 0.01, 1, 1, step, 10); step += 10
    assert r[4, 0] == 0.1 and r[2, 0] < 0 and st[0] == 0   # u_m == m_ref; tilting backward
    u[:, 0] = [1, 0.1, 0.3, 0]                     # mode_v, v_ref = 0.3
    st = run(L, vp, gp, r, u, 0.01, 1, 1, step, 1000); step += 1000
    assert abs(r[1, 0] - 0.3) < 1e-3 and st[0] == 0
    u[2, 0] = -np.inf
    st = run(L, vp, gp, r, u, 0.01, 1, 1, step, 1000); step += 1000
    v_lim = 0.4 * 0.32 * 0.08 / 0.0189
    assert abs(r[1, 0] + v_lim) < 1e-3 and st[0] == 0
    u[:, 0] = [2, 0.1, -np.inf, 1.0]               # mode_η, η_ref = 1
    st = run(L, vp, gp, r, u, 0.01, 1, 1, step, 2000); step += 2000
    assert abs(r[3, 0] - 1.0) < 1e-3 and st[0] == 0


def test_lost_balance_flag(oracle):
    """robot2d.jl:553-561 — |θ| > 45° terminates (LostBalance) -> sticky status bit 32, state frozen."""
    L = oracle.lib
    vp = DEFAULT_VP.copy(); gp = gains_from_h5()
    r = np.zeros((10, 1)); r[2, 0] = 0.7   # 40°: falls over in open loop with the motor off
    u = np.zeros((4, 1))
    st = run(L, vp, gp, r, u, 0.01, 2, 1, 0, 500)
    assert st[0] == 32 and abs(r[2, 0]) > np.pi / 4
