"""Robot2D on the GPU (through the C ABI) against the CPU oracle and against the reference's closed-loop tests."""
import ctypes as C
import numpy as np
import pytest

from test_oracle_robot2d import DEFAULT_VP, gains_from_h5, run

pytestmark = pytest.mark.gpu
_D = C.POINTER(C.c_double)


def oracle_init(L, vp, ip):
    r = np.zeros((10, ip.shape[1]))
    L.fo_robot2d_init(C.c_int64(ip.shape[1]), vp.ctypes.data_as(_D), np.ascontiguousarray(ip).ctypes.data_as(_D), r.ctypes.data_as(_D))
    return r


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-11), ("f32", 2e-3)])
def test_robot2d_closed_loop_matches_oracle(fb, oracle, dtype, tol):
    """README example 1 configuration (dt = 0.01, Δt = 0.02), mixed modes and references across the batch."""
    n = 4096
    rng = np.random.default_rng(5)
    w = fb.Robot2DWorld(n, dtype=dtype)
    ipar = fb.InitParameters(u_m=rng.uniform(-0.2, 0.2, n), ω=rng.uniform(-0.05, 0.05, n), η=rng.uniform(-1, 1, n))
    fb.f_init(w, ipar)
    u = np.zeros((4, n))
    u[0] = rng.integers(0, 3, n); u[1] = rng.uniform(-0.3, 0.3, n); u[2] = rng.uniform(-0.5, 0.5, n); u[3] = rng.uniform(-2, 2, n)
    w.u = u
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=37)   # launch size not a multiple of Δt/dt on purpose
    fb.step(sim, 5.0); w.sync()
    vp = DEFAULT_VP.copy(); gp = gains_from_h5()
    r = oracle_init(oracle.lib, vp, ipar.pack(n))
    st = run(oracle.lib, vp, gp, r, u, 0.01, 2, 1, 0, 500)
    if dtype == "f64":
        assert np.array_equal(w.status != 0, st != 0)
    ok = (st == 0) & (w.status == 0)
    assert ok.mean() > 0.5
    err = np.max(np.abs(w.x[:, ok] - r[:, ok]) / np.maximum(np.abs(r[:, ok]), 1.0))
    print(dtype, "max scaled error after 500 steps:", err, "terminated:", int((st != 0).sum()))
    assert err < tol
    xd = np.zeros((4, n)); fb.f_ode(w, xd)
    xdo = np.zeros((4, n))
    oracle.lib.fo_robot2d_f_ode(C.c_int64(n), vp.ctypes.data_as(_D), np.ascontiguousarray(w.x).ctypes.data_as(_D), xdo.ctypes.data_as(_D))
    assert np.max(np.abs(xd - xdo)[:, ok] / np.maximum(np.abs(xdo[:, ok]), 1.0)) < (1e-12 if dtype == "f64" else 1e-4)
    y = w.y
    assert np.allclose(y[:4], w.x[:4]) and np.allclose(y[6:8], xd[:2])
    w.close()


def test_robot2d_reference_closed_loop_scenario(fb):
    """lib/FlightApps/test/robot2d/test_robot2d.jl:70-97 on the GPU: Vehicle(L = 0.1, R = 0.08, m_b = 0.5), dt = Δt = 0.01."""
    w = fb.Robot2DWorld(256, vehicle=dict(L=0.1, R=0.08, m_b=0.5))
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=100)
    fb.init(sim, fb.InitParameters())
    u = np.zeros((4, 256)); u[0] = 0; u[1] = 0.1
    w.u = u; fb.step(sim, 0.1); w.sync()
    x = w.x
    assert np.all(x[4] == 0.1) and np.all(x[2] < 0)
    u[0] = 1; u[2] = 0.3; w.u = u; fb.step(sim, 10.0); w.sync()
    assert np.all(np.abs(w.x[1] - 0.3) < 1e-3)
    u[2] = -np.inf; w.u = u; fb.step(sim, 10.0); w.sync()
    assert np.all(np.abs(w.x[1] + 0.4 * 0.32 * 0.08 / 0.0189) < 1e-3)
    u[0] = 2; u[3] = 1.0; w.u = u; fb.step(sim, 20.0); w.sync()
    assert np.all(np.abs(w.x[3] - 1.0) < 1e-3) and (w.status == 0).all()
    w.close()


def test_saved_sample_is_taken_after_the_user_callback(fb):
    """CallbackSet order cb_step, cb_periodic, cb_user, cb_save (FC/sim.jl:204-218): what the user callback changes is in the sample
    saved at that instant. And the constructor says so — up front, with the ways out — when the default device log cannot hold the run
    (it is then capped; tests/test_gpu_docs.py steps such a Simulation)."""
    n = 64
    w = fb.Robot2DWorld(n)

    def cb(mdl):
        x = mdl.x; x[3] = mdl.t; mdl.x = x          # overwrite the position state with the clock

    sim = fb.Simulation(w, dt=0.01, Δt=0.02, t_end=0.5, saveat=0.1, user_callback=cb)
    fb.init(sim, fb.InitParameters())
    fb.run(sim)
    ts = fb.TimeSeries(sim)
    assert len(ts) == 6 and np.allclose(ts.t, 0.1 * np.arange(6))
    assert np.array_equal(ts.x[1:, 3, :], np.repeat((0.01 * np.arange(10, 51, 10))[:, None], n, axis=1))
    assert np.array_equal(ts.x[-1], w.x)
    w.close()
    big = fb.Robot2DWorld(1 << 20)
    with pytest.warns(UserWarning, match="log"):
        fb.Simulation(big, dt=0.01)                  # the reference's defaults: save every step until t = 10000
    fb.Simulation(big, dt=0.01, save_on=False)
    big.close()


def test_robot2d_lost_balance_freezes(fb):
    w = fb.Robot2DWorld(128)
    fb.f_init(w, fb.InitParameters())
    x = w.x; x[2] = 0.7; w.x = x
    u = np.zeros((4, 128)); w.u = u     # mode_m with zero command: falls over
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.step(sim, 5.0); w.sync()
    assert (w.status == fb.K["FB_ST_LOST_BALANCE"]).all() and np.all(np.abs(w.x[2]) > np.pi / 4)
    xa = w.x; fb.step(sim, 1.0); w.sync()
    assert np.array_equal(xa, w.x)   # terminated robots stay frozen
    w.close()


def test_robot2d_device_log_f32(fb):
    """The device log of an fp32 batch: samples are stored in fp32 on the GPU and widened on read."""
    n = 512
    w = fb.Robot2DWorld(n, dtype="f32")
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, t_end=1.0, saveat=0.1, save_rows=[0, 1, 6])
    fb.init(sim, fb.InitParameters(u_m=0.05))
    fb.run(sim)
    ts = fb.TimeSeries(sim)
    assert len(ts) == 11 and ts.x.shape == (11, 10, n) and ts.y.shape == (11, 3, n)
    assert np.array_equal(ts.x[-1], w.x)
    assert np.array_equal(ts.y[:, 0], ts.x[:, 0]) and np.array_equal(ts.y[:, 1], ts.x[:, 1])   # y.ω, y.v are the states
    assert np.all(np.abs(np.diff(ts.x[:, 1, 0])) > 0)   # it moves
    w.close()


def test_mixed_fleet_c172_and_robot2d(fb, oracle):
    """Config 5 (BASELINE.json configs[4]) at test size: 50 % Cessna172Sv0 / 50 % Robot2D interleaved in the input order,
    dt = 0.01, Δt = 0.02; the packed run must equal the two homogeneous runs, vehicle by vehicle."""
    n = 4096
    types = np.where(np.arange(n) % 2 == 0, fb.K["FB_MODEL_C172S0"], fb.K["FB_MODEL_ROBOT2D"])
    types[:64] = fb.K["FB_MODEL_ROBOT2D"]     # not a pure alternation
    fleet = fb.MixedFleet(types, {fb.K["FB_MODEL_C172S0"]: lambda m: fb.BatchedWorld(m), fb.K["FB_MODEL_ROBOT2D"]: lambda m: fb.Robot2DWorld(m, dtype="f32")})
    fleet.simulate(dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    nc, nr = fleet.index[fb.K["FB_MODEL_C172S0"]].size, fleet.index[fb.K["FB_MODEL_ROBOT2D"]].size
    assert nc + nr == n and nr == n // 2 + 32
    EAS = 40 + 10 * (fleet.index[fb.K["FB_MODEL_C172S0"]] / n)            # a property of the vehicle, not of its slot
    fleet.init({fb.K["FB_MODEL_C172S0"]: fb.TrimParameters(EAS=EAS, h_e=1000.0), fb.K["FB_MODEL_ROBOT2D"]: fb.InitParameters(u_m=0.0)})
    rw = fleet.worlds[fb.K["FB_MODEL_ROBOT2D"]]
    u = rw.u; u[0] = 1; u[2] = 0.3; rw.u = u                               # mode_v, v_ref = 0.3
    fleet.step(2.0); fleet.sync()
    x = fleet.gather("x")
    st = fleet.gather("status", fill=-1)
    assert x.shape == (27, n) and (st == 0).all()
    is_r = types == fb.K["FB_MODEL_ROBOT2D"]
    assert np.isnan(x[10:, is_r]).all() and np.isfinite(x[:, ~is_r]).all()
    assert np.all(np.abs(x[1, is_r] - 0.3) < 0.05)                         # robots approach v_ref
    # homogeneous reference runs
    w = fb.BatchedWorld(nc); sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters(EAS=EAS, h_e=1000.0)); fb.step(sim, 2.0); w.sync()
    assert np.array_equal(w.x, x[:, ~is_r])
    w.close(); fleet.close()


def test_user_callback_runs_after_every_step(fb):
    """Simulation(mdl; user_callback!) (FC/sim.jl:190, 331-341): the callback sees the model after each step and may change its
    inputs — here a velocity reference that follows sin(t), like the reference's own use in FPt/test_control.jl:221-232."""
    n = 64
    w = fb.Robot2DWorld(n)
    calls = []

    def cb(mdl):
        calls.append(mdl.t)
        u = mdl.u; u[0] = 1; u[2] = 0.2 * np.sin(mdl.t); mdl.u = u

    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, user_callback=cb)
    fb.init(sim, fb.InitParameters())
    fb.step(sim, 3.0); w.sync()
    assert len(calls) == 300 and np.allclose(calls, 0.01 * np.arange(1, 301))
    v = w.x[1]
    assert np.all((np.abs(v) > 0.02) & (np.abs(v) < 0.25)) and (w.status == 0).all()   # driven by the moving reference (with the loop's lag)
    w.close()
