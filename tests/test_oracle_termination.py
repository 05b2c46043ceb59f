"""The oracle's termination semantics (oracle/fo_c172.hpp `c172_step`, fo_robot2d.hpp `r2_step`) pinned on the CPU: the reference
stops a simulation at the first exception (lib/FlightCore/src/sim.jl:561-570) and leaves mdl.x as it stands at the throw —
f_ode_wrapper! copies the integrator's argument into mdl.x before it calls f_ode! (sim.jl:306), so for an f_ode! that threw at RK
stage k_j that is the stage's ARGUMENT x_n + c_j dt k_{j-1}; cb_step_affect! (sim.jl:318-328) throws out of f_step! with x_{n+1}
partly updated. These tests rebuild what the frozen state must be from single f_ode! / step calls of the same oracle (no second
implementation: they pin the BOOKKEEPING — which state, which step, which place, which status bit — not the physics)."""
import ctypes as C

import numpy as np
import pytest

_D = C.POINTER(C.c_double)


def _qmul(a, b):
    return np.stack([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                     a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])


def _trimmed(fb, oracle, n, seed):
    rng = np.random.default_rng(seed)
    tp = fb.TrimParameters(EAS=rng.uniform(40.0, 55.0, n), h_e=1000.0, ψ_nb=rng.uniform(-np.pi, np.pi, n))
    r = oracle.trim(tp.pack(n), fb.TrimState(n), oracle.default_env())
    assert r["ok"].all()
    return r["x"], r["s"], r["u"], r["ui"], rng


def _pitch(x, climb):
    V = np.sqrt(x[24] ** 2 + x[25] ** 2 + x[26] ** 2)
    d = np.arcsin(np.clip(climb / V, -0.9, 0.9))
    z = np.zeros_like(d)
    x[12:16] = _qmul(x[12:16], np.stack([np.cos(d / 2), z, np.sin(d / 2), z]))


def test_f_ode_throw_leaves_the_stage_argument(fb, oracle):
    K = fb.K
    n = 256
    x, s, u, ui, rng = _trimmed(fb, oracle, n, 5)
    a = 6378137.0
    ceiling = 84852.0 * a / (a - 84852.0)                        # h_orth whose geopotential altitude is the last ISA ceiling
    geoid = oracle.lib.fo_geoid_height
    n_e = np.array([1.0, 0.0, 0.0])
    x[16:20] = np.array([np.sqrt(0.5), 0.0, -np.sqrt(0.5), 0.0])[:, None]          # q_ew of lat = lon = 0 (Ry(-π/2))
    x[20] = ceiling + geoid(n_e.ctypes.data_as(_D)) - rng.uniform(0.2, 6.0, n)
    _pitch(x, rng.uniform(10.0, 30.0, n))
    env = oracle.default_env()
    dt = 0.01
    xo, so, st, tstep, twhere = oracle.step_term(x, u, ui, s, env, dt, 80)
    term = st != 0
    assert term.sum() > 100 and (st[term] == K["FB_ST_ISA_RANGE"]).all()
    assert set(np.unique(twhere[term])) <= {K["FB_TERM_F_ODE_K2"], K["FB_TERM_F_ODE_K3"], K["FB_TERM_F_ODE_K4"], K["FB_TERM_F_ODE_NEW"]}
    assert (tstep[~term] == -1).all() and (twhere[~term] == K["FB_TERM_NONE"]).all()
    assert np.array_equal(so, s)                                                   # no f_step! ran at the throw
    # rebuild each terminated aircraft's frozen state: x_n from a run of exactly `step` steps (which must NOT terminate), then the
    # stage arguments from single f_ode! calls, in OrdinaryDiffEq's RK4 order
    for k in np.nonzero(term)[0][:48]:
        col = slice(k, k + 1)
        xs, us, uis, ss = x[:, col].copy(), u[:, col].copy(), ui[col].copy(), s[:, col].copy()
        m = int(tstep[k]) - (1 if twhere[k] == K["FB_TERM_F_ODE_NEW"] else 0)       # full steps before the one that threw
        xn, sn, stn = oracle.step(xs, us, uis, ss, env, dt, m)
        assert stn[0] == 0
        def f(xx):
            xd, _, stf = oracle.f_ode(np.ascontiguousarray(xx), us, uis, sn, env)
            return xd, int(stf[0])
        k1, b1 = f(xn); assert b1 == 0
        a2 = xn + (dt / 2) * k1
        k2, b2 = f(a2)
        if twhere[k] == K["FB_TERM_F_ODE_K2"]:
            assert b2 == K["FB_ST_ISA_RANGE"] and np.array_equal(xo[:, col], a2); continue
        assert b2 == 0
        a3 = xn + (dt / 2) * k2
        k3, b3 = f(a3)
        if twhere[k] == K["FB_TERM_F_ODE_K3"]:
            assert b3 == K["FB_ST_ISA_RANGE"] and np.array_equal(xo[:, col], a3); continue
        assert b3 == 0
        a4 = xn + dt * k3
        k4, b4 = f(a4)
        if twhere[k] == K["FB_TERM_F_ODE_K4"]:
            assert b4 == K["FB_ST_ISA_RANGE"] and np.array_equal(xo[:, col], a4); continue
        assert b4 == 0
        xn1 = xn + (dt / 6) * (2 * (k2 + k3) + (k1 + k4))
        _, b5 = f(xn1)
        assert b5 == K["FB_ST_ISA_RANGE"] and np.array_equal(xo[:, col], xn1)      # thrown at the new state, ahead of the callbacks
    # a terminated aircraft is left alone by a later call, and the others go on
    x2, s2, st2, ts2, tw2 = oracle.step_term(xo, u, ui, so, env, dt, 5, step0=80, status=st)
    assert np.array_equal(x2[:, term], xo[:, term]) and np.array_equal(st2[term], st[term])
    assert not np.array_equal(x2[:, ~term], xo[:, ~term]) or (~term).sum() == 0


def test_crash_in_f_step_leaves_the_part_of_f_step_that_ran(fb, oracle):
    K = fb.K
    n = 512
    rng = np.random.default_rng(11)
    x = np.zeros((27, n))
    x[8] = 0.5
    z = np.zeros(n)
    th = rng.uniform(-0.02, 0.08, n)
    x[12:16] = np.stack([np.cos(th / 2), z, np.sin(th / 2), z]) * 1.00000003        # off-unit by 3e-8: f_step! renormalises (kinematics.jl:114-118)
    lat, lon = 0.7, -0.3
    aa = -(lat + np.pi / 2)
    x[16:20] = _qmul(np.array([np.cos(lon / 2), 0, 0, np.sin(lon / 2)])[:, None] * np.ones(n), np.array([np.cos(aa / 2), 0, np.sin(aa / 2), 0])[:, None] * np.ones(n))
    n_e = np.array([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    x[20] = oracle.lib.fo_geoid_height(n_e.ctypes.data_as(_D)) + rng.uniform(2.2, 3.5, n)
    x[24] = rng.uniform(25, 40, n); x[26] = rng.uniform(11.0, 15.0, n)              # sink rates beyond the dampers' 10 m/s limit (landinggear.jl:341-344)
    x[2:8] = rng.normal(0, 0.2, (6, n))                                             # non-zero friction regulators
    x[9] = 100.0
    s = np.zeros((2, n), np.int32); s[1] = 2
    u = np.zeros((16, n)); u[11:16] = np.array([75, 75, 0, 0, 50.0])[:, None]; u[0] = 0.2; u[1] = 0.5
    ui = np.full(n, 4 | 8 | 2, np.int32)                                            # engine stop requested: the state machine would act on it
    env = oracle.default_env()
    xo, so, st, tstep, twhere = oracle.step_term(x, u, ui, s, env, 0.01, 40)
    term = st != 0
    assert term.sum() > 300 and (st[term] == K["FB_ST_GROUND_CRASH"]).all() and (twhere[term] == K["FB_TERM_F_STEP"]).all()
    # the quaternions have been through f_step!'s renormalisation (kinematics first, aircraftbase.jl:178): the 3e-8 they started with is
    # gone, what is left is within the 1e-8 that normalize_block! tolerates (kinematics.jl:114-118)
    qn = np.sqrt((xo[12:16, term] ** 2).sum(0))
    assert np.abs(qn - 1.0).max() <= 1e-8 * (1 + 1e-6)
    # ... the engine's state machine (last in c172.jl:715-724) does not run in an f_step! that throws: f_step! alone on the frozen states
    # (the struts throw again), engine "running" and a stop requested -> still running; on a state that does not throw -> stopped
    sel = np.nonzero(term)[0]
    s_run = np.zeros((2, sel.size), np.int32); s_run[1] = 2
    x_f, s_f, st_f = oracle.f_step(xo[:, sel], u[:, sel], ui[sel], s_run, env)
    assert (st_f & K["FB_ST_GROUND_CRASH"]).all() and (s_f[1] == 2).all()
    x_g, s_g, st_g = oracle.f_step(x[:, sel], u[:, sel], ui[sel], s_run, env)          # the initial states: in the air
    assert (st_g == 0).all() and (s_g[1] == 0).all()
    # the step before the crash is a plain step: re-running `step - 1` steps terminates nobody and one more step crashes
    k = int(np.nonzero(term)[0][0])
    col = slice(k, k + 1)
    xs, ss, stn = oracle.step(x[:, col].copy(), u[:, col].copy(), ui[col].copy(), s[:, col].copy(), env, 0.01, int(tstep[k]) - 1)
    assert stn[0] == 0
    x1, s1, st1, ts1, tw1 = oracle.step_term(xs, u[:, col].copy(), ui[col].copy(), ss, env, 0.01, 1, step0=int(tstep[k]) - 1)
    assert st1[0] == K["FB_ST_GROUND_CRASH"] and ts1[0] == tstep[k] and np.array_equal(x1, xo[:, col])


def test_robot2d_lost_balance_stops_after_the_rk_update(fb, oracle):
    from test_oracle_robot2d import DEFAULT_VP, gains_from_h5
    n = 64
    vp = DEFAULT_VP.copy(); gp = gains_from_h5()
    r = np.zeros((10, n)); r[2] = np.linspace(0.5, 0.78, n)                         # tilted, motor command 0: falls
    u = np.zeros((4, n))
    st = np.zeros(n, np.int32); ts = np.full(n, -1, np.int64)
    r0 = r.copy()
    oracle.lib.fo_robot2d_step_term(C.c_int64(n), vp.ctypes.data_as(_D), gp.ctypes.data_as(_D), C.c_double(0.01), C.c_int32(2), C.c_int32(1),
                                    u.ctypes.data_as(_D), r.ctypes.data_as(_D), C.c_int64(0), C.c_int64(400),
                                    st.ctypes.data_as(C.POINTER(C.c_int32)), ts.ctypes.data_as(C.POINTER(C.c_int64)))
    assert (st == fb.K["FB_ST_LOST_BALANCE"]).all() and (ts > 0).all()
    assert (np.abs(r[2]) > np.pi / 4).all()
    # one step earlier nobody has fallen: the tilt crossed 45° in the step recorded
    k = 0
    r1 = r0[:, k:k + 1].copy(); st1 = np.zeros(1, np.int32)
    oracle.lib.fo_robot2d_step(C.c_int64(1), vp.ctypes.data_as(_D), gp.ctypes.data_as(_D), C.c_double(0.01), C.c_int32(2), C.c_int32(1),
                               u[:, :1].copy().ctypes.data_as(_D), r1.ctypes.data_as(_D), C.c_int64(0), C.c_int64(int(ts[k]) - 1), st1.ctypes.data_as(C.POINTER(C.c_int32)))
    assert st1[0] == 0 and abs(r1[2, 0]) <= np.pi / 4
