"""Terminated aircraft: where "identical results" is decided at the edges. The reference STOPS a simulation at the first
SimulationTermination / ArgumentError (lib/FlightCore/src/sim.jl:561-570; thrown from FlightPhysics/src/landinggear.jl:331-347,
geodesy.jl:218-221, atmosphere.jl:133, FlightApps/src/robot2d/robot2d.jl:553-561) and leaves mdl.x / mdl.s as they stand at the throw.
The oracle's steppers do exactly that (oracle/fo_c172.hpp `c172_step`: throwing mode + catch); the HIP path must agree with it on
every aircraft of a batch driven into each termination:

    * the status WORD, bit for bit (one bit: the first exception),
    * the step index and the place (fb_get_termination: FB_TERM_* of include/flightbatch.h),
    * the frozen state to 1e-6 — for an f_ode! that threw at an RK stage that is the stage's ARGUMENT, for a crash inside f_step! the
      new state with the part of f_step! that ran before the throw,
    * and on everything that is still flying, as before.
"""
import ctypes as C

import numpy as np
import pytest

import conditioning

from oracle_binding import OracleX
from test_gpu_parity import lattice_trim_params, state_scale
from test_gpu_c172x import ref_to_dev_rows, x_scale

pytestmark = pytest.mark.gpu
_D = C.POINTER(C.c_double)


def qmul(a, b):
    return np.stack([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                     a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])


def q_wb_from_euler(ps, th, ph):
    z = np.zeros_like(ps)
    return qmul(qmul(np.stack([np.cos(ps/2), z, z, np.sin(ps/2)]), np.stack([np.cos(th/2), z, np.sin(th/2), z])),
                np.stack([np.cos(ph/2), np.sin(ph/2), z, z]))


def q_ew_from_latlon(lat, lon):
    """q_ew = Rz(lon) ∘ Ry(-(lat + π/2)) (wander angle 0)"""
    a = -(lat + np.pi / 2)
    z = np.zeros_like(lat)
    return qmul(np.stack([np.cos(lon/2), z, z, np.sin(lon/2)]), np.stack([np.cos(a/2), z, np.sin(a/2), z]))


def geoid(oracle, lat, lon):
    out = np.zeros_like(lat)
    for k in range(lat.size):
        n_e = np.array([np.cos(lat[k]) * np.cos(lon[k]), np.cos(lat[k]) * np.sin(lon[k]), np.sin(lat[k])])
        out[k] = oracle.lib.fo_geoid_height(n_e.ctypes.data_as(_D))
    return out


def compare_terminated(fb, w, xo, so, sto, tstep_o, twhere_o, label, min_terminated=1000, tol=1e-6):
    """GPU world against the oracle's (x, s, status, term_step, term_where) on ALL aircraft."""
    K = fb.K
    st = w.status
    tstep, twhere = w.termination
    term = sto != 0
    nterm = int(term.sum())
    print(f"{label}: {nterm} of {st.size} aircraft terminated; places:", {int(c): int((twhere_o == c).sum()) for c in np.unique(twhere_o)})
    assert nterm >= min_terminated, f"{label}: the scenario must terminate at least {min_terminated} aircraft, it terminated {nterm}"
    assert np.array_equal(st, sto), f"{label}: {int((st != sto).sum())} status words differ (GPU {np.unique(st)}, oracle {np.unique(sto)})"
    assert np.array_equal(twhere, twhere_o), f"{label}: place of termination differs on {int((twhere != twhere_o).sum())} aircraft"
    assert np.array_equal(tstep, tstep_o), f"{label}: step of termination differs on {int((tstep != tstep_o).sum())} aircraft"
    assert (tstep[~term] == -1).all() and (twhere[~term] == K["FB_TERM_NONE"]).all()
    x = w.x
    err = np.abs(x - xo) / state_scale(xo)
    e_term = err[:, term].max() if nterm else 0.0
    e_live = err[:, ~term].max() if (~term).any() else 0.0
    print(f"{label}: max scaled error, terminated {e_term:.3e}, still running {e_live:.3e}")
    assert e_term < tol, (label, e_term, np.unravel_index(err[:, term].argmax(), err[:, term].shape))
    assert e_live < tol, (label, e_live)
    assert np.array_equal(w.s, so), f"{label}: discrete states differ"
    return term


def ground_batch(fb, oracle, n, seed, sink):
    """aircraft a few metres above a runway at 0 m, level, descending at `sink` m/s: some touch down within limits, the hard ones
    exceed the dampers' 10 m/s compression-rate limit at touchdown (landinggear.jl:341-344)"""
    rng = np.random.default_rng(seed)
    x = np.zeros((27, n))
    x[8] = 0.5
    th = rng.uniform(-0.02, 0.08, n); ph = rng.uniform(-0.03, 0.03, n); ps = rng.uniform(-np.pi, np.pi, n)
    x[12:16] = q_wb_from_euler(ps, th, ph)
    lat = np.full(n, 0.7); lon = np.full(n, -0.3)
    x[16:20] = q_ew_from_latlon(lat, lon)
    x[20] = geoid(oracle, lat[:1], lon[:1])[0] + rng.uniform(2.2, 4.0, n)     # terrain at 0 m orthometric; gear legs ~1.9 m long
    x[21:24] = rng.normal(0, 0.02, (3, n))
    x[24] = rng.uniform(25, 40, n); x[25] = rng.normal(0, 0.3, n); x[26] = sink
    x[9] = 100.0
    s = np.zeros((2, n), np.int32); s[1] = 2
    u = np.zeros((16, n)); u[11:16] = np.array([75, 75, 0, 0, 50.0])[:, None]; u[0] = 0.2; u[1] = 0.5
    ui = np.full(n, 4 | 8, np.int32)
    return x, s, u, ui


@pytest.mark.parametrize("spl", [1, 25])
def test_ground_crash_on_hard_landings(fb, oracle, spl):
    """GroundCrash out of f_step! (landinggear.jl:331-347): x = x_{n+1} with the quaternions renormalised and the stall flag updated,
    the regulator resets of the units ahead of the one that threw done — and the engine's state machine not run."""
    n = 4096
    rng = np.random.default_rng(31)
    x, s, u, ui = ground_batch(fb, oracle, n, 31, rng.uniform(7.5, 15.0, n))
    x[2:8] = rng.normal(0, 0.2, (6, n))          # non-zero friction regulators: the resets that do / do not happen are visible
    ui[::5] |= 2                                  # engine stop requested on a fifth: a state-machine step a crash must NOT take
    w = fb.BatchedWorld(n)
    w.set_state(x, s); w.u = u; w.ui = ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=spl)
    fb.step(sim, 0.6); w.sync()
    xo, so, sto, tso, two = oracle.step_term(x, u, ui, s, oracle.default_env(), 0.01, 60)
    term = compare_terminated(fb, w, xo, so, sto, tso, two, f"hard landings (steps_per_launch {spl})")
    assert (sto[term] == fb.K["FB_ST_GROUND_CRASH"]).all() and (two[term] == fb.K["FB_TERM_F_STEP"]).all()
    assert (~term).sum() > 200, "some aircraft must survive the touchdown"
    # frozen means frozen
    xa, sa = w.x, w.s
    fb.step(sim, 0.1); w.sync()
    assert np.array_equal(xa[:, term], w.x[:, term]) and np.array_equal(sa[:, term], w.s[:, term])
    w.close()


def flying_batch(fb, oracle, n, seed, lat, lon, h_e, climb, env_kw):
    """trimmed aircraft (device trim at a benign altitude) moved to altitude h_e with a vertical speed `climb` (m/s, + up)"""
    rng = np.random.default_rng(seed)
    tp = fb.TrimParameters(EAS=rng.uniform(40.0, 55.0, n), h_e=1000.0, ψ_nb=rng.uniform(-np.pi, np.pi, n))
    w = fb.BatchedWorld(n)
    fb.f_init(w, tp)
    assert w.trim_success.all()
    x, s, u, ui = w.x, w.s, w.u, w.ui
    w.close()
    x[16:20] = q_ew_from_latlon(lat, lon)
    x[20] = h_e
    # pitch the velocity vector: v_eb_b keeps its trimmed body components, the attitude is pitched by asin(climb / V) about body y
    V = np.sqrt(x[24] ** 2 + x[25] ** 2 + x[26] ** 2)
    dth = np.arcsin(np.clip(climb / V, -0.9, 0.9))
    z = np.zeros(n)
    x[12:16] = qmul(x[12:16], np.stack([np.cos(dth / 2), z, np.sin(dth / 2), z]))
    return x, s, u, ui


def run_range_case(fb, oracle, x, s, u, ui, env_kw, nsteps, spl, label, bit):
    n = x.shape[1]
    env = oracle.default_env(**env_kw)
    w = fb.BatchedWorld(n)
    if "h_trn" in env_kw:
        w.set_params(h_terrain=env_kw["h_trn"])
    w.set_state(x, s); w.u = u; w.ui = ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=spl)
    fb.step(sim, nsteps * 0.01); w.sync()
    xo, so, sto, tso, two = oracle.step_term(x, u, ui, s, env, 0.01, nsteps)
    term = compare_terminated(fb, w, xo, so, sto, tso, two, label)
    assert (sto[term] == bit).all()
    places = set(np.unique(two[term]).tolist())
    K = fb.K
    assert {K["FB_TERM_F_ODE_K2"], K["FB_TERM_F_ODE_K4"]} <= places, (label, places)   # stage arguments at t + dt/2 and at t + dt
    xa = w.x
    fb.step(sim, 0.05); w.sync()
    assert np.array_equal(xa[:, term], w.x[:, term])
    w.close()
    return term, two


@pytest.mark.parametrize("spl", [1, 40])
def test_altitude_floor_orthometric(fb, oracle, spl):
    """ArgumentError of Altitude{D}(h) (geodesy.jl:218-221) raised by HOrth(h_e, n_e) in the kinematics (kinematics.jl:199): a descent
    through h_o = -1000 m where the geoid is above the ellipsoid (N = +17 m at 0°N 0°E), over terrain far below. The throw comes at
    whichever evaluation first sees h_o < h_min: RK stages k2, k3, k4 (x = the stage's argument) or the new state."""
    n = 2048
    rng = np.random.default_rng(41)
    lat = np.zeros(n); lon = np.zeros(n)
    N = geoid(oracle, lat[:1], lon[:1])[0]
    assert N > 5
    h_e = -1000.0 + N + rng.uniform(0.05, 6.0, n)
    x, s, u, ui = flying_batch(fb, oracle, n, 41, lat, lon, h_e, -rng.uniform(3.0, 9.0, n), {})
    term, two = run_range_case(fb, oracle, x, s, u, ui, dict(h_trn=-5000.0), 120, spl, f"h_o floor (spl {spl})", fb.K["FB_ST_ALT_RANGE"])
    assert term.mean() > 0.5


def test_altitude_floor_ellipsoidal_and_wheels(fb, oracle):
    """Where the geoid is BELOW the ellipsoid (N ≈ -100 m south of India) h_e reaches -1000 m first — and the wheels, which hang below the
    body origin, reach it before the origin does: Geographic(r_ew0_e) in the strut (landinggear.jl:240) throws first. Near the floor the
    stepping kernels evaluate the wheels' altitudes even though the terrain is far away."""
    n = 2048
    rng = np.random.default_rng(43)
    lat = np.full(n, 0.05); lon = np.full(n, 1.38)
    N = geoid(oracle, lat[:1], lon[:1])[0]
    assert N < -50
    h_e = -1000.0 + rng.uniform(1.0, 9.0, n)     # (the lowest start with their wheels under the floor: the first evaluation after init throws)
    x, s, u, ui = flying_batch(fb, oracle, n, 43, lat, lon, h_e, -rng.uniform(3.0, 9.0, n), {})
    term, two = run_range_case(fb, oracle, x, s, u, ui, dict(h_trn=-5000.0), 150, 30, "h_e floor, wheels first", fb.K["FB_ST_ALT_RANGE"])
    assert term.mean() > 0.5 and (two == fb.K["FB_TERM_F_ODE_REEVAL"]).sum() > 50


@pytest.mark.parametrize("spl", [1, 50])
def test_isa_ceiling(fb, oracle, spl):
    """ArgumentError("Altitude out of bounds") of ISAData above the last layer (atmosphere.jl:116-135): geopotential 84 852 m, i.e.
    h_o = 86 000 m, crossed in a zoom climb."""
    n = 2048
    rng = np.random.default_rng(47)
    lat = rng.uniform(-1.0, 1.0, n); lon = rng.uniform(-3.0, 3.0, n)
    a = 6378137.0
    h_o_ceiling = 84852.0 * a / (a - 84852.0)                     # h_geop = h a / (a + h)  (geodesy.jl:232-246)
    h_e = h_o_ceiling + geoid(oracle, lat, lon) - rng.uniform(0.1, 12.0, n)
    x, s, u, ui = flying_batch(fb, oracle, n, 47, lat, lon, h_e, rng.uniform(8.0, 30.0, n), {})
    term, two = run_range_case(fb, oracle, x, s, u, ui, {}, 100, spl, f"ISA ceiling (spl {spl})", fb.K["FB_ST_ISA_RANGE"])
    assert term.mean() > 0.5


def test_robot2d_lost_balance_matches_oracle(fb, oracle):
    """LostBalance out of f_step! (robot2d.jl:553-561): x = x_k right after the RK update, no f_periodic! in that step."""
    from test_oracle_robot2d import DEFAULT_VP, gains_from_h5
    from test_gpu_robot2d import oracle_init
    n = 4096
    rng = np.random.default_rng(53)
    w = fb.Robot2DWorld(n)
    ipar = fb.InitParameters(u_m=rng.uniform(-0.2, 0.2, n), ω=rng.uniform(-0.05, 0.05, n), η=rng.uniform(-1, 1, n))
    fb.f_init(w, ipar)
    vp = DEFAULT_VP.copy(); gp = gains_from_h5()
    r = oracle_init(oracle.lib, vp, ipar.pack(n))
    th0 = rng.uniform(0.02, 0.5, n) * rng.choice([-1.0, 1.0], n)       # initial tilt: the velocity loop saves some, not the steep ones
    r[2] = th0
    w.set_state(r)
    u = np.zeros((4, n)); u[0] = rng.integers(0, 3, n); u[2] = rng.uniform(-0.5, 0.5, n); u[3] = rng.uniform(-2, 2, n)
    w.u = u
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=33)
    fb.step(sim, 3.0); w.sync()
    st = np.zeros(n, np.int32); ts = np.full(n, -1, np.int64)
    oracle.lib.fo_robot2d_step_term(C.c_int64(n), vp.ctypes.data_as(_D), gp.ctypes.data_as(_D), C.c_double(0.01), C.c_int32(2), C.c_int32(1),
                                    np.ascontiguousarray(u).ctypes.data_as(_D), r.ctypes.data_as(_D), C.c_int64(0), C.c_int64(300),
                                    st.ctypes.data_as(C.POINTER(C.c_int32)), ts.ctypes.data_as(C.POINTER(C.c_int64)))
    term = st != 0
    print("Robot2D:", int(term.sum()), "of", n, "lost their balance")
    assert term.sum() >= 1000 and (~term).sum() >= 200
    tstep, twhere = w.termination
    assert np.array_equal(w.status, st) and (st[term] == fb.K["FB_ST_LOST_BALANCE"]).all()
    assert np.array_equal(tstep, ts) and (twhere[term] == fb.K["FB_TERM_F_STEP"]).all() and (twhere[~term] == 0).all()
    err = np.abs(w.x - r) / np.maximum(np.abs(r), 1.0)
    print("Robot2D: max scaled error, fallen", err[:, term].max(), "standing", err[:, ~term].max())
    assert err.max() < 1e-9
    w.close()


def test_x2_crash_under_autopilot(fb, oracle):
    """Cessna172Xv2 flown into the ground by its own autopilot (EAS + climb-rate mode, a steep descent demanded over a runway 15-60 m
    below): GroundCrash out of f_step! at touchdown — no control-law update in that step (cb_step throws before cb_periodic,
    sim.jl:204-218) — state, control-law record, status word, step and place against the oracle."""
    gains = fb.ctl_gains.ctl_gains_blob()
    K = fb.K
    n = 4096
    rng = np.random.default_rng(59)
    h_agl = rng.uniform(15.0, 60.0, n)
    N0 = geoid(oracle, np.zeros(1), np.zeros(1))[0]
    tp = fb.TrimParameters(EAS=rng.uniform(42.0, 55.0, n), h_e=N0 + 2.0 + h_agl, ψ_nb=rng.uniform(-np.pi, np.pi, n), γ_wb_n=-0.05)
    w = fb.Cessna172Xv2World(n, gains=gains)
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.init(sim, tp)
    assert w.trim_success.all()
    X = OracleX(oracle, gains)
    env = oracle.default_env()
    o = X.trim_init(tp.pack(n), fb.TrimState(n), env, 0.02)
    o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = float(fb.ModeControlLon.EAS_clm)
    cu[K["FB_CU_LAT_MODE_REQ"]] = float(fb.ModeControlLat.φ_β)
    cu[K["FB_CU_CLM_REF"]] = -rng.uniform(7.0, 15.0, n)
    w.cu = cu
    o["cu"] = np.ascontiguousarray(cu.copy())
    perm = ref_to_dev_rows(K)
    o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
    fb.step(sim, 12.0); w.sync()
    o_start = {k: np.array(v, copy=True) for k, v in o.items() if isinstance(v, np.ndarray)}
    X.step_term(o, env, 0.01, 2, 1200)
    pert = conditioning.x2_perturbed_runs(X, o_start, env, 1200, 20, N0, 1e-12, K=4, jitter=conditioning.ULP_R, seed=3, threads=16)   # the oracle against itself, below
    st, sto = w.status, o["status"]
    term = sto != 0
    print("Xv2 under autopilot:", int(term.sum()), "of", n, "crashed; status words", np.unique(sto), "places", np.unique(o["term_where"]))
    assert np.array_equal(st, sto)
    assert term.sum() >= 1000 and (~term).sum() >= 200
    tstep, twhere = w.termination
    assert np.array_equal(twhere, o["term_where"]) and np.array_equal(tstep, o["term_step"])
    assert (sto[term] == K["FB_ST_GROUND_CRASH"]).all() and (twhere[term] == K["FB_TERM_F_STEP"]).all()
    xo = o["x"][perm]
    sc = x_scale(o["x"])[perm]
    err = np.abs(w.x - xo) / sc
    he_row = int(np.where(perm == 20)[0][0])
    flying = ~term & (xo[he_row] - N0 > 8.0)          # still clear of the runway at the end
    rolling = ~term & ~flying                          # touched down hard, survived, and have been rolling / bouncing since
    print("Xv2: max scaled state error, crashed", err[:, term].max(), "| still flying", err[:, flying].max() if flying.any() else 0.0,
          f"({int(flying.sum())}) | rolling", err[:, rolling].max() if rolling.any() else 0.0, f"({int(rolling.sum())})")
    # the crashed aircraft (the point of this test) and the ones still in the air hold the north-star tolerance. The survivors of a
    # 6-10 m/s touchdown have spent up to ten seconds bouncing on dampers and stick-slip friction regulators (k_i = 400 1/s,
    # landinggear.jl:411-476): they are held to the MEASURED conditioning of that roll — the oracle run again with v_eb_b nudged by 1e-12
    # at touchdown and the altitude jittered by one ulp of the geocentric radius per step (tests/conditioning.py): per aircraft max(1e-6, 10 x |oracle - oracle'|), state and control-law record together
    assert err[:, term].max() < 1e-6
    assert not flying.any() or err[:, flying].max() < 1e-6
    cerr = np.abs(w.cs - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)
    assert cerr[:, term | flying].max() < 1e-6, cerr[:, term | flying].max()
    if rolling.any():
        def lane_err(xx, cc):
            return np.maximum((np.abs(xx - o["x"]) / x_scale(o["x"])).max(0), (np.abs(cc - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)).max(0))
        E = np.stack([lane_err(p["x"], p["cs"])[rolling] for p in pert])
        conditioning.check_against_envelope(np.maximum(err.max(0), cerr.max(0))[rolling], E, "Xv2 under autopilot, rolling survivors")
    assert np.array_equal(w.s, o["s"])
    w.close()


@pytest.mark.parametrize("kin", ["WA", "ECEF", "NED"])
def test_steep_descents_into_the_ground_in_every_mechanisation(fb, oracle, kin):
    """Rolling, steep descents (sink 7-17 m/s) from 8-30 m onto the terrain, in each kinematic mechanisation against the oracle in the same
    one: the aircraft that hit hard end in GroundCrash — one wheel first, 10 cm into the ground within a step, 170 kN on a strut — and their
    frozen state, status word, step and place must match. The step that carries an aircraft into contact is where the RK stages' off-unit
    attitude quaternion matters: kinematics.y.q_en is q_eb ∘ q_nb' in WA (kinematics.jl:195: off unit by |q_wb|² at a stage) but ltf(n_e)
    itself in ECEF and NED (:293, :377); until the end of round 3 the contact branch formed it the WA way in every mechanisation, and
    ECEF aircraft ended 6·10⁻⁵ away from the oracle (found by the 10 000-step ECEF soak, tools/soak_duo.py)."""
    K = fb.K
    n = 4096
    nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
    rng = np.random.default_rng(83)
    h_trn = 120.0
    lat, lon = 0.6, -1.1
    N0 = geoid(oracle, np.array([lat]), np.array([lon]))[0]
    tp = fb.TrimParameters(EAS=rng.uniform(45.0, 58.0, n), h_e=h_trn + N0 + 1.9 + rng.uniform(8.0, 30.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n),
                           γ_wb_n=-rng.uniform(0.15, 0.30, n), n_e=(np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)))
    w = fb.BatchedWorld(n, kinematics=kin)
    w.set_params(h_terrain=h_trn)
    fb.f_init(w, tp)
    x0, s0, u0, ui0 = w.x, w.s, w.u.copy(), w.ui
    u0[K["FB_U_AILERON"]] += rng.uniform(-0.4, 0.4, n)      # they roll on the way down: one wheel first
    u0[K["FB_U_ELEVATOR"]] += rng.uniform(-0.1, 0.1, n)
    w.u = u0
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=40)
    fb.step(sim, 4.0); w.sync()
    env = oracle.default_env(h_trn=h_trn)
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        x27 = np.zeros((27, n)); x27[:12 + nk] = x0[:12 + nk]; x27[21:] = x0[12 + nk:]
        xo27, so, sto, tso, two = oracle.step_term(x27, u0, ui0, s0, env, 0.01, 400, threads=16)
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    xo = np.vstack([xo27[:12 + nk], xo27[21:]])
    st = w.status
    tstep, twhere = w.termination
    term = sto != 0
    print(f"{kin}: {int(term.sum())} of {n} ended; status words {np.unique(sto)}, places {np.unique(two[term])}")
    assert term.sum() >= 1000 and np.array_equal(st, sto) and np.array_equal(tstep, tso) and np.array_equal(twhere, two)
    assert (sto[term] == K["FB_ST_GROUND_CRASH"]).all()
    err = np.abs(w.x - xo) / np.maximum(np.abs(xo), 1.0)
    per = err[:, term].max(0)
    print(f"{kin}: frozen state of the crashed aircraft against the oracle: quantiles 50 / 99 / 100 % {np.quantile(per, [0.5, 0.99, 1.0])}")
    assert per.max() < 1e-6
    assert np.array_equal(w.s[:, term], so[:, term])
    w.close()


def test_survivors_sharing_a_wave_with_a_thrower_keep_their_own_launch(fb, oracle):
    """Regression test of the termination replay (k_step_air's `goto restart`): a lane whose f_ode! threw is stepped a second time, up to
    the evaluation that threw, while the OTHER lanes of its wave sit that pass out. Those bystanders must write back what their own launch
    left — here their stall flag and engine state CHANGE during the launch (a stall flag that starts set at a small angle of attack is
    cleared by the first f_step!, c172.jl:720; an engine stop request takes the running engine to `off`, piston.jl:300-312), so a write-back
    of launch-start values (the defect: s at t for a state x at t + K dt) shows in `s` and, one launch later, in the engine-speed row.
    Throwers (a descent through h_e = -1000 m where the geoid is below the ellipsoid) and survivors (a slow climb away from it, within the
    10 m of the floor that keep a lane in the ground-capable pass) are interleaved at random, so practically every 64-lane wave holds both;
    every aircraft against the oracle. (Seen to fail on a library built with -DFB_REPLAY_BYSTANDER_DEFECT: profiles/r04_replay_defect.txt.)"""
    n = 4096
    rng = np.random.default_rng(97)
    lat = np.full(n, 0.05); lon = np.full(n, 1.38)     # the geoid is ~100 m BELOW the ellipsoid here: h_e reaches h_min first, and a lane
    N = geoid(oracle, lat[:1], lon[:1])[0]              # within 10 m of it is stepped by the ground-capable pass (the one that replays)
    assert N < -50
    thrower = rng.random(n) < 0.5
    h_e = -1000.0 + np.where(thrower, rng.uniform(2.5, 8.0, n), rng.uniform(4.0, 7.0, n))
    climb = np.where(thrower, -rng.uniform(3.0, 9.0, n), rng.uniform(0.3, 1.0, n))
    x, s, u, ui = flying_batch(fb, oracle, n, 97, lat, lon, h_e, climb, {})
    s = s.copy(); ui = ui.copy()
    flagged = rng.random(n) < 0.5
    s[0, flagged] = 1                                   # stall flag set at a cruise angle of attack: the first f_step! clears it
    stopping = rng.random(n) < 0.33
    ui[stopping] |= 2                                   # FB_UI_ENG_STOP: running -> off at the first f_step!
    env_kw = dict(h_trn=-5000.0)
    env = oracle.default_env(**env_kw)
    for spl, nsteps in ((40, 120), (25, 25)):           # (the second: ONE launch, so what is compared is exactly what the replay's epilogue wrote)
        w = fb.BatchedWorld(n)
        w.set_params(h_terrain=env_kw["h_trn"])
        w.set_state(x, s); w.u = u; w.ui = ui
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=spl)
        fb.step(sim, nsteps * 0.01); w.sync()
        xo, so, sto, tso, two = oracle.step_term(x, u, ui, s, env, 0.01, nsteps)
        term = compare_terminated(fb, w, xo, so, sto, tso, two, f"throwers and survivors in one wave (spl {spl}, {nsteps} steps)", min_terminated=200)
        live = ~term
        assert live.sum() > 1000
        changed = live & ((so[0] != s[0]) | (so[1] != s[1]))
        mixed_waves = sum(1 for k in range(0, n, 64) if term[k:k + 64].any() and changed[k:k + 64].any())
        print(f"spl {spl}: {int(changed.sum())} survivors changed stall flag / engine state during the run; {mixed_waves} of {n // 64} waves hold such a survivor AND a thrower")
        assert changed.sum() > 500 and mixed_waves > 40
        assert np.array_equal(w.s[:, live], so[:, live])
        w.close()


def test_x2_survivors_sharing_a_wave_with_a_thrower_keep_their_actuators(fb, oracle):
    """The same for Cessna172Xv2 through the altitude floor under its autopilot: the bystanders of a replayed wave must keep the ACTUATOR
    positions their launch reached (device rows 27-33, moving under the control laws' commands) — and the brake actuators the steps they
    completed — not the launch-start values. All 34 rows, the control-law record, s, status word, step and place against the oracle."""
    gains = fb.ctl_gains.ctl_gains_blob()
    K = fb.K
    n = 2048
    rng = np.random.default_rng(101)
    assert geoid(oracle, np.array([0.05]), np.array([1.38]))[0] < -50
    tp = fb.TrimParameters(EAS=rng.uniform(42.0, 55.0, n), h_e=1000.0, ψ_nb=rng.uniform(-np.pi, np.pi, n))
    w = fb.Cessna172Xv2World(n, gains=gains)
    w.set_params(h_terrain=-5000.0)
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.init(sim, tp)
    assert w.trim_success.all()
    perm = ref_to_dev_rows(K)
    thrower = rng.random(n) < 0.5
    xw = w.x
    he_row = int(np.where(perm == 20)[0][0])
    qew_rows = [int(np.where(perm == 16 + k)[0][0]) for k in range(4)]
    xw[qew_rows] = q_ew_from_latlon(np.full(n, 0.05), np.full(n, 1.38))      # geoid ~100 m below the ellipsoid: h_e reaches h_min first
    xw[he_row] = -1000.0 + np.where(thrower, rng.uniform(2.5, 7.0, n), rng.uniform(4.0, 8.0, n))   # (within 10 m of it: the ground-capable pass)
    w.x = xw
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = float(fb.ModeControlLon.EAS_clm)
    cu[K["FB_CU_LAT_MODE_REQ"]] = float(fb.ModeControlLat.φ_β)
    cu[K["FB_CU_CLM_REF"]] = np.where(thrower, -rng.uniform(5.0, 12.0, n), rng.uniform(0.3, 1.0, n))
    cu[K["FB_CU_PHI_REF"]] = rng.uniform(-0.3, 0.3, n)            # the lateral channel moves aileron and rudder too
    w.cu = cu
    uu = w.u
    uu[K["FB_U_BRAKE_LEFT"]] = rng.uniform(0.2, 1.0, n); uu[K["FB_U_BRAKE_RIGHT"]] = rng.uniform(0.2, 1.0, n)   # brake actuators on their way to a command
    w.u = uu
    X = OracleX(oracle, gains)
    env = oracle.default_env(h_trn=-5000.0)
    o = X.trim_init(tp.pack(n), fb.TrimState(n), env, 0.02)
    o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
    o["cu"] = np.ascontiguousarray(cu.copy())
    o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
    act0 = o["x"][27:34].copy()
    fb.step(sim, 2.0); w.sync()
    X.step_term(o, env, 0.01, 2, 200)
    st, sto = w.status, o["status"]
    term = sto != 0
    print("Xv2 through the altitude floor:", int(term.sum()), "of", n, "ended; places", np.unique(o["term_where"][term]))
    assert np.array_equal(st, sto) and (sto[term] == K["FB_ST_ALT_RANGE"]).all()
    assert term.sum() >= 400 and (~term).sum() >= 400
    tstep, twhere = w.termination
    assert np.array_equal(twhere, o["term_where"]) and np.array_equal(tstep, o["term_step"])
    xo = o["x"][perm]
    err = np.abs(w.x - xo) / x_scale(o["x"])[perm]
    act_rows = [int(np.where(perm == 27 + k)[0][0]) for k in range(7)]
    moved = np.abs(o["x"][27:34] - act0).max(0)
    live = ~term
    mixed_waves = sum(1 for k in range(0, n, 64) if term[k:k + 64].any() and live[k:k + 64].any())
    print(f"Xv2: max scaled error, ended {err[:, term].max():.2e}, survivors {err[:, live].max():.2e} (their actuator rows {err[act_rows][:, live].max():.2e}; "
          f"actuators moved by up to {moved[live].max():.2f}); {mixed_waves} of {n // 64} waves hold both")
    assert mixed_waves > 20 and (moved[live] > 0.01).mean() > 0.9
    assert err[:, term].max() < 1e-6 and err[:, live].max() < 1e-6
    cerr = np.abs(w.cs - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)
    assert cerr.max() < 1e-6 and np.array_equal(w.s, o["s"])
    w.close()


def test_termination_record_outside_step_and_checkpoint(fb, oracle):
    """fb_get_termination promises a step and a place next to EVERY termination bit, and -1 / FB_TERM_NONE without one: a bit raised by
    a single-call verb (fb_f_ode on a state below the altitude floor) or set by the host carries FB_TERM_OUTSIDE_STEP with the step
    count of the moment; whatever clears the status words clears the record; a checkpoint carries the record along."""
    K = fb.K
    n = 256
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 50, n)))
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=10)
    fb.step(sim, 0.3); w.sync()
    ts, tw = w.termination
    assert (ts == -1).all() and (tw == K["FB_TERM_NONE"]).all()
    x = w.x; x[20, ::4] = -1500.0; w.x = x          # every fourth aircraft below h_min: its next f_ode! raises FB_ST_ALT_RANGE
    fb.f_ode(w); w.sync()
    st = w.status
    ts, tw = w.termination
    hit = np.zeros(n, bool); hit[::4] = True
    assert (st[hit] == K["FB_ST_ALT_RANGE"]).all() and (st[~hit] == 0).all()
    assert (tw[hit] == K["FB_TERM_OUTSIDE_STEP"]).all() and (ts[hit] == 30).all() and (ts[~hit] == -1).all() and (tw[~hit] == 0).all()
    # a checkpoint carries the record; a restore into a fresh world reproduces it
    ck = w.checkpoint()
    w2 = fb.BatchedWorld(n)
    w2.restore(ck)
    ts2, tw2 = w2.termination
    assert np.array_equal(ts2, ts) and np.array_equal(tw2, tw) and np.array_equal(w2.status, st)
    # fb_set_status alone: new words get FB_TERM_OUTSIDE_STEP at the current step count, cleared words lose their record
    st3 = np.zeros(n, np.int32); st3[1::4] = K["FB_ST_GROUND_CRASH"]
    fb._lib.check(fb.lib.fb_set_status(w2._h, st3.ctypes.data_as(C.POINTER(C.c_int32))))
    ts3, tw3 = w2.termination
    assert (tw3[1::4] == K["FB_TERM_OUTSIDE_STEP"]).all() and (ts3[1::4] == 30).all() and (ts3[::4] == -1).all() and (tw3[::4] == 0).all()
    # init semantics clear both
    w2.set_state(ck["x"], ck["s"])
    ts4, tw4 = w2.termination
    assert (w2.status == 0).all() and (ts4 == -1).all() and (tw4 == 0).all()
    w.close(); w2.close()
