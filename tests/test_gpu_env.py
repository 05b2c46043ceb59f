"""Per-aircraft environment rows (fb_set_env): every aircraft in its own wind, sea-level conditions and terrain elevation — in the
reference each simulation owns its world, so N simulations have N environments (FP/atmosphere.jl:75-84,156-165,269-278;
FP/terrain.jl:34-48; FP/world.jl:20-32). The oracle evaluates each aircraft with its own Env; the HIP path reads the rows in every
verb (trim, f_ode!, f_step!, the steppers of both passes, Cessna172Xv2's control-law tap)."""
import numpy as np
import pytest

from test_gpu_parity import lattice_trim_params, state_scale

pytestmark = pytest.mark.gpu


def random_env(fb, n, seed, h_trn=None):
    K = fb.K
    rng = np.random.default_rng(seed)
    e = np.zeros((K["FB_NENV"], n))
    e[K["FB_ENV_WIND_N"]] = rng.uniform(-12, 12, n); e[K["FB_ENV_WIND_E"]] = rng.uniform(-12, 12, n); e[K["FB_ENV_WIND_D"]] = rng.uniform(-2, 2, n)
    e[K["FB_ENV_T_SL"]] = rng.uniform(258.0, 313.0, n); e[K["FB_ENV_P_SL"]] = rng.uniform(97000.0, 104500.0, n)
    e[K["FB_ENV_H_TERRAIN"]] = rng.uniform(-50.0, 150.0, n) if h_trn is None else h_trn
    return e


@pytest.mark.parametrize("kin", ["WA", "ECEF", "NED"])
def test_per_aircraft_env_trim_f_ode_and_trajectory_match_oracle(fb, oracle, kin):
    """4096 aircraft with random winds / sea-level conditions / terrain elevations: the trim (it depends on density and wind), f_ode!
    with the full output record, and 1000 RK4 steps against the oracle at 1e-6 — and the rows really act (the same batch in the
    batch-wide default environment ends elsewhere)."""
    K = fb.K
    n = 4096
    nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
    tp = lattice_trim_params(fb, n, seed=61)
    env6 = random_env(fb, n, 7)
    w = fb.BatchedWorld(n, kinematics=kin)
    w.env = env6
    assert np.array_equal(w.env, env6)
    fb.f_init(w, tp)
    assert w.trim_success.mean() > 0.99
    ok = w.trim_success
    oenv = oracle.env_rows(env6)
    with oracle.per_aircraft_env():
        ref = oracle.trim(tp.pack(n), fb.TrimState(n), oenv)
        assert np.array_equal(ref["ok"], ok)
        assert np.abs(w.trim_state - ref["ts"])[:, ok].max() < 1e-8
        # off trim, so that the dynamics are exercised
        rng = np.random.default_rng(3)
        x = w.x
        x[12 + nk:12 + nk + 3] += rng.normal(0, 0.02, (3, n)); x[12 + nk + 3:] += rng.normal(0, 1.0, (3, n))
        w.set_state(x, w.s)
        x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
        oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
        try:
            xo0 = np.zeros((27, n)); xo0[:12 + nk] = x0[:12 + nk]; xo0[21:] = x0[12 + nk:]
            xd = np.zeros((18 + nk, n)); fb.f_ode(w, xd)
            xdo, yo, sto0 = oracle.f_ode(xo0, u0, ui0, s0, oenv)
            xdo_abi = np.vstack([xdo[:12 + nk], xdo[21:]])
            assert (np.abs(xd - xdo_abi) / np.maximum(np.abs(xdo_abi), 1.0)).max() < 1e-9
            sc_y = np.maximum(np.abs(yo), 1.0); sc_y[22:25] = 6.4e6
            assert (np.abs(w.y - yo) / sc_y).max() < 1e-9
            sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
            fb.step(sim, 10.0); w.sync()
            xo, so, sto = oracle.step(xo0, u0, ui0, s0, oenv, 0.01, 1000)
        finally:
            oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    assert np.array_equal(w.status, sto) and np.array_equal(w.s, so)
    xo_abi = np.vstack([xo[:12 + nk], xo[21:]])
    err = (np.abs(w.x - xo_abi) / np.maximum(np.abs(xo_abi), 1.0))[:, sto == 0]
    print(kin, "per-aircraft environment, max scaled error after 1000 steps: %.3e" % err.max())
    assert err.max() < 1e-6
    # the rows act: the same initial state in the batch-wide default environment goes somewhere else
    w2 = fb.BatchedWorld(n, kinematics=kin)
    w2.set_state(x0, s0); w2.u = u0; w2.ui = ui0
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim2, 10.0); w2.sync()
    assert np.abs(w2.x[12 + nk + 3:] - w.x[12 + nk + 3:]).max() > 1.0      # body velocities differ by metres per second
    # env = None returns to the batch-wide block, bit for bit the plain path
    w.env = None
    assert w.env is None
    w.set_state(x0, s0)
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 10.0); w.sync()
    assert np.array_equal(w.x, w2.x)
    w.close(); w2.close()


def test_uniform_rows_equal_the_batch_wide_block(fb):
    """rows that repeat fb_params' block give the batch-wide kernels' results (same arithmetic: only where the nine values live differs),
    on both stepping paths: the one-wave kernel to the last bit, the wave-pair kernel (FLIGHTBATCH_DUO default) to rounding."""
    n = 2048
    tp = lattice_trim_params(fb, n, seed=5)
    wind = (3.0, -2.0, 0.5)
    wa = fb.BatchedWorld(n); wa.set_params(wind_ned=wind, T_sl=279.0, p_sl=99000.0)
    wb = fb.BatchedWorld(n); wb.set_params(wind_ned=wind, T_sl=279.0, p_sl=99000.0)
    wb.set_env()                                   # rows = the block
    fb.f_init(wa, tp); fb.f_init(wb, tp)
    assert np.array_equal(wa.trim_success, wb.trim_success)
    assert np.abs(wa.trim_state - wb.trim_state).max() < 1e-12
    wb.set_state(wa.x, wa.s); wb.u = wa.u; wb.ui = wa.ui
    xa = np.zeros((27, n)); xb = np.zeros((27, n))
    fb.f_ode(wa, xa); fb.f_ode(wb, xb)
    assert np.array_equal(xa, xb) and np.array_equal(wa.y, wb.y)
    for w in (wa, wb):
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
        fb.step(sim, 5.0); w.sync()
    assert (wa.status == 0).all() and (wb.status == 0).all()
    assert (np.abs(wa.x - wb.x) / state_scale(wa.x)).max() < 1e-10
    wa.close(); wb.close()


def test_per_aircraft_env_xv2_closed_loop_matches_oracle(fb, oracle):
    """Cessna172Xv2 with the autopilot every 2 steps, each aircraft in its own wind and air mass, 500 closed-loop steps vs the oracle"""
    from oracle_binding import OracleX
    from test_gpu_c172x import ref_to_dev_rows, x_scale
    K = fb.K
    n = 1024
    gains = fb.ctl_gains.ctl_gains_blob()
    tp = lattice_trim_params(fb, n, seed=71)
    env6 = random_env(fb, n, 9)
    w = fb.Cessna172Xv2World(n, gains=gains)
    w.env = env6
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.init(sim, tp)
    ok = w.trim_success
    assert ok.mean() > 0.99
    rng = np.random.default_rng(5)
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, n); cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, n)
    cu[K["FB_CU_EAS_REF"]] += rng.uniform(-3, 3, n); cu[K["FB_CU_CLM_REF"]] += rng.uniform(-1.5, 1.5, n)
    cu[K["FB_CU_PHI_REF"]] += rng.uniform(-0.3, 0.3, n); cu[K["FB_CU_CHI_REF"]] += rng.uniform(-0.5, 0.5, n)
    w.cu = cu
    perm = ref_to_dev_rows(K)
    X = OracleX(oracle, gains)
    oenv = oracle.env_rows(env6)
    o = dict(x=np.empty((34, n)), u=w.u, ui=w.ui, s=w.s, cu=np.ascontiguousarray(cu), cs=w.cs, status=np.zeros(n, np.int32), nstep=0)
    o["x"][perm] = w.x
    fb.step(sim, 5.0); w.sync()
    with oracle.per_aircraft_env():
        X.step(o, oenv, 0.01, 2, 500)
    assert np.array_equal(w.status, o["status"])
    live = (o["status"] == 0) & ok
    err = (np.abs(w.x - o["x"][perm]) / x_scale(o["x"])[perm])[:, live]
    cerr = (np.abs(w.cs - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0))[:, live]
    print("Xv2, per-aircraft environment: max scaled error after 500 closed-loop steps %.3e (record %.3e)" % (err.max(), cerr.max()))
    assert err.max() < 1e-6 and cerr.max() < 1e-6
    w.close()


def test_per_aircraft_terrain_elevation_ground_contact(fb, oracle):
    """aircraft descending onto terrain whose elevation differs per aircraft: touchdown happens where each aircraft's own terrain is
    (landing gear + ground-capable pass read the row; h_e is ellipsoidal, the terrain elevation orthometric: the geoid stands 17.2 m above the
    ellipsoid at ϕ = λ = 0), status words and the state of every aircraft against the oracle"""
    K = fb.K
    n = 512
    rng = np.random.default_rng(13)
    h_trn = rng.uniform(0.0, 400.0, n)
    env6 = random_env(fb, n, 15, h_trn=h_trn)
    env6[K["FB_ENV_WIND_N"]:K["FB_ENV_WIND_D"] + 1] *= 0.2
    tp = fb.TrimParameters(EAS=rng.uniform(33, 40, n), h_e=h_trn + 17.2 + rng.uniform(4.0, 14.0, n), γ_wb_n=-np.deg2rad(rng.uniform(2, 4, n)),
                           ψ_nb=rng.uniform(-np.pi, np.pi, n), flaps=1.0)
    w = fb.BatchedWorld(n)
    w.env = env6
    fb.f_init(w, tp)
    ok = w.trim_success
    assert ok.mean() > 0.95
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 10.0); w.sync()
    with oracle.per_aircraft_env():
        xo, so, sto = oracle.step(x0, u0, ui0, s0, oracle.env_rows(env6), 0.01, 1000)
    assert np.array_equal(w.status, sto)
    fb.f_ode(w)
    wow = w.y[K["FB_Y_LDG"] + 1] + w.y[K["FB_Y_LDG"] + 12] + w.y[K["FB_Y_LDG"] + 23]
    touched = (wow > 0) | (w.status != 0)
    print("touched down or terminated:", int(touched.sum()), "of", n, "; terminated:", int((sto != 0).sum()))
    assert touched.sum() > n // 4          # (weight on wheels at the LAST instant, or terminated: many more have bounced or float in ground effect)
    air = ok & (sto == 0) & ~touched
    err = (np.abs(w.x - xo) / state_scale(xo))
    assert err[:, air].max() < 1e-6
    # on the ground the friction regulators are ill-conditioned (docs/design/ground.md): position and attitude only, loosely
    gnd = ok & (sto == 0) & touched
    assert np.abs(w.x[20] - xo[20])[gnd].max() < 1e-3 and np.abs(w.x[12:16] - xo[12:16])[:, gnd].max() < 1e-3
    w.close()


def test_per_aircraft_env_every_stepping_path_agrees(fb, monkeypatch):
    """The rows are read by four stepping kernels in the WA mechanisation: k_step_duo<WA, X, PERENV> (default), k_step_air<WA, X, false, PERENV>
    (FLIGHTBATCH_DUO=0, and FB_F32 handles — the fp32 stepper has no per-aircraft form, such a handle is stepped in fp64) and the ground-capable
    pass behind each. Same batch, same rows: the paths agree to rounding (they order a few sums differently, like their batch-wide forms)."""
    K = fb.K
    n = 2048
    tp = lattice_trim_params(fb, n, seed=81)
    env6 = random_env(fb, n, 21)
    runs = {}
    w0 = fb.BatchedWorld(n); w0.env = env6
    fb.f_init(w0, tp)
    x0, s0, u0, ui0 = w0.x, w0.s, w0.u, w0.ui
    w0.close()
    for name, duo, dtype in (("duo", "1", "f64"), ("air", "0", "f64"), ("f32 handle", "1", "f32")):
        monkeypatch.setenv("FLIGHTBATCH_DUO", duo)       # (read when a handle is created)
        w = fb.BatchedWorld(n, dtype=dtype)
        w.env = env6
        w.set_state(x0, s0); w.u = u0; w.ui = ui0
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
        fb.step(sim, 5.0); w.sync()
        assert (w.status == 0).all()
        runs[name] = w.x
        w.close()
    monkeypatch.delenv("FLIGHTBATCH_DUO")
    sc = state_scale(runs["duo"])
    assert (np.abs(runs["air"] - runs["duo"]) / sc).max() < 1e-10
    assert np.array_equal(runs["f32 handle"], runs["air"])      # the very same kernel
    # Cessna172Xv2: wave-pair against one-wave kernel, closed loop
    gains = fb.ctl_gains.ctl_gains_blob()
    xs = {}
    for duo in ("1", "0"):
        monkeypatch.setenv("FLIGHTBATCH_DUO", duo)
        w = fb.Cessna172Xv2World(n, gains=gains)
        w.env = env6
        sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
        fb.init(sim, tp)
        w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 1.0
        w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = 0.2
        fb.step(sim, 5.0); w.sync()
        ok = w.trim_success & (w.status == 0)
        xs[duo] = (w.x, ok)
        w.close()
    monkeypatch.delenv("FLIGHTBATCH_DUO")
    assert np.array_equal(xs["1"][1], xs["0"][1]) and xs["1"][1].mean() > 0.99
    d = np.abs(xs["1"][0] - xs["0"][0])[:, xs["1"][1]] / np.maximum(np.abs(xs["0"][0][:, xs["1"][1]]), 1.0)
    print("Xv2 with per-aircraft rows, wave-pair vs one-wave kernel after 500 closed-loop steps: %.2e" % d.max())
    assert d.max() < 1e-8


def test_sea_level_rows_saturate_like_the_references_ranged_inputs(fb, oracle):
    """TunableSeaLevelU holds T and p as Ranged values: an assignment saturates to [T_std - 50, T_std + 50] K and [p_std - 10000, p_std + 10000] Pa
    (FP/atmosphere.jl:69-77). Rows outside that range are brought to the bound by fb_set_env (fb_get_env returns what is in force) and by the
    oracle alike — the same trajectory on both sides — non-finite values are refused, and the batch-wide block saturates the same way."""
    K = fb.K
    n = 512
    tp = lattice_trim_params(fb, n, seed=63)
    env6 = random_env(fb, n, 9, h_trn=0.0)
    env6[K["FB_ENV_T_SL"], ::4] = 150.0; env6[K["FB_ENV_T_SL"], 1::4] = 400.0      # far outside [238.15, 338.15]
    env6[K["FB_ENV_P_SL"], ::3] = 50000.0; env6[K["FB_ENV_P_SL"], 1::3] = 130000.0  # far outside [91325, 111325]
    w = fb.BatchedWorld(n)
    w.env = env6
    got = w.env
    sat = env6.copy()
    sat[K["FB_ENV_T_SL"]] = np.clip(env6[K["FB_ENV_T_SL"]], 288.15 - 50, 288.15 + 50); sat[K["FB_ENV_P_SL"]] = np.clip(env6[K["FB_ENV_P_SL"]], 101325.0 - 1e4, 101325.0 + 1e4)
    assert np.array_equal(got, sat) and not np.array_equal(got, env6)
    fb.f_init(w, tp)
    ok = w.trim_success
    assert ok.mean() > 0.9
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 5.0); w.sync()
    with oracle.per_aircraft_env():
        xo, so, sto = oracle.step(x0, u0, ui0, s0, oracle.env_rows(env6), 0.01, 500)     # the oracle is handed the UNSATURATED rows
    assert np.array_equal(w.status, sto)
    err = (np.abs(w.x - xo) / np.maximum(np.abs(xo), 1.0))[:, sto == 0]
    print("saturated sea-level rows, max scaled error after 500 steps: %.3e" % err.max())
    assert err.max() < 1e-6
    for row, val in ((K["FB_ENV_WIND_E"], np.inf), (K["FB_ENV_T_SL"], np.nan), (K["FB_ENV_H_TERRAIN"], -np.inf)):
        bad = env6.copy(); bad[row, 7] = val
        with pytest.raises(fb.FlightBatchError, match="non-finite"):
            w.env = bad
    assert np.array_equal(w.env, sat), "a refused call leaves the rows as they were"
    w.env = None
    w.set_params(T_sl=500.0, p_sl=10.0)
    assert w.params.T_sl == 288.15 + 50 and w.params.p_sl == 101325.0 - 1e4
    w.close()


@pytest.mark.parametrize("rows_in_checkpoint", [True, False])
def test_checkpoint_round_trip_with_environment_rows(fb, rows_in_checkpoint):
    """checkpoint -> np.savez -> restore: a checkpoint of a world WITH rows carries them (the resumed trajectory is the uninterrupted one, bit for
    bit); a checkpoint of a world WITHOUT rows says so and clears the rows of the world it is restored into; one that says nothing about rows
    (written before they existed) leaves the world's rows alone. Whether a handle has rows is asked of the handle (fb_has_env)."""
    import io
    n = 1024
    tp = lattice_trim_params(fb, n, seed=64)
    env6 = random_env(fb, n, 11, h_trn=0.0)
    w = fb.BatchedWorld(n)
    if rows_in_checkpoint:
        w.env = env6
    assert w.has_env == rows_in_checkpoint
    fb.f_init(w, tp)
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 2.0); w.sync()
    buf = io.BytesIO(); np.savez(buf, **fb.checkpoint(sim)); buf.seek(0)
    fb.step(sim, 3.0); w.sync()
    x_ref, st_ref = w.x, w.status
    w.close()
    ck = dict(np.load(buf))
    assert bool(ck["has_env"]) == rows_in_checkpoint and ("env" in ck) == rows_in_checkpoint
    w2 = fb.BatchedWorld(n)
    w2.env = random_env(fb, n, 12, h_trn=0.0)          # whatever the target world had before
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
    fb.restore(sim2, ck)
    assert w2.has_env == rows_in_checkpoint
    if rows_in_checkpoint:
        assert np.array_equal(w2.env, env6)
    fb.step(sim2, 3.0); w2.sync()
    assert np.array_equal(w2.x, x_ref) and np.array_equal(w2.status, st_ref)
    # a checkpoint that says nothing about rows (older format) leaves the world's rows alone
    old = {k: v for k, v in ck.items() if k not in ("has_env", "env")}
    w2.env = env6
    fb.restore(sim2, old)
    assert w2.has_env and np.array_equal(w2.env, env6)
    w2.close()
