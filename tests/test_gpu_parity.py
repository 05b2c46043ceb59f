"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances. The path is IEEE fp64 on both sides; the GPU build contracts a*b+c into FMA, uses ROCm's
libm (ocml) instead of glibc's, and evaluates a few sub-expressions in a different but algebraically
identical arrangement (see csrc/c172_device.hpp header). Single-RHS agreement is therefore a few ulp of
the dominant term, asserted as 1e-9 of a per-field scale; trajectories are asserted at the north star's
1e-6 relative (per-field floors as in SURVEY.md §8d) and the observed value is far tighter.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# per-state scale floors: quaternion components 1, rates 1e-3 rad/s
def state_scale(x):
    sc = np.maximum(np.abs(x), 1e-3)
    sc[12:20] = 1.0               # q_wb, q_ew
    sc[0:2] = np.maximum(np.abs(x[0:2]), 1e-2)   # alpha/beta filt
    sc[2:8] = 1.0                 # contact regulators (zero airborne)
    sc[10:12] = 1.0               # engine PI states
    sc[20] = np.maximum(np.abs(x[20]), 1.0)
    sc[24:27] = np.maximum(np.abs(x[24:27]), 1.0)
    return sc


def lattice_trim_params(fb, n, seed=172):
    rng = np.random.default_rng(seed)
    lat = rng.uniform(-1.2, 1.2, n); lon = rng.uniform(-np.pi, np.pi, n)
    n_e = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    return fb.TrimParameters(n_e=n_e, h_e=rng.uniform(200.0, 3000.0, n), EAS=rng.uniform(35.0, 55.0, n),
                             ψ_nb=rng.uniform(-np.pi, np.pi, n), γ_wb_n=rng.uniform(-0.02, 0.02, n),
                             ψ_wb_dot=rng.uniform(-0.03, 0.03, n), flaps=rng.choice([0.0, 0.0, 0.33], n),
                             fuel_load=rng.uniform(0.1, 1.0, n))


def test_trim_default_matches_oracle(fb, oracle):
    """f_init!(world, C172.TrimParameters()): reference test lib/FlightApps/test/c172/test_c172s.jl:22-38
    (trim must succeed from the default guess); device solution equals the oracle's."""
    w = fb.BatchedWorld(64)
    fb.f_init(w, fb.TrimParameters())
    assert w.trim_success.all()
    assert (w.trim_cost <= 1e-16).all()
    ref = oracle.trim(fb.TrimParameters().pack(64), fb.TrimState(64), oracle.default_env())
    assert ref["ok"].all()
    assert np.max(np.abs(w.trim_state - ref["ts"])) < 1e-9
    assert np.max(np.abs(w.x - ref["x"]) / state_scale(ref["x"])) < 1e-9
    assert (w.s == ref["s"]).all() and (w.ui == ref["ui"]).all()
    assert np.max(np.abs(w.u - ref["u"])) < 1e-9
    w.close()


def test_trim_lattice_matches_oracle(fb, oracle):
    """randomised TrimParameters (location, altitude, EAS, heading, flight-path angle, turn rate, flaps, fuel): the device
    solver and the oracle's must make the same success / failure call on every aircraft and land on the same trim."""
    n = 2048
    tp = lattice_trim_params(fb, n)
    w = fb.BatchedWorld(n)
    fb.f_init(w, tp)
    ref = oracle.trim(tp.pack(n), fb.TrimState(n), oracle.default_env())
    assert (ref["ok"] == w.trim_success).all(), f"{(ref['ok'] != w.trim_success).sum()} aircraft with different success flags"
    ok = ref["ok"]
    assert ok.mean() > 0.99          # the few that fail ask for a climb at high EAS and altitude with flaps out: no power left
    assert (w.trim_cost[ok] <= 1e-16).all() and (w.trim_cost[~ok] > 1e-16).all()
    assert np.max(np.abs(w.trim_state[:, ok] - ref["ts"][:, ok])) < 1e-9
    assert np.max(np.abs(w.x[:, ok] - ref["x"][:, ok]) / state_scale(ref["x"][:, ok])) < 1e-9
    w.close()


def test_trim_bench_lattice_all_succeed(fb, oracle):
    """The 32 x 32 (EAS, h) cells of bench.py's config-3 lattice at 16 headings: every aircraft has a trim (throttle <= 0.88)
    and the solver must find it from TrimState() — including the band EAS 50.5 m/s, h = 1735 ... 2187 m, throttle ~0.70,
    where round 1's clamped Newton ran into the throttle = 1 corner."""
    i, j, k = np.meshgrid(np.arange(32), np.arange(32), np.arange(16), indexing="ij")
    n = i.size
    EAS = (35.0 + 20.0 * i / 31.0).ravel(); h = (200.0 + 2800.0 * j / 31.0).ravel(); psi = (-np.pi + 2 * np.pi * k / 16.0).ravel()
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    assert w.trim_success.all(), f"{(~w.trim_success).sum()} of {n} aircraft failed to trim"
    assert w.trim_cost.max() < 1e-20
    ts = w.trim_state.reshape(7, 32, 32, 16)
    band = ts[:, 24, 17:23, :]
    assert (band[3] > 0.66).all() and (band[3] < 0.75).all() and (band[2] > 0.89).all() and (band[2] < 0.95).all()
    assert ts[3].max() < 0.9 and ts[2].max() < 1.05
    sel = np.arange(0, n, 37)
    ref = oracle.trim(fb.TrimParameters(EAS=EAS[sel], h_e=h[sel], ψ_nb=psi[sel]).pack(len(sel)), fb.TrimState(len(sel)), oracle.default_env())
    assert ref["ok"].all()
    assert np.max(np.abs(w.trim_state[:, sel] - ref["ts"])) < 1e-9
    w.close()


def test_trim_theta_constraint_gives_the_requested_flight_path_angle(fb):
    """FPt/test_aircraft_base.jl:15-41 on the device: at every trimmed state of a batch with random flight-path angles, sideslip and turn rates
    (bank angles up to ~0.4 rad) the ground flight-path angle of f_ode!'s output record equals the requested γ_wb_n (still air)."""
    n = 4096
    rng = np.random.default_rng(12)
    tp = fb.TrimParameters(h_e=rng.uniform(500, 2500, n), EAS=rng.uniform(40, 52, n), ψ_nb=rng.uniform(-3, 3, n), γ_wb_n=rng.uniform(-0.07, 0.05, n),
                           ψ_wb_dot=rng.uniform(-0.08, 0.08, n), β_a=rng.uniform(-0.05, 0.05, n))
    for kin in ("WA", "NED"):
        w = fb.BatchedWorld(n, kinematics=kin)
        fb.f_init(w, tp)
        ok = w.trim_success
        assert ok.mean() > 0.9 and np.abs(w.trim_state[1][ok]).max() > 0.3
        fb.f_ode(w)
        assert np.abs(w.y[fb.K["FB_Y_KIN"] + 39] - np.asarray(tp.γ_wb_n))[ok].max() < 1e-12
        w.close()


def test_trim_batch_sizes_around_the_wave_size(fb):
    """k_trim runs one wave per 64 aircraft up to one per SIMD and hands aircraft out from a queue: batches of 1, 2, 63, 64, 65, 127 and 1000
    copies of TrimParameters() all trim, every copy to the same bits."""
    ref = None
    for n in (1, 2, 63, 64, 65, 127, 1000):
        w = fb.BatchedWorld(n)
        fb.f_init(w, fb.TrimParameters())
        ts = w.trim_state.copy()
        assert w.trim_success.all(), n
        if ref is None:
            ref = ts[:, 0].copy()
        assert np.array_equal(ts, np.tile(ref[:, None], (1, n))), n
        w.close()


def test_trim_does_not_depend_on_which_lane_takes_an_aircraft(fb):
    """k_trim is persistent: a lane takes the next aircraft from a queue whenever its own has converged, so which lane and wave trims
    an aircraft, and beside whom, depends on the batch order and on timing. What it computes for an aircraft must not: the same
    aircraft in another order (and in a batch of another size, through the continuation fallback as well — the wide envelope has
    points without a trim) give the same trim state, cost and initial condition BIT FOR BIT."""
    n = 8192
    rng = np.random.default_rng(5)
    lat = rng.uniform(-1.2, 1.2, n); lon = rng.uniform(-np.pi, np.pi, n)
    f = dict(n_e=np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]), h_e=rng.uniform(100.0, 4500.0, n),
             EAS=rng.uniform(24.0, 62.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n), γ_wb_n=rng.uniform(-0.05, 0.05, n),
             ψ_wb_dot=rng.uniform(-0.03, 0.03, n), flaps=rng.choice([0.0, 0.33, 1.0], n), fuel_load=rng.uniform(0.1, 1.0, n))
    def run(idx):
        w = fb.BatchedWorld(len(idx))
        fb.f_init(w, fb.TrimParameters(**{k: (v[:, idx] if v.ndim == 2 else v[idx]) for k, v in f.items()}))
        out = (w.trim_state.copy(), w.trim_cost.copy(), w.trim_success.copy(), w.x.copy(), w.u.copy())
        w.close()
        return out
    ident = np.arange(n)
    perm = rng.permutation(n)
    part = np.sort(rng.choice(n, 1000, replace=False))
    a = run(ident); b = run(perm); c = run(part)
    assert 0.02 < 1 - a[2].mean() < 0.6, "the envelope should hold aircraft without a trim (the continuation's path) and with one"
    for x, y, z in zip(a, b, c):
        xa = x[..., perm] if x.ndim == 2 else x[perm]
        assert np.array_equal(xa, y, equal_nan=True)
        xp = x[..., part] if x.ndim == 2 else x[part]
        assert np.array_equal(xp, z, equal_nan=True)


def test_trim_wide_envelope_and_hard_cases(fb, oracle):
    """An envelope wider than the aircraft can fly (28 % of the points have no trim: below the stall speed, beyond the power
    available) plus the five knot-trapped cases of tests/golden/trim_hard_cases.npz: same success set as the oracle, same
    trims; failed aircraft stay inside the bounds of c172.jl:901-917."""
    import os
    n = 8192
    rng = np.random.default_rng(23)
    lat = rng.uniform(-1.4, 1.4, n); lon = rng.uniform(-np.pi, np.pi, n)
    tp = fb.TrimParameters(n_e=np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]),
                           h_e=rng.uniform(150.0, 4000.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n), EAS=rng.uniform(28.0, 62.0, n),
                           γ_wb_n=rng.uniform(-0.08, 0.06, n), ψ_wb_dot=rng.uniform(-0.1, 0.1, n), θ_wb_dot=rng.uniform(-0.02, 0.02, n),
                           β_a=rng.uniform(-0.1, 0.1, n), fuel_load=rng.uniform(0.05, 1.0, n), flaps=rng.choice([0.0, 0.0, 0.33, 0.66, 1.0], n),
                           payload=rng.uniform(0.0, 90.0, (5, n)))
    packed = tp.pack(n)
    hard = np.load(os.path.join(os.path.dirname(__file__), "golden", "trim_hard_cases.npz"))["trim_params"]
    packed[:, : hard.shape[1]] = hard
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(n_e=packed[0:3], h_e=packed[3], ψ_nb=packed[4], EAS=packed[5], γ_wb_n=packed[6], ψ_wb_dot=packed[7],
                                   θ_wb_dot=packed[8], β_a=packed[9], fuel_load=packed[10], mixture=packed[11], flaps=packed[12], payload=packed[13:18]))
    ref = oracle.trim(packed, fb.TrimState(n), oracle.default_env())
    assert w.trim_success[: hard.shape[1]].all() and ref["ok"][: hard.shape[1]].all()
    agree = ref["ok"] == w.trim_success
    assert agree.mean() >= 0.9995, f"{(~agree).sum()} aircraft with different success flags"
    assert 0.6 < ref["ok"].mean() < 0.8
    both = ref["ok"] & w.trim_success
    assert np.max(np.abs(w.trim_state[:, both] - ref["ts"][:, both])) < 1e-8
    lo = np.array([-np.pi / 12, -np.pi / 3, 0.4, 0, -1, -1, -1])[:, None]; hi = np.array([0.36, np.pi / 3, 1.1, 1, 1, 1, 1])[:, None]
    assert (w.trim_state >= lo - 1e-15).all() and (w.trim_state <= hi + 1e-15).all()
    w.close()


def test_nonlevel_trims_are_cross_checked_by_the_other_sides_cost(fb, oracle):
    """The reference's stored trims (tests/test_reference_trim_points.py) pin the solver at 28 wings-level conditions; turning, climbing and
    sideslipping trims (C172.TrimParameters ψ_wb_dot, θ_wb_dot, γ_wb_n, β_a: FA/c172/c172.jl:806-818) have no reference solution to be held to.
    There the two sides check EACH OTHER: the trim found on the device, evaluated by the oracle's own cost function (c172.jl:857-867: the squared
    residuals of v̇_b / |v|, ω̇_b, the engine's ω̇ / ω_rated), and the trim found by the oracle, evaluated by the device's f_ode! — both must be
    solutions (<= 1e-15) in the OTHER side's model, on a grid of 432 non-level conditions (about four in five inside the envelope)."""
    import ctypes as C
    import itertools
    K = fb.K
    grid = list(itertools.product((38.0, 46.0, 54.0), (400.0, 2400.0), (-0.06, 0.0, 0.1), (0.0, 0.012), (-0.08, 0.0, 0.06), (-0.08, 0.0, 0.08), (0.0, 0.5)))
    n = len(grid)
    g = np.array(grid).T
    tp = fb.TrimParameters(EAS=g[0], h_e=g[1], ψ_wb_dot=g[2], θ_wb_dot=g[3], γ_wb_n=g[4], β_a=g[5], flaps=g[6], ψ_nb=np.linspace(-3, 3, n))
    w = fb.BatchedWorld(n)
    fb.f_init(w, tp)
    env = oracle.default_env()
    ref = oracle.trim(tp.pack(n), fb.TrimState(n), env)
    ok = w.trim_success
    assert np.array_equal(ok, ref["ok"]) and ok.mean() > 0.7, (ok.mean(), (ok != ref["ok"]).sum())   # (the grid reaches beyond the envelope: both sides must agree on where)
    assert (w.trim_cost[ok] <= 1e-16).all() and (ref["cost"][ok] <= 1e-16).all()
    # (1) the device's solutions in the oracle's cost
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    tpp = tp.pack(n)
    c_orc = np.array([oracle.lib.fo_c172_trim_cost(dp(np.ascontiguousarray(tpp[:, i])), dp(np.ascontiguousarray(w.trim_state[:, i])), dp(env)) for i in range(n)])
    # (2) the oracle's solutions in the device's model: its trimmed state, inputs and discrete states handed to fb_f_ode
    def device_cost(x, s, u, ui):
        w.set_state(x, s); w.u = u; w.ui = ui
        xd = np.zeros((K["FB_NX"], n)); fb.f_ode(w, xd)
        vd, wd, ed = xd[K["FB_X_V_EB_B"]:K["FB_X_V_EB_B"] + 3], xd[K["FB_X_OMEGA_EB_B"]:K["FB_X_OMEGA_EB_B"] + 3], xd[K["FB_X_ENG_OMEGA"]]
        nv = np.sqrt((x[K["FB_X_V_EB_B"]:K["FB_X_V_EB_B"] + 3] ** 2).sum(0))
        return ((vd / nv) ** 2).sum(0) + (wd ** 2).sum(0) + (ed / (2700 * np.pi / 30)) ** 2     # ω_rated: 2700 rpm (FP/piston.jl:203)
    x_dev, s_dev, u_dev, ui_dev = w.x, w.s, w.u, w.ui
    c_self = device_cost(x_dev, s_dev, u_dev, ui_dev)
    c_dev = device_cost(ref["x"], ref["s"], ref["u"], ref["ui"])
    print("non-level trims (%d of %d trimmable): device solution in the oracle's cost <= %.2e; oracle solution in the device's model <= %.2e; device in its own <= %.2e; "
          "max |Δ trim state| %.2e" % (ok.sum(), n, c_orc[ok].max(), c_dev[ok].max(), c_self[ok].max(), np.abs(w.trim_state - ref["ts"])[:, ok].max()))
    assert c_orc[ok].max() <= 1e-15 and c_dev[ok].max() <= 1e-15 and c_self[ok].max() <= 1e-15
    assert np.abs(w.trim_state - ref["ts"])[:, ok].max() < 1e-7
    w.close()


def test_f_ode_matches_oracle(fb, oracle):
    """Single RHS: xdot and the full 174-double output record at trimmed and perturbed states."""
    n = 4096
    tp = lattice_trim_params(fb, n, seed=7)
    w = fb.BatchedWorld(n)
    w.set_params(wind_ned=(3.0, -2.0, 0.5), T_sl=293.15, p_sl=100500.0)
    env = oracle.default_env(T_sl=293.15, p_sl=100500.0, wind=(3.0, -2.0, 0.5))
    fb.f_init(w, tp)
    rng = np.random.default_rng(1)
    x = w.x
    x[21:24] += rng.normal(0, 0.05, (3, n))      # body rates
    x[24:27] += rng.normal(0, 2.0, (3, n))       # velocity
    x[0:2] += rng.normal(0, 0.01, (2, n))
    x[9] *= rng.uniform(0.8, 1.1, n)
    s = w.s
    s[0] = rng.integers(0, 2, n)                 # stall flag both ways
    w.set_state(x, s)
    u = w.u
    u[2:5] += rng.normal(0, 0.1, (3, n))
    u[0] = rng.uniform(0, 1, n)
    w.u = u
    xd = np.zeros((27, n))
    fb.f_ode(w, xd)
    y = w.y
    xdo, yo, sto = oracle.f_ode(x, u, w.ui, s, env)
    assert (w.status == 0).all() and (sto == 0).all()
    sc_xd = np.maximum(np.abs(xdo), np.array([1.0] * 2 + [1.0] * 6 + [1e-4] + [10.0, 1, 1] + [1e-2] * 8 + [1.0] + [1.0] * 3 + [5.0] * 3)[:, None])
    err = np.abs(xd - xdo) / sc_xd
    assert err.max() < 1e-9, f"xdot mismatch {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    sc_y = np.maximum(np.abs(yo), 1.0)
    sc_y[22:25] = 6.4e6          # r_eb_e
    erry = np.abs(y - yo) / sc_y
    assert erry.max() < 1e-9, f"y mismatch {erry.max()} at {np.unravel_index(erry.argmax(), erry.shape)}"
    w.close()


def test_step_trajectory_matches_oracle(fb, oracle):
    """1000 RK4 steps at dt = 0.01 (config-3 style randomised trims, perturbed so that the dynamics are
    exercised): final state and decimated trajectory within 1e-6 (north star); observed error reported."""
    n = 4096
    tp = lattice_trim_params(fb, n, seed=11)
    w = fb.BatchedWorld(n)
    fb.f_init(w, tp)
    rng = np.random.default_rng(2)
    x = w.x
    x[21:24] += rng.normal(0, 0.02, (3, n))
    x[24:27] += rng.normal(0, 1.0, (3, n))
    w.set_state(x, w.s)
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=0.01, t_end=10.0, save_on=True, saveat=1.0)
    fb.init(sim)
    fb.step(sim, 10.0)
    ts = fb.TimeSeries(sim)
    xo, so, sto, traj = oracle.step(x0, u0, ui0, s0, oracle.default_env(), 0.01, 1000, save_every=100)
    assert len(ts) == 11 and traj.shape[0] == 11
    assert (w.status == 0).all() and (sto == 0).all()
    assert (w.s == so).all()
    errs = [np.max(np.abs(ts.x[k] - traj[k]) / state_scale(traj[k])) for k in range(11)]
    print("trajectory max scaled error per saved sample:", ["%.2e" % e for e in errs])
    assert max(errs) < 1e-6
    w.close()


def test_globally_scattered_batch_matches_oracle(fb, oracle):
    """A fleet spread over the whole Earth (bench.py's batch sits at ϕ = λ = 0: one EGM96 cell): 1000 RK4 steps against the oracle on
    aircraft placed uniformly over the sphere, within 1° and within 300 m of either pole (|ϕ| > 89°: the geoid grid's first / last row,
    `Line` extrapolation beyond it; WA kinematics carry an aircraft across the pole), astride the antimeridian and astride λ = 0 (the
    λ ∈ [0, 2π) wrap of the geoid lookup, FP/geodesy.jl:186-211) close enough to cross them during the run, every one on its own heading."""
    rng = np.random.default_rng(41)
    groups = []
    m = 2048; groups.append((np.arcsin(rng.uniform(-1, 1, m)), rng.uniform(-np.pi, np.pi, m)))              # uniform over the sphere
    for sgn in (1.0, -1.0):
        m = 256; groups.append((sgn * (np.pi / 2 - np.deg2rad(rng.uniform(0.0, 1.0, m))), rng.uniform(-np.pi, np.pi, m)))       # within 1° of a pole
        m = 256; groups.append((sgn * (np.pi / 2 - rng.uniform(0.0, 300.0, m) / 6.36e6), rng.uniform(-np.pi, np.pi, m)))        # within 300 m of it
    m = 512; groups.append((rng.uniform(-1.4, 1.4, m), np.pi * rng.choice([-1.0, 1.0], m) * (1 - rng.uniform(0, 2e-5, m) / np.pi)))     # antimeridian ± 130 m
    m = 512; groups.append((rng.uniform(-1.4, 1.4, m), rng.uniform(-2e-5, 2e-5, m)))                                                  # λ = 0 ± 130 m
    lat = np.concatenate([g[0] for g in groups]); lon = np.concatenate([g[1] for g in groups])
    n = lat.size
    assert n == 4096 and (np.abs(lat) > np.deg2rad(89.0)).sum() >= 1024
    n_e = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(n_e=n_e, h_e=rng.uniform(200.0, 3000.0, n), EAS=rng.uniform(35.0, 55.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n)))
    assert w.trim_success.all()
    x = w.x
    x[21:24] += rng.normal(0, 0.02, (3, n)); x[24:27] += rng.normal(0, 1.0, (3, n))
    w.set_state(x, w.s)
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    fb.f_ode(w); y0 = w.y
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 10.0); w.sync()
    fb.f_ode(w); y1 = w.y
    K = fb.K
    la0, lo0, la1, lo1 = y0[K["FB_Y_KIN"] + 15], y0[K["FB_Y_KIN"] + 16], y1[K["FB_Y_KIN"] + 15], y1[K["FB_Y_KIN"] + 16]
    crossed_anti = int(((np.abs(lo0) > 3.0) & (np.sign(lo0) != np.sign(lo1))).sum())
    crossed_zero = int(((np.abs(lo0) < 0.1) & (np.sign(lo0) != np.sign(lo1))).sum())
    over_pole = int(((np.abs(la0) > 1.57) & (np.abs(np.angle(np.exp(1j * (lo1 - lo0)))) > np.pi / 2)).sum())
    print(f"crossed the antimeridian: {crossed_anti}, crossed λ = 0: {crossed_zero}, passed within sight of a pole (longitude swung by > 90°): {over_pole}")
    assert crossed_anti > 50 and crossed_zero > 50 and over_pole > 20
    xo, so, sto = oracle.step(x0, u0, ui0, s0, oracle.default_env(), 0.01, 1000)
    assert (w.status == 0).all() and (sto == 0).all() and np.array_equal(w.s, so)
    err = np.abs(w.x - xo) / state_scale(xo)
    print("globally scattered batch: max scaled error after 1000 steps: %.3e" % err.max())
    assert err.max() < 1e-6
    yo = oracle.f_ode(xo, u0, ui0, so, oracle.default_env())[1]
    h_o = K["FB_Y_KIN"] + 21    # orthometric altitude: the geoid lookup's output (h_e − N)
    assert np.abs(y1[h_o] - yo[h_o]).max() < 1e-6
    w.close()


def test_steps_per_launch_invariance(fb):
    """Fusing k steps per launch must not change the result (bit-for-bit): idempotence of the launch split."""
    n = 1024
    w = fb.BatchedWorld(n)
    fb.f_init(w, lattice_trim_params(fb, n, seed=3))
    x0, s0 = w.x, w.s
    outs = []
    for k in (1, 7, 50):
        w.set_state(x0, s0)
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
        fb.step(sim, 1.0)
        w.sync()
        outs.append(w.x)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    w.close()


def test_f_step_matches_oracle(fb, oracle):
    """f_step!(world): quaternion renormalisation (reference forces it the same way,
    lib/FlightPhysics/test/test_kinematics.jl:26-30), stall hysteresis, engine state machine, regulator reset."""
    n = 512
    w = fb.BatchedWorld(n)
    fb.f_init(w, lattice_trim_params(fb, n, seed=5))
    rng = np.random.default_rng(4)
    x = w.x; s = w.s; ui = w.ui
    x[12] = np.where(rng.random(n) < 0.5, 3.0, x[12])          # q_wb[1] = 3 forces renormalisation
    x[16:20] *= (1 + 1e-7 * rng.random(n))                      # norm drift > 1e-8 on some
    x[2:8] = rng.normal(0, 1, (6, n))                           # airborne -> regulators must be reset to 0
    x[24] = np.where(rng.random(n) < 0.3, 12.0, x[24]); x[26] = np.where(rng.random(n) < 0.3, 8.0, x[26])  # high alpha -> stall
    s[0] = rng.integers(0, 2, n)
    s[1] = rng.integers(0, 3, n)
    x[9] = rng.uniform(10, 300, n)                              # engine speeds around stall / idle thresholds
    x[8] = np.where(rng.random(n) < 0.2, -0.01, x[8])           # no fuel available
    ui = ui | np.where(rng.random(n) < 0.5, 1, 0).astype(np.int32) | np.where(rng.random(n) < 0.2, 2, 0).astype(np.int32)
    w.set_state(x, s); w.ui = ui
    fb.f_step(w)
    w.sync()
    xo, so, sto = oracle.f_step(x, w.u, ui, s, oracle.default_env())
    assert (w.s == so).all()
    assert np.max(np.abs(w.x - xo) / state_scale(xo)) < 1e-14
    w.close()


def test_ground_contact_matches_oracle(fb, oracle):
    """Aircraft sitting on / dropping onto the runway: the full landing-gear branch (reference exercises it
    in lib/FlightApps/test/c172/test_c172s.jl:52-72 with h = terrain + 1.8 m)."""
    n = 256
    w = fb.BatchedWorld(n)
    rng = np.random.default_rng(9)
    x = np.zeros((27, n))
    x[8] = 0.5
    th = rng.uniform(-0.03, 0.06, n); ph = rng.uniform(-0.03, 0.03, n); ps = rng.uniform(-np.pi, np.pi, n)
    def qmul(a, b):
        return np.stack([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
    z = np.zeros(n)
    q = qmul(qmul(np.stack([np.cos(ps/2), z, z, np.sin(ps/2)]), np.stack([np.cos(th/2), z, np.sin(th/2), z])),
             np.stack([np.cos(ph/2), np.sin(ph/2), z, z]))
    x[12:16] = q
    lat, lon = 0.7, -0.3
    # q_ew = Rz(lon) ∘ Ry(-(lat + π/2))
    a = -(lat + np.pi / 2)
    qe = qmul(np.array([np.cos(lon/2), 0, 0, np.sin(lon/2)])[:, None] * np.ones(n), np.array([np.cos(a/2), 0, np.sin(a/2), 0])[:, None] * np.ones(n))
    x[16:20] = qe
    n_e = np.array([np.cos(lat)*np.cos(lon), np.cos(lat)*np.sin(lon), np.sin(lat)])
    geoid = oracle.lib.fo_geoid_height(np.ascontiguousarray(n_e).ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
    x[20] = 0.0 + geoid + rng.uniform(1.70, 1.95, n)     # terrain at 0 m orthometric; gear legs ~1.9 m long
    x[21:24] = rng.normal(0, 0.02, (3, n))
    x[24] = rng.uniform(0, 15, n); x[25] = rng.normal(0, 0.3, n); x[26] = rng.uniform(-0.2, 0.6, n)
    x[2:8] = rng.normal(0, 0.2, (6, n))
    x[9] = 100.0
    s = np.zeros((2, n), np.int32); s[1] = 2
    u = np.zeros((16, n)); u[11:16] = np.array([75, 75, 0, 0, 50.0])[:, None]; u[0] = 0.2; u[1] = 0.5
    u[9] = rng.uniform(0, 1, n); u[10] = rng.uniform(0, 1, n); u[4] = rng.uniform(-1, 1, n)
    ui = np.full(n, 4 | 8, np.int32)
    ui[::3] = 4   # steering disengaged on a third
    w.set_state(x, s); w.u = u; w.ui = ui
    xd = np.zeros((27, n))
    fb.f_ode(w, xd)
    y = w.y
    xdo, yo, sto = oracle.f_ode(x, u, ui, s, oracle.default_env())
    wow = yo[78 + 1] + yo[89 + 1] + yo[100 + 1]
    assert (wow > 0).mean() > 0.5, "test must exercise the contact branch"
    assert (w.status == sto).all()
    sc = np.maximum(np.abs(xdo), 1.0)
    err = np.abs(xd - xdo) / sc
    # contact forces come from strut compression = difference of ECEF positions (6.4e6 m, ulp 1e-9 m) times
    # k_s = 4e4 N/m: they are conditioned to ~1e-4 N, i.e. 1e-7 rad/s² — for the reference itself too.
    assert err.max() < 1e-6, f"xdot mismatch {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    scy = np.maximum(np.abs(yo), 1.0); scy[22:25] = 6.4e6
    # a wheel within rounding of the surface (|ξ| ~ 1e-10 m) may read wow on one side only; its force is then
    # ~1e-5 N either way, so forces/torques are compared on a 100 N / 100 N·m floor
    for g in range(3):
        k0 = 78 + 11 * g
        scy[k0 + 4: k0 + 11] = np.maximum(scy[k0 + 4: k0 + 11], 100.0)
        borderline = np.abs(yo[k0]) < 1e-6            # |Δh| within rounding of the surface: wow may differ
        scy[k0 + 1: k0 + 11, borderline] = np.inf
    scy[134 + 13: 134 + 19] = np.maximum(scy[134 + 13: 134 + 19], 100.0)
    erry = np.abs(y - yo) / scy
    assert erry.max() < 1e-6, f"y mismatch {erry.max()} at {np.unravel_index(erry.argmax(), erry.shape)}"
    # and a short roll-out on the ground
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=10)
    fb.step(sim, 0.5); w.sync()
    xo, so, st2 = oracle.step(x, u, ui, s, oracle.default_env(), 0.01, 50)
    # the same aircraft end their simulation (GroundCrash on the hardest touchdowns), with the same status word, and — the oracle stops
    # where the reference stops (FC/sim.jl:561-570) — in the same state: terminated lanes are compared like the rest
    assert np.array_equal(st2, w.status), f"{int((st2 != w.status).sum())} status words differ"
    e2 = np.abs(w.x - xo) / np.maximum(np.abs(xo), 1.0)
    print("ground roll-out max scaled error", e2.max(), "terminated:", int((st2 != 0).sum()))
    assert e2.max() < 1e-6 and np.array_equal(w.s, so)
    w.close()


def test_status_stays_clean_from_trim_to_step(fb):
    """Regression: stepping straight from the device trim (no host set_state in between) must leave every status word 0
    and every aircraft advancing (an earlier build returned garbage status words here, freezing the whole batch)."""
    n = 65536
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 48, n), ψ_nb=np.linspace(-3, 3, n)))
    assert w.trim_success.all() and (w.status == 0).all()
    x0 = w.x
    for k in (1, 50):
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
        fb.step(sim, 0.5); w.sync()
        assert (w.status == 0).all()
    x1 = w.x
    assert np.all(x1[8] < x0[8]), "fuel must have burnt on every aircraft: nobody is frozen"
    w.close()


def test_device_log_matches_host_snapshots(fb):
    """The on-device TimeSeries log (cb_save, FC/sim.jl:210-217,345-347): decimated y rows + x saved on the GPU must
    equal what f_ode!/mdl.y give when the same run is stopped at the same instants; t0 sample from init!."""
    n = 2048
    tp = lattice_trim_params(fb, n, seed=4)
    rows = ["FB_Y_KIN", fb.K["FB_Y_KIN"] + 7, fb.K["FB_Y_AIR"] + 3, fb.K["FB_Y_DYN"] + 20, fb.K["FB_Y_PWP"] + 4]
    w = fb.BatchedWorld(n)
    fb.f_init(w, tp)
    x0, s0 = w.x, w.s
    sim = fb.Simulation(w, dt=0.01, t_end=3.0, saveat=0.25, save_rows=rows, steps_per_launch=40)   # 40 does not divide 25
    fb.init(sim)
    fb.run(sim)
    ts = fb.TimeSeries(sim)
    assert len(ts) == 13 and np.allclose(ts.t, np.arange(13) * 0.25)
    assert ts.x.shape == (13, 27, n) and ts.y.shape == (13, 5, n)
    assert np.array_equal(ts.x[-1], w.x)
    # the same run, stopped by hand
    w2 = fb.BatchedWorld(n)
    fb.f_init(w2, tp); w2.set_state(x0, s0)
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=25)
    ridx = [fb.K[r] if isinstance(r, str) else r for r in rows]
    for k in range(13):
        if k:
            fb.step(sim2, 0.25)
        fb.f_ode(w2)
        assert np.array_equal(ts.x[k], w2.x), k
        assert np.array_equal(ts.y[k], w2.y[ridx]), k
    # capacity is enforced loudly, and init! restarts the log
    with pytest.raises(fb.FlightBatchError, match="capacity"):
        fb.step(sim, 10.0)
    fb.init(sim)
    assert len(fb.TimeSeries(sim)) == 1
    w.close(); w2.close()


@pytest.mark.parametrize("kin", ["ECEF", "NED"])
def test_ecef_and_ned_mechanisations(fb, oracle, kin):
    """Cessna172Sv0(ECEF()) / (NED()) (FP/kinematics.jl:250-425): f_ode! and a 10 s trajectory against the oracle with the same
    mechanisation, and — like the reference's own test (FPt/test_kinematics.jl:75-95) — against the WA run of the same
    aircraft: the three mechanisations must agree on position, attitude and velocity."""
    K = fb.K
    n = 1024
    nk = {"ECEF": 8, "NED": 6}[kin]
    tp = lattice_trim_params(fb, n, seed=31)
    w = fb.BatchedWorld(n, kinematics=kin)
    fb.f_init(w, tp)
    rng = np.random.default_rng(3)
    x = w.x
    assert x.shape[0] == 18 + nk
    x[12 + nk:12 + nk + 3] += rng.normal(0, 0.02, (3, n)); x[12 + nk + 3:] += rng.normal(0, 1.0, (3, n))
    w.set_state(x, w.s)
    # oracle: same mechanisation, 27-row layout with the unused kinematic rows zero
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        xo0 = np.zeros((27, n)); xo0[:12 + nk] = x[:12 + nk]; xo0[21:] = x[12 + nk:]
        xd = np.zeros((18 + nk, n)); fb.f_ode(w, xd)
        xdo, yo, _ = oracle.f_ode(xo0, w.u, w.ui, w.s, oracle.default_env())
        xdo_abi = np.vstack([xdo[:12 + nk], xdo[21:]])
        assert (np.abs(xd - xdo_abi) / np.maximum(np.abs(xdo_abi), 1.0)).max() < 1e-9
        sc_y = np.maximum(np.abs(yo), 1.0); sc_y[22:25] = 6.4e6
        assert (np.abs(w.y - yo) / sc_y).max() < 1e-9
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
        fb.step(sim, 10.0); w.sync()
        xo, so, sto = oracle.step(xo0, w.u, w.ui, np.array([[0] * n, [2] * n], dtype=np.int32), oracle.default_env(), 0.01, 1000)
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    assert (w.status == 0).all() and (sto == 0).all()
    xo_abi = np.vstack([xo[:12 + nk], xo[21:]])
    sc = np.maximum(np.abs(xo_abi), 1.0)
    err = np.abs(w.x - xo_abi) / sc
    print(kin, "vs oracle after 1000 steps:", err.max())
    assert err.max() < 1e-6
    # against the wander-azimuth run of the same aircraft
    w0 = fb.BatchedWorld(n)
    fb.f_init(w0, tp)
    x0 = w0.x; x0[21:] = x[12 + nk:]; w0.set_state(x0, w0.s)
    sim0 = fb.Simulation(w0, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim0, 10.0); w0.sync()
    fb.f_ode(w); fb.f_ode(w0)
    ya, yb = w.y, w0.y
    # (the reference compares the bare kinematics to √eps; here the full aircraft dynamics sit in the loop for 10 s, so the
    # mechanisations' different rounding is fed back and amplified: 1e-5 m, 5e-9, 1e-6 m/s observed)
    for rows, tol in ((slice(K["FB_Y_KIN"] + 22, K["FB_Y_KIN"] + 25), 1e-3),      # r_eb_e [m]
                      (slice(K["FB_Y_KIN"] + 3, K["FB_Y_KIN"] + 7), 1e-7),        # q_nb
                      (slice(K["FB_Y_KIN"] + 34, K["FB_Y_KIN"] + 37), 1e-5)):     # v_eb_n
        d = np.abs(ya[rows] - yb[rows])
        if rows.start == K["FB_Y_KIN"] + 3:
            d = np.minimum(d, np.abs(ya[rows] + yb[rows]))                       # q and -q are the same rotation
        print(kin, "vs WA", rows, d.max())
        assert d.max() < tol, (kin, rows, d.max())
    w.close(); w0.close()


def test_approach_crosses_the_air_ground_handover(fb, oracle):
    """Descending approaches that start above the 10 m clearance limit of the airborne stepping instance and sink through it
    (some down to the runway) inside fused launches: the lanes handed over to the ground-capable instance must give the same
    trajectory as the oracle, and the result must not depend on where the launch boundaries fall."""
    n = 1024
    rng = np.random.default_rng(17)
    h_trn = 300.0
    tp = fb.TrimParameters(EAS=rng.uniform(33, 40, n), h_e=h_trn + rng.uniform(14, 40, n), γ_wb_n=-np.deg2rad(rng.uniform(2, 5, n)),
                           flaps=1.0, ψ_nb=rng.uniform(-3, 3, n))
    env = oracle.default_env(h_trn=h_trn)
    results = []
    for spl in (50, 7):
        w = fb.BatchedWorld(n)
        w.set_params(h_terrain=h_trn)
        fb.f_init(w, tp)
        x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
        ok = w.trim_success
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=spl)
        fb.step(sim, 6.0); w.sync()
        results.append((w.x, w.s, w.status))
        w.close()
    xo, so, sto = oracle.step(x0, u0, ui0, s0, env, 0.01, 600)
    xg, sg, stg = results[0]
    assert ok.mean() > 0.5
    h_o_final = xo[20]   # ellipsoidal; the geoid is ~17 m here, only used to show the regime was crossed
    _, yo, _ = oracle.f_ode(xo, u0, ui0, so, env)
    agl = yo[fb.K["FB_Y_KIN"] + 21] - h_trn
    wow = (yo[fb.K["FB_Y_LDG"] + 1] + yo[fb.K["FB_Y_LDG"] + 12] + yo[fb.K["FB_Y_LDG"] + 23]) > 0
    print("final clearance: min %.2f m, below 10 m: %d, weight on wheels: %d, terminated: %d" % (agl[ok].min(), (agl[ok] < 10).sum(), wow[ok].sum(), (sto[ok] != 0).sum()))
    assert (agl[ok] < 10).sum() > 50 and wow[ok].sum() > 5
    assert np.array_equal(stg[ok], sto[ok])
    live = ok & (sto == 0)
    err = np.abs(xg - xo) / state_scale(xo)
    near = agl < 2.5          # has been (or is about to be) on its wheels: contact forces are conditioned to ~1e-7 (see above)
    print("approach, max scaled error after 600 steps: airborne %.2e, touched down %.2e" % (err[:, live & ~near].max(), err[:, live & near].max()))
    assert err[:, live & ~near].max() < 1e-6 and err[:, live & near].max() < 2e-5 and np.array_equal(sg[:, live], so[:, live])
    # launch boundaries elsewhere: same trajectories to rounding (a lane is stepped by one instance or the other per launch)
    x2, s2, st2 = results[1]
    assert np.array_equal(st2[ok], stg[ok])
    assert (np.abs(x2 - xg) / state_scale(xo))[:, live & ~near].max() < 1e-9


@pytest.mark.parametrize("kin", ["WA", "ECEF", "NED"])
def test_ground_handover_is_reproducible(fb, kin):
    """The same approach-to-touchdown batch twice, bit for bit, for every kinematic mechanisation: the ground-capable stepping
    instances run at the highest register pressure of the library, where a miscompiled spill shows up as run-to-run differences
    (tools/check_isa_spills.py; tests/test_gpu_scenarios.py does the same for Cessna172Xv2)."""
    n = 1024
    rng = np.random.default_rng(29)
    h_trn = 300.0
    tp = fb.TrimParameters(EAS=rng.uniform(33, 40, n), h_e=h_trn + rng.uniform(12, 30, n), γ_wb_n=-np.deg2rad(rng.uniform(2, 5, n)),
                           flaps=1.0, ψ_nb=rng.uniform(-3, 3, n))
    res = []
    for _ in range(2):
        w = fb.BatchedWorld(n, kinematics=kin)
        w.set_params(h_terrain=h_trn)
        fb.f_init(w, tp)
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=25)
        fb.step(sim, 6.0); w.sync()
        fb.f_ode(w)
        y = w.y
        res.append((w.x, w.s, w.status, (y[fb.K["FB_Y_LDG"] + 1] + y[fb.K["FB_Y_LDG"] + 12] + y[fb.K["FB_Y_LDG"] + 23]) > 0))
        w.close()
    assert res[0][3].sum() > 20, "the batch must reach the ground"
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0], res[1]))


def test_f_ode_fuzz_wide_envelope(fb, oracle):
    """f_ode! on 65 536 random states far outside the benchmark's envelope — every attitude, −500 m … 25 km (three ISA layers),
    0 … 120 m/s in any direction, tumbling rates, every engine state, stalled or not, random inputs, wind and a non-standard day —
    against the oracle: ẋ, the output record and the status words."""
    n = 65536
    rng = np.random.default_rng(2026)
    x = np.zeros((27, n))
    x[0] = rng.uniform(-0.3, 0.5, n); x[1] = rng.uniform(-0.3, 0.3, n)
    x[2:8] = rng.normal(0, 0.3, (6, n)) * (rng.random((1, n)) < 0.3)
    x[8] = rng.uniform(0, 1, n)
    x[9] = rng.uniform(0, 320, n); x[10] = rng.uniform(-0.5, 0.5, n); x[11] = rng.uniform(-1, 1, n)
    q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0); x[12:16] = q * (1 + rng.uniform(-1e-8, 1e-8, n))
    qe = rng.normal(size=(4, n)); qe /= np.linalg.norm(qe, axis=0); x[16:20] = qe * (1 + rng.uniform(-1e-8, 1e-8, n))
    x[20] = np.where(rng.random(n) < 0.2, rng.uniform(11000, 25000, n), rng.uniform(-500, 11000, n))
    x[21:24] = rng.normal(0, 0.5, (3, n))
    x[24:27] = rng.normal(0, 1, (3, n)); x[24:27] *= rng.uniform(0, 120, n) / np.linalg.norm(x[24:27], axis=0)
    s = np.stack([rng.integers(0, 2, n), rng.integers(0, 3, n)]).astype(np.int32)
    u = np.zeros((16, n))
    u[0] = rng.uniform(-0.2, 1.2, n); u[1] = rng.uniform(-0.2, 1.2, n); u[2:8] = rng.uniform(-1.3, 1.3, (6, n)); u[8] = rng.uniform(-0.2, 1.2, n)
    u[9:11] = rng.uniform(0, 1, (2, n)); u[11:16] = rng.uniform(-10, 120, (5, n))
    ui = rng.integers(0, 16, n).astype(np.int32)
    env = oracle.default_env(T_sl=300.0, p_sl=99000.0, wind=(12.0, -7.0, 1.5), h_trn=-600.0)   # terrain far below: airborne everywhere
    w = fb.BatchedWorld(n)
    w.set_params(T_sl=300.0, p_sl=99000.0, wind_ned=(12.0, -7.0, 1.5), h_terrain=-600.0)
    w.set_state(x, s); w.u = u; w.ui = ui
    xd = np.zeros((27, n)); fb.f_ode(w, xd)
    y, st = w.y, w.status
    xdo, yo, sto = oracle.f_ode(x, u, ui, s, env)
    assert np.array_equal(st, sto)
    ok = sto == 0
    assert ok.mean() > 0.9
    err = (np.abs(xd - xdo) / np.maximum(np.abs(xdo), 1.0))[:, ok]
    sc_y = np.maximum(np.abs(yo), 1.0); sc_y[22:25] = 6.4e6
    erry = (np.abs(y - yo) / sc_y)[:, ok]
    print("fuzz: max scaled xdot error %.2e (row %d), y error %.2e (row %d)" % (err.max(), err.max(1).argmax(), erry.max(), erry.max(1).argmax()))
    assert err.max() < 1e-8 and erry.max() < 1e-8
    # and the stepping kernels on the same wild states: 20 steps against the oracle, airborne lanes that stay valid
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=20)
    fb.step(sim, 0.2); w.sync()
    xo, so, sto2 = oracle.step(x, u, ui, s, env, 0.01, 20)
    both = (w.status == 0) & (sto2 == 0)
    assert np.array_equal(w.status != 0, sto2 != 0) and both.mean() > 0.85
    e2 = (np.abs(w.x - xo) / state_scale(xo))[:, both]
    print("fuzz: 20 steps, max scaled state error %.2e (row %d)" % (e2.max(), e2.max(1).argmax()))
    assert e2.max() < 1e-6 and np.array_equal(w.s[:, both], so[:, both])
    w.close()


def test_rccl_gather_state_single_rank(fb):
    """fb_comm_* / fb_gather_state: the C-level trajectory collection (RCCL all-gather on the handle's stream). The GPU box has
    one GPU, so this is the world-size-1 communicator: the gathered panel must equal the state. Runs in a child process that
    never imports torch: torch bundles its own HIP / HSA / RCCL runtimes, and the system RCCL this entry point loads must see the
    same runtime as libflightbatch."""
    import os
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import ctypes as C, os, sys, numpy as np
        sys.path.insert(0, os.path.join(os.getcwd(), "flight.jl_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        import flightbatch as fb
        hip = C.CDLL("libamdhip64.so")
        n = 4096
        w = fb.BatchedWorld(n)
        fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 50, n)))
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=10)
        fb.step(sim, 0.1); w.sync()
        print("STEPPED_OK", flush=True)
        uid = C.create_string_buffer(128)
        fb._lib.check(fb.lib.fb_comm_unique_id(uid))
        comm = C.c_void_p()
        fb._lib.check(fb.lib.fb_comm_init(w._h, 1, 0, uid, C.byref(comm)))
        recv = C.c_void_p()
        assert hip.hipMalloc(C.byref(recv), C.c_size_t(27 * n * 8)) == 0
        fb._lib.check(fb.lib.fb_gather_state(w._h, comm, recv))
        w.sync()
        out = np.zeros((27, n))
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), recv, C.c_size_t(27 * n * 8), 2) == 0
        assert np.array_equal(out, w.x), "gathered panel differs from the state"
        fb._lib.check(fb.lib.fb_comm_destroy(comm))
        print("RCCL_GATHER_OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        # seen once on a GPU box: ncclCommInitRank of the system RCCL never returned (nothing of this library was running any
        # more: the stepping before it had completed). That is the box's RCCL, not the gather: skip, but only in exactly that case.
        assert "STEPPED_OK" in out, "the child hung before reaching RCCL: " + out[-2000:]
        pytest.skip("RCCL communicator initialisation did not return on this box within 150 s")
    assert "RCCL_GATHER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rccl_gather_state_two_ranks(fb):
    """fb_comm_init / fb_gather_state with world = 2: two child processes, one GPU each, the unique id handed over through a file.
    Needs two visible devices (the per-round GPU box has one: skipped there; the 8-GPU node runs it). Rank r trims its own shard
    (flightbatch.sharding.shard_range of 8191 aircraft: RAGGED shards, 4096 + 4095) and both must end up with the same gathered
    [2][27][n_max] panel, rank r's rows holding n_of[r] valid entries (fb_comm_shard_sizes)."""
    import os
    import subprocess
    import sys
    import tempfile
    import textwrap
    import torch
    if torch.cuda.device_count() < 2:      # counting devices does not initialise the GPU
        pytest.skip("fb_gather_state with 2 ranks needs 2 GPUs; this box has %d" % torch.cuda.device_count())
    code = textwrap.dedent("""
        import ctypes as C, os, sys, time, numpy as np
        rank, idfile, outfile = int(sys.argv[1]), sys.argv[2], sys.argv[3]
        sys.path.insert(0, os.path.join(os.getcwd(), "flight.jl_amd"))
        import flightbatch as fb
        hip = C.CDLL("libamdhip64.so")
        N = 8191
        lo, hi = fb.sharding.shard_range(N, rank, 2)
        n = hi - lo
        w = fb.BatchedWorld(n, device=rank)
        fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 50, N)[lo:hi]))
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=10)
        fb.step(sim, 0.1); w.sync()
        uid = C.create_string_buffer(128)
        if rank == 0:
            fb._lib.check(fb.lib.fb_comm_unique_id(uid))
            open(idfile + ".tmp", "wb").write(uid.raw); os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                assert time.time() - t0 < 60, "rank 0 never published the unique id"
                time.sleep(0.05)
            uid = C.create_string_buffer(open(idfile, "rb").read(), 128)
        comm = C.c_void_p()
        fb._lib.check(fb.lib.fb_comm_init(w._h, 2, rank, uid, C.byref(comm)))
        n_of = (C.c_int64 * 2)(); n_max = C.c_int64()
        fb._lib.check(fb.lib.fb_comm_shard_sizes(comm, n_of, C.byref(n_max)))
        assert list(n_of) == [4096, 4095] and n_max.value == 4096 and n_of[rank] == n
        recv = C.c_void_p()
        assert hip.hipSetDevice(rank) == 0 and hip.hipMalloc(C.byref(recv), C.c_size_t(2 * 27 * n_max.value * 8)) == 0
        fb._lib.check(fb.lib.fb_gather_state(w._h, comm, recv))
        w.sync()
        out = np.zeros((2, 27, n_max.value))
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), recv, C.c_size_t(out.nbytes), 2) == 0
        assert np.array_equal(out[rank][:, :n], w.x), "own shard differs in the gathered panel"
        out[1][:, n_of[1]:] = 0.0      # (the padding of the shorter shard is unspecified)
        np.save(outfile, out)
        fb._lib.check(fb.lib.fb_comm_destroy(comm))
        print("RANK_OK", rank)
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        idfile = os.path.join(tmp, "uid")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, "-c", code, str(r), idfile, os.path.join(tmp, f"out{r}.npy")], cwd=root, env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = []
        for p_ in procs:
            try:
                outs.append(p_.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        for r, o in enumerate(outs):
            assert f"RANK_OK {r}" in o, o[-3000:]
        g0, g1 = np.load(os.path.join(tmp, "out0.npy")), np.load(os.path.join(tmp, "out1.npy"))
        assert np.array_equal(g0, g1) and not np.array_equal(g0[0], g0[1])


def test_takeoff_ground_to_air_handover(fb, oracle):
    """Take-off roll from rest, rotation, lift-off and climb through the 10 m limit: lanes start in the ground-capable pass
    (non-zero contact regulators, weight on wheels) and must migrate to the airborne pass without a seam. Ground contact is
    ill-conditioned (see test_ground_contact_matches_oracle), so the comparison with the oracle is on the flight path."""
    n = 128
    rng = np.random.default_rng(23)
    env = oracle.default_env()
    r = oracle.trim(lattice_trim_params(fb, 1, seed=1).pack(1), fb.TrimState(1), env)
    x = np.repeat(r["x"], n, axis=1)
    x[21:27] = 0; x[12:16] = np.array([1.0, 0, 0, 0])[:, None]            # level, at rest
    y0 = oracle.f_ode(x[:, :1], r["u"], r["ui"], r["s"], env)[1][:, 0]
    x[20] += 1.85 - y0[fb.K["FB_Y_KIN"] + 21]                              # wheels just compressed on the runway (terrain at 0 m)
    u = np.repeat(r["u"], n, axis=1)
    u[fb.K["FB_U_THROTTLE"]] = 1.0; u[fb.K["FB_U_FLAPS"]] = 0.3
    u[fb.K["FB_U_ELEVATOR"]] = rng.uniform(-0.35, -0.2, n)                 # stick held back: rotates when the tail has authority
    u[fb.K["FB_U_AILERON"]] = 0.0; u[fb.K["FB_U_RUDDER"]] = rng.uniform(0.0, 0.08, n)
    ui = np.repeat(r["ui"], n); s = np.repeat(r["s"], n, axis=1)
    w = fb.BatchedWorld(n)
    w.set_state(x, s); w.u = u; w.ui = ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    xo, so = x.copy(), s.copy()
    lift_g = np.full(n, np.nan); lift_o = np.full(n, np.nan)
    for k in range(35):
        fb.step(sim, 1.0); w.sync()
        xo, so, sto = oracle.step(xo, u, ui, so, env, 0.01, 100)
        fb.f_ode(w); yg = w.y
        yo = oracle.f_ode(xo, u, ui, so, env)[1]
        wg = (yg[79] + yg[90] + yg[101]) > 0; wo = (yo[79] + yo[90] + yo[101]) > 0
        lift_g = np.where(np.isnan(lift_g) & ~wg, k + 1.0, lift_g); lift_o = np.where(np.isnan(lift_o) & ~wo, k + 1.0, lift_o)
    assert np.array_equal(w.status, sto)
    ok = (sto == 0)
    agl = yo[fb.K["FB_Y_KIN"] + 21]
    print("lift-off after %.0f-%.0f s; final height %.1f-%.1f m; still on wheels: %d; terminated %d"
          % (np.nanmin(lift_o), np.nanmax(lift_o), agl[ok].min(), agl[ok].max(), int(np.isnan(lift_o).sum()), int((~ok).sum())))
    assert np.isnan(lift_o).sum() == 0 and (agl[ok] > 10).mean() > 0.8, "the scenario must leave the ground and cross the hand-over"
    assert np.array_equal(lift_g, lift_o)
    err = (np.abs(w.x - xo) / state_scale(xo))[:, ok]
    print("take-off, max scaled error after 35 s: %.2e" % err.max())
    assert err.max() < 1e-3 and np.abs(w.x[20] - xo[20])[ok].max() < 1e-2
    w.close()


@pytest.mark.parametrize("eps, tol", [(0.0, 2e-12), (1e-8, 2e-12), (1e-6, 2e-9)])
def test_gravity_at_com_with_off_norm_states(fb, oracle, eps, tol):
    """The kernels get the gravity vector at the centre of mass from a first-order expansion about the body origin instead of the
    reference's ECEF->geodetic conversion and local-level quaternion (docs/design/k_step_air.md), carrying the state quaternions' norm errors
    (~1e-8 at RK stages) to first order: g_c and the accelerations must match the oracle to rounding at eps = 1e-8, and to
    O(eps^2 g) at a norm error a hundred times larger than any that occurs."""
    n = 16384
    rng = np.random.default_rng(77)
    x = np.zeros((27, n))
    x[0] = rng.uniform(-0.1, 0.3, n); x[1] = rng.uniform(-0.1, 0.1, n)
    x[8] = rng.uniform(0, 1, n)
    x[9] = rng.uniform(150, 280, n)
    q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0); x[12:16] = q * (1 + rng.uniform(-eps, eps, n))
    qe = rng.normal(size=(4, n)); qe /= np.linalg.norm(qe, axis=0); x[16:20] = qe * (1 + rng.uniform(-eps, eps, n))
    x[20] = rng.uniform(100, 10000, n)
    x[21:24] = rng.normal(0, 0.3, (3, n))
    x[24:27] = rng.normal(0, 1, (3, n)); x[24:27] *= rng.uniform(20, 80, n) / np.linalg.norm(x[24:27], axis=0)
    s = np.stack([np.zeros(n), np.full(n, 2)]).astype(np.int32)
    u = np.zeros((16, n)); u[0] = 0.6; u[9] = 0.5; u[11:16] = rng.uniform(0, 100, (5, n))
    ui = np.full(n, fb.K["FB_UI_MIXTURE_AUTO"], dtype=np.int32)
    env = oracle.default_env(h_trn=-600.0)
    w = fb.BatchedWorld(n)
    w.set_params(h_terrain=-600.0)
    w.set_state(x, s); w.u = u; w.ui = ui
    xd = np.zeros((27, n)); fb.f_ode(w, xd)
    xdo, yo, sto = oracle.f_ode(x, u, ui, s, env)
    ok = (sto == 0) & (w.status == 0)
    assert ok.mean() > 0.99
    k = fb.K["FB_Y_DYN"] + 37
    eg = np.abs(w.y[k:k + 3] - yo[k:k + 3])[:, ok].max()
    ev = np.abs(xd[24:27] - xdo[24:27])[:, ok].max()
    print("eps %.0e: |g_c - oracle| <= %.2e m/s^2, |vdot - oracle| <= %.2e" % (eps, eg, ev))
    assert eg < tol and ev < 5 * tol
    w.close()


def test_ground_roll_with_steering_and_brakes_matches_oracle(fb, oracle):
    """One second of ground roll through the STEPPING kernels' contact branch (the FAST forms of gear_ground_kinematics / gear_ground_force,
    c172_kernels.hpp) in every variant it has: nose wheel steered (rudder + offset, a series for cos / sin of half the steering angle) on
    two thirds of the aircraft and castoring (half-angle of the velocity azimuth) on the rest, differential braking, taxi speeds from 1
    to 15 m/s with sideslip, engine at idle and running — against the oracle, which follows the reference operation by operation."""
    n = 1536
    rng = np.random.default_rng(19)
    x = np.zeros((27, n))
    x[8] = 0.5
    th = rng.uniform(-0.01, 0.03, n); ph = rng.uniform(-0.01, 0.01, n); ps = rng.uniform(-np.pi, np.pi, n)

    def qmul(a, b):
        return np.stack([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
    z = np.zeros(n)
    x[12:16] = qmul(qmul(np.stack([np.cos(ps/2), z, z, np.sin(ps/2)]), np.stack([np.cos(th/2), z, np.sin(th/2), z])), np.stack([np.cos(ph/2), np.sin(ph/2), z, z]))
    lat, lon = -0.4, 2.1
    a = -(lat + np.pi / 2)
    x[16:20] = qmul(np.array([np.cos(lon/2), 0, 0, np.sin(lon/2)])[:, None] * np.ones(n), np.array([np.cos(a/2), 0, np.sin(a/2), 0])[:, None] * np.ones(n))
    n_e = np.array([np.cos(lat)*np.cos(lon), np.cos(lat)*np.sin(lon), np.sin(lat)])
    import ctypes
    geoid = oracle.lib.fo_geoid_height(np.ascontiguousarray(n_e).ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    x[20] = geoid + rng.uniform(1.80, 1.88, n)           # on its wheels, struts a little compressed
    x[24] = rng.uniform(1.0, 15, n)      # rolling. (Below ~5 cm/s the friction regulators' stick-slip — k_i = 400 1/s at dt = 0.01 — amplifies
                                         # rounding without bound: the ORACLE against itself with the velocities moved by one ulp is 1e-5 apart
                                         # after 40 steps and 0.2 after 100 on 1 % of such aircraft. Nothing a parity test can hold)
    x[25] = rng.normal(0, 0.2, n)
    x[9] = np.where(np.arange(n) % 2 == 0, 70.0, 230.0)
    s = np.zeros((2, n), np.int32); s[1] = 2
    u = np.zeros((16, n)); u[11:16] = np.array([75, 75, 0, 0, 50.0])[:, None]
    u[0] = np.where(np.arange(n) % 2 == 0, 0.0, 0.4); u[1] = 0.5
    K = fb.K
    u[K["FB_U_RUDDER"]] = rng.uniform(-1, 1, n); u[K["FB_U_RUDDER_OFFSET"]] = rng.uniform(-0.3, 0.3, n)     # the steering input
    u[K["FB_U_BRAKE_LEFT"]] = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 1, n))
    u[K["FB_U_BRAKE_RIGHT"]] = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0, 1, n))
    ui = np.full(n, K["FB_UI_MIXTURE_AUTO"] | K["FB_UI_STEERING_ENGAGED"], np.int32)
    ui[::3] = K["FB_UI_MIXTURE_AUTO"]                     # castoring
    w = fb.BatchedWorld(n)
    w.set_state(x, s); w.u = u; w.ui = ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=25)
    # one step: no time for the contact dynamics to amplify anything — what differs here is the arithmetic of one evaluation chain
    # (strut compression is a difference of ECEF positions: 1e-9 m of rounding in front of 4e4 N/m, 2e-7 m/s^2 on a 1000 kg aircraft)
    fb.step(sim, 0.01); w.sync()
    x1, s1, st1 = oracle.step(x, u, ui, s, oracle.default_env(), 0.01, 1)
    e1 = np.abs(w.x - x1) / np.maximum(np.abs(x1), 1.0)
    print("ground roll, 1 step: max error %.2e (per unit of state, or relative above 1)" % e1.max())
    assert e1.max() < 2e-8 and np.array_equal(w.status, st1) and np.array_equal(w.s, s1)
    fb.step(sim, 0.99); w.sync()
    xo, so, sto = oracle.step(x, u, ui, s, oracle.default_env(), 0.01, 100)
    assert np.array_equal(w.status, sto) and np.array_equal(w.s, so)
    live = sto == 0
    assert live.mean() > 0.95
    _, yo, _ = oracle.f_ode(xo, u, ui, so, oracle.default_env())
    wow = (yo[K["FB_Y_LDG"] + 1] + yo[K["FB_Y_LDG"] + 12] + yo[K["FB_Y_LDG"] + 23])
    assert (wow[live] == 3).mean() > 0.9, "the batch must stay on its wheels"
    err = (np.abs(w.x - xo) / state_scale(xo))[:, live]
    per = err.max(0)
    steered = (ui[live] & K["FB_UI_STEERING_ENGAGED"]) != 0
    q = np.quantile(per, [0.5, 0.99, 1.0])
    print("ground roll, 100 steps: per-aircraft max scaled error, quantiles 50 / 99 / 100 %%: %.2e %.2e %.2e (steered %.2e, castoring %.2e)" % (
        q[0], q[1], q[2], per[steered].max(), per[~steered].max()))
    # 100 steps of braking, skidding and bouncing (the batch is dropped onto struts compressed by up to 8 cm) amplify that: the median stays
    # at the one-step level, a steered and braked aircraft in a skid reaches 1e-4. The stepping kernels' own forms of the contact branch and
    # the reference's operations in their place (-DFB_GROUND_REFERENCE_FORMS) give these same three numbers to three digits.
    assert q[0] < 1e-6 and q[1] < 1e-5 and q[2] < 1e-3
    w.close()


@pytest.mark.parametrize("kin", ["ECEF", "NED"])
def test_f_step_in_the_other_mechanisations(fb, oracle, kin):
    """f_step!(world) for Cessna172Sv0(ECEF()) — q_eb and n_e renormalised beyond 1e-8 (kinematics.jl:317-320) — and (NED()) — nothing to
    renormalise (:409) — with the stall hysteresis, the engine's state machine and the regulator resets, against the oracle."""
    K = fb.K
    n = 512
    nk = {"ECEF": 8, "NED": 6}[kin]
    w = fb.BatchedWorld(n, kinematics=kin)
    fb.f_init(w, lattice_trim_params(fb, n, seed=5))
    rng = np.random.default_rng(4)
    x = w.x; s = w.s; ui = w.ui
    if kin == "ECEF":
        x[12] = np.where(rng.random(n) < 0.5, 3.0, x[12])          # q_eb[1] = 3 forces renormalisation
        x[16:19] *= (1 + 1e-7 * rng.random(n))                      # n_e: norm drift > 1e-8 on some, below it on others
    x[2:8] = rng.normal(0, 1, (6, n))
    v0 = 12 + nk + 3
    x[v0] = np.where(rng.random(n) < 0.3, 12.0, x[v0]); x[v0 + 2] = np.where(rng.random(n) < 0.3, 8.0, x[v0 + 2])   # high alpha -> stall
    s[0] = rng.integers(0, 2, n); s[1] = rng.integers(0, 3, n)
    x[9] = rng.uniform(10, 300, n)
    x[8] = np.where(rng.random(n) < 0.2, -0.01, x[8])
    ui = ui | np.where(rng.random(n) < 0.5, 1, 0).astype(np.int32) | np.where(rng.random(n) < 0.2, 2, 0).astype(np.int32)
    w.set_state(x, s); w.ui = ui
    fb.f_step(w); w.sync()
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        x27 = np.zeros((27, n)); x27[:12 + nk] = x[:12 + nk]; x27[21:] = x[12 + nk:]
        xo27, so, sto = oracle.f_step(x27, w.u, ui, s, oracle.default_env())
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    xo = np.vstack([xo27[:12 + nk], xo27[21:]])
    assert (w.s == so).all()
    assert np.max(np.abs(w.x - xo) / np.maximum(np.abs(xo), 1e-3)) < 1e-14
    if kin == "ECEF":
        assert (np.abs(np.sqrt((w.x[12:16] ** 2).sum(0)) - 1) < 1e-8 * (1 + 1e-6)).all() and (np.abs(np.sqrt((w.x[16:19] ** 2).sum(0)) - 1) <= 1e-7).all()
    w.close()


@pytest.mark.parametrize("kin", ["WA", "ECEF", "NED"])
def test_ground_contact_with_off_norm_attitude_states(fb, oracle, kin):
    """f_ode! in ground contact with the attitude quaternions off unit norm by up to 1e-6 — what an RK stage hands the landing gear while an
    aircraft is being thrown about (the states are renormalised only in f_step!, and only beyond 1e-8) — in each mechanisation against the
    oracle. kinematics.y.q_en, which rotates the terrain normal (landinggear.jl:267), is q_eb ∘ q_nb' in WA (off norm by |q_wb|²) and
    ltf(n_e) in ECEF / NED (kinematics.jl:195, 293, 377): a contact branch that forms it the WA way everywhere is off by 2e-6 x the strut
    force for ECEF here."""
    K = fb.K
    n = 768
    nk = {"WA": 9, "ECEF": 8, "NED": 6}[kin]
    rng = np.random.default_rng(29)

    def qmul(a, b):
        return np.stack([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                         a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])
    z = np.zeros(n); one = np.ones(n)
    th = rng.uniform(-0.05, 0.10, n); ph = rng.uniform(-0.12, 0.12, n); ps = rng.uniform(-np.pi, np.pi, n)
    q_nb = qmul(qmul(np.stack([np.cos(ps/2), z, z, np.sin(ps/2)]), np.stack([np.cos(th/2), z, np.sin(th/2), z])), np.stack([np.cos(ph/2), np.sin(ph/2), z, z]))
    lat, lon = 0.35, 2.4
    a = -(lat + np.pi / 2)
    q_en = qmul(np.stack([np.cos(lon/2) * one, z, z, np.sin(lon/2) * one]), np.stack([np.cos(a/2) * one, z, np.sin(a/2) * one, z]))   # ltf: Rz(lon) ∘ Ry(-(lat + π/2))
    n_e = np.array([np.cos(lat)*np.cos(lon), np.cos(lat)*np.sin(lon), np.sin(lat)])
    import ctypes
    geoid = oracle.lib.fo_geoid_height(np.ascontiguousarray(n_e).ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    h_e = geoid + rng.uniform(1.50, 1.80, n)                       # struts compressed by up to 35 cm (banked: one side more)
    off = lambda k: 1 + rng.uniform(-1e-6, 1e-6, (1, n)) * np.ones((k, 1))
    if kin == "WA":   # ψ_nw = 0: q_wb = q_nb, q_ew = q_en (the position quaternion turns at 1e-5 rad/s: it is never further off norm than f_step!'s 1e-8)
        kinrows = np.vstack([q_nb * off(4), q_en * (1 + rng.uniform(-4e-9, 4e-9, (1, n))), h_e[None]])
    elif kin == "ECEF":
        kinrows = np.vstack([qmul(q_en, q_nb) * off(4), n_e[:, None] * (1 + rng.uniform(-8e-9, 8e-9, (1, n))), h_e[None]])
    else:
        kinrows = np.vstack([ps[None], th[None], ph[None], lat * one[None], lon * one[None], h_e[None]])
    x = np.zeros((18 + nk, n))
    x[8] = 0.5; x[9] = 100.0
    x[2:8] = rng.normal(0, 0.2, (6, n))
    x[12:12 + nk] = kinrows
    x[12 + nk:12 + nk + 3] = rng.normal(0, 0.05, (3, n))
    x[12 + nk + 3] = rng.uniform(0, 20, n); x[12 + nk + 4] = rng.normal(0, 0.5, n); x[12 + nk + 5] = rng.uniform(-0.5, 3.0, n)
    s = np.zeros((2, n), np.int32); s[1] = 2
    u = np.zeros((16, n)); u[11:16] = np.array([75, 75, 0, 0, 50.0])[:, None]; u[0] = 0.2; u[1] = 0.5
    u[K["FB_U_BRAKE_LEFT"]] = rng.uniform(0, 1, n); u[K["FB_U_BRAKE_RIGHT"]] = rng.uniform(0, 1, n); u[K["FB_U_RUDDER"]] = rng.uniform(-1, 1, n)
    ui = np.full(n, K["FB_UI_MIXTURE_AUTO"] | K["FB_UI_STEERING_ENGAGED"], np.int32); ui[::3] = K["FB_UI_MIXTURE_AUTO"]
    w = fb.BatchedWorld(n, kinematics=kin)
    w.set_state(x, s); w.u = u; w.ui = ui
    xd = np.zeros((18 + nk, n)); fb.f_ode(w, xd)
    y = w.y
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        x27 = np.zeros((27, n)); x27[:12 + nk] = x[:12 + nk]; x27[21:] = x[12 + nk:]
        xdo27, yo, sto = oracle.f_ode(x27, u, ui, s, oracle.default_env())
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    xdo = np.vstack([xdo27[:12 + nk], xdo27[21:]])
    wow = yo[K["FB_Y_LDG"] + 1] + yo[K["FB_Y_LDG"] + 12] + yo[K["FB_Y_LDG"] + 23]
    assert (wow > 0).mean() > 0.8 and (w.status == sto).all()
    err = np.abs(xd - xdo) / np.maximum(np.abs(xdo), 1.0)
    Fz = np.abs(yo[K["FB_Y_LDG"] + 7]).max()
    print(f"{kin}: ground contact with off-norm attitude states: max xdot error {err.max():.2e} (largest strut force {Fz:.0f} N)")
    assert err.max() < 1e-6, (kin, err.max(), np.unravel_index(err.argmax(), err.shape))
    scy = np.maximum(np.abs(yo), 1.0); scy[22:25] = 6.4e6
    for g in range(3):
        k0 = K["FB_Y_LDG"] + 11 * g
        scy[k0 + 4: k0 + 11] = np.maximum(scy[k0 + 4: k0 + 11], 100.0)
        scy[k0 + 1: k0 + 11, np.abs(yo[k0]) < 1e-6] = np.inf
    scy[134 + 13: 134 + 19] = np.maximum(scy[134 + 13: 134 + 19], 100.0)
    erry = np.abs(y - yo) / scy
    r_, l_ = np.unravel_index(erry.argmax(), erry.shape)
    print(f"{kin}: max y error {erry.max():.2e} at row {r_} (gpu {y[r_, l_]:.9e}, oracle {yo[r_, l_]:.9e}); rows above 3e-7: {np.nonzero(erry.max(1) > 3e-7)[0]}")
    # (strut forces reach 46 kN here, three times those of test_ground_contact_matches_oracle: the 1e-9 m of rounding in a compression is
    # 1e-4 N, 1.3e-6 of the 100 N floor. The contact branch with the WA form of q_en in every mechanisation is off by 2e-4 in xdot for ECEF.)
    assert erry.max() < 3e-6, (kin, erry.max(), np.unravel_index(erry.argmax(), erry.shape))
    w.close()
