"""The oracle's trim solver (oracle/fo_trim.hpp, fo_c172.hpp: trim_solve) against the reference's contract for
f_init!(vehicle, TrimParameters) (FlightApps/src/c172/c172.jl:883-942): from TrimState(), inside the bounds of
:901-917, success <=> cost <= stopval = 1e-16. NLopt's BOBYQA iterates cannot be reproduced (third-party, absent),
so the arbiter of "a trim exists" is continuation: a point whose trim can be reached from a neighbouring solution
must also trim from the default guess."""
import ctypes as C
import numpy as np

TS0 = np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])   # TrimState() defaults, c172.jl:796-804


def default_tp(n):
    tp = np.zeros((18, n)); tp[0] = 1; tp[3] = 1050; tp[5] = 50; tp[10] = 0.5; tp[11] = 0.5
    tp[13:18] = np.array([75, 75, 0, 0, 50.0])[:, None]      # PayloadY() defaults
    return tp


def bench_lattice_tp():
    """the 32 x 32 (EAS, h) cells of bench.py's config-3 lattice (heading 0)"""
    i, j = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    tp = default_tp(1024)
    tp[5] = (35.0 + 20.0 * i / 31.0).ravel(); tp[3] = (200.0 + 2800.0 * j / 31.0).ravel()
    return tp


def test_bench_lattice_trims_everywhere(oracle):
    tp = bench_lattice_tp()
    r = oracle.trim(tp, np.tile(TS0[:, None], (1, 1024)), oracle.default_env())
    assert r["ok"].all(), f"{(~r['ok']).sum()} lattice cells failed to trim"
    assert r["cost"].max() < 1e-24          # iterated to the rounding floor, far below stopval
    ts = r["ts"].reshape(7, 32, 32)
    # the band the round-1 clamped Newton lost (EAS 50.5 m/s, h = 1735 ... 2187 m): throttle ~0.70, n_eng ~0.92
    band = ts[:, 24, 17:23]
    assert abs(35.0 + 20.0 * 24 / 31.0 - 50.48) < 0.01 and abs(200.0 + 2800.0 * 17 / 31.0 - 1735.5) < 0.1
    assert (band[3] > 0.66).all() and (band[3] < 0.75).all(), band[3]
    assert (band[2] > 0.89).all() and (band[2] < 0.95).all(), band[2]
    # nothing sits on a bound it does not need: full throttle is never required inside this lattice
    assert ts[3].max() < 0.9 and ts[2].max() < 1.05 and ts[0].min() > 0.0 and ts[0].max() < 0.2


def test_trim_equals_continuation_from_a_neighbour(oracle):
    """continuation is the arbiter: start every cell from the solution of the previous cell along h and along EAS"""
    tp = bench_lattice_tp()
    env = oracle.default_env()
    r = oracle.trim(tp, np.tile(TS0[:, None], (1, 1024)), env)
    ts = r["ts"].reshape(7, 32, 32)
    for shift_axis in (1, 2):
        guess = np.roll(ts, 1, axis=shift_axis).reshape(7, 1024)
        rn = oracle.trim(tp, guess, env)
        assert rn["ok"].all()
        assert np.abs(rn["ts"] - r["ts"]).max() < 1e-9


def test_success_set_is_the_flight_envelope(oracle):
    """wider than the aircraft can fly: below the stall speed, and where the engine's power runs out with altitude, there is
    no trim and the solver must say so (success = false, like the reference's warning path c172.jl:935-937); in between
    the success set is a solid region: the descent from TrimState() alone (no continuation fallback) already finds every
    trim that parameter continuation from TrimParameters() reaches."""
    i, j = np.meshgrid(np.arange(40), np.arange(24), indexing="ij")
    n = i.size
    tp = default_tp(n)
    EAS = (25.0 + 45.0 * i / 39.0); h = (150.0 + 5850.0 * j / 23.0)
    tp[5] = EAS.ravel(); tp[3] = h.ravel()
    env = oracle.default_env()
    guess = np.tile(TS0[:, None], (1, n))
    oracle.lib.fo_trim_continued.restype = C.c_int64
    full = oracle.trim(tp, guess, env)
    assert oracle.lib.fo_trim_continued() == 0          # the fallback rescued nobody here ...
    oracle.lib.fo_set_trim_continuation(0)
    try:
        primary = oracle.trim(tp, guess, env)
    finally:
        oracle.lib.fo_set_trim_continuation(1)
    assert (primary["ok"] == full["ok"]).all()           # ... and changes nothing
    ok = full["ok"].reshape(40, 24)
    assert not ok[EAS[:, 0] < 27.0].any()                # below the clean stall speed
    assert not ok[EAS[:, 0] > 67.5].any()                # beyond the power available at any altitude
    assert ok[(EAS[:, 0] > 35) & (EAS[:, 0] < 45)].all()
    for a in range(40):                                   # along altitude the feasible set is an interval starting at the bottom
        row = ok[a]
        if row.any():
            assert row[: np.nonzero(row)[0].max() + 1].all()
    # failed lanes report a cost above stopval and stay inside the bounds of c172.jl:901-917
    lo = np.array([-np.pi / 12, -np.pi / 3, 0.4, 0, -1, -1, -1])[:, None]; hi = np.array([0.36, np.pi / 3, 1.1, 1, 1, 1, 1])[:, None]
    assert (full["cost"][~full["ok"]] > 1e-16).all()
    assert (full["ts"] >= lo - 1e-15).all() and (full["ts"] <= hi + 1e-15).all()


def test_kinked_maps_do_not_trap_the_descent(oracle):
    """tests/golden/trim_hard_cases.npz: five fuzzed TrimParameters whose trim sits within 1e-3 of the n = 1.074 knot of the
    engine maps (piston.jl:110,140), or whose descent path runs along the alpha = 0.09 / 0.10 / 0.26 knots of C_L
    (c172.jl:121) — a smooth-model trust region creeps or stops there. All five have a trim."""
    import os
    tp = np.load(os.path.join(os.path.dirname(__file__), "golden", "trim_hard_cases.npz"))["trim_params"]
    n = tp.shape[1]
    env = oracle.default_env()
    r = oracle.trim(tp, np.tile(TS0[:, None], (1, n)), env)
    assert r["ok"].all(), r["cost"]
    assert r["cost"].max() < 1e-24
    D = C.POINTER(C.c_double)
    for k in range(n):   # the reported cost is the cost of the reported state
        cost = oracle.lib.fo_c172_trim_cost(np.ascontiguousarray(tp[:, k]).ctypes.data_as(D), np.ascontiguousarray(r["ts"][:, k]).ctypes.data_as(D), env.ctypes.data_as(D))
        assert cost <= 1e-16
    assert np.abs(r["ts"][2, 1:3] - 1.074).max() < 1e-3      # the two engine-knot cases


def test_per_aircraft_environment_equals_one_call_per_aircraft(oracle):
    """the oracle's per-aircraft environment mode (fo_set_env_per_aircraft: env [7, n], one Env per aircraft — every simulation of the
    reference owns its world, FP/atmosphere.jl:75-84,156-165) against n single-aircraft calls with that aircraft's block: trim, f_ode!
    and 200 steps, bit for bit."""
    n = 16
    rng = np.random.default_rng(4)
    tp = default_tp(n)
    tp[5] = rng.uniform(38, 52, n); tp[3] = rng.uniform(300, 2500, n); tp[4] = rng.uniform(-3, 3, n)
    env6 = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-1, 1, n), rng.uniform(260, 310, n), rng.uniform(97e3, 104e3, n),
                     rng.uniform(0, 100, n)])
    envs = oracle.env_rows(env6)
    ts0 = np.tile(TS0[:, None], (1, n))
    with oracle.per_aircraft_env():
        r = oracle.trim(tp, ts0, envs)
        xd, y, st = oracle.f_ode(r["x"], r["u"], r["ui"], r["s"], envs)
        xf, sf, stf = oracle.step(r["x"], r["u"], r["ui"], r["s"], envs, 0.01, 200)
    assert r["ok"].all() and (stf == 0).all()
    for k in range(n):
        e1 = np.ascontiguousarray(envs[:, k])
        r1 = oracle.trim(np.ascontiguousarray(tp[:, k:k + 1]), ts0[:, k:k + 1], e1)
        assert np.array_equal(r1["x"][:, 0], r["x"][:, k]) and np.array_equal(r1["ts"][:, 0], r["ts"][:, k])
        xd1, y1, _ = oracle.f_ode(r1["x"], r1["u"], r1["ui"], r1["s"], e1)
        assert np.array_equal(xd1[:, 0], xd[:, k]) and np.array_equal(y1[:, 0], y[:, k])
        xf1, _, _ = oracle.step(r1["x"], r1["u"], r1["ui"], r1["s"], e1, 0.01, 200)
        assert np.array_equal(xf1[:, 0], xf[:, k])
    # the environments differ, and so do the trims (density and wind enter the trim)
    assert np.ptp(r["ts"][3]) > 0.02
