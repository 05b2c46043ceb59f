"""The two airborne fp64 steppers of Cessna172Sv0 against each other, in each kinematic mechanisation: the wave-specialised k_step_duo (two
waves per SIMD, the default) and the one-wave-per-SIMD k_step_air (FLIGHTBATCH_DUO=0). Same physics, different evaluation order and fma
contraction: they agree to rounding, lane by lane, including on the lanes that are not ordinary — beyond the end of a ragged batch,
terminated before the launch, sitting on the ground, sinking through the hand-over clearance in the middle of a launch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


H_E_ROW = {"WA": 20, "ECEF": 19, "NED": 17}     # the ellipsoidal altitude in the C ABI's state of each mechanisation


def _world(fb, n, duo, kin="WA"):
    old = os.environ.get("FLIGHTBATCH_DUO")
    os.environ["FLIGHTBATCH_DUO"] = "1" if duo else "0"
    try:
        return fb.BatchedWorld(n, kinematics=kin)
    finally:
        if old is None:
            del os.environ["FLIGHTBATCH_DUO"]
        else:
            os.environ["FLIGHTBATCH_DUO"] = old


def _scale(x):
    return np.maximum(np.abs(x), 1e-3)


@pytest.mark.parametrize("n,spl,kin", [(1000, 50, "WA"), (333, 7, "WA"), (64, 1, "WA"), (1000, 50, "ECEF"), (333, 7, "NED"), (1000, 50, "NED"), (64, 1, "ECEF")])
def test_duo_and_air_steppers_agree(fb, n, spl, kin):
    rng = np.random.default_rng(23 + n)
    he = H_E_ROW[kin]
    h_trn = 250.0
    cruise = rng.random(n) < 0.6
    h = np.where(cruise, h_trn + rng.uniform(300, 4000, n), h_trn + rng.uniform(11, 30, n))       # the rest: short final, will cross 10 m
    gam = np.where(cruise, np.deg2rad(rng.uniform(-2, 4, n)), -np.deg2rad(rng.uniform(2, 5, n)))
    tp = fb.TrimParameters(EAS=np.where(cruise, rng.uniform(38, 58, n), rng.uniform(33, 40, n)), h_e=h, γ_wb_n=gam,
                           flaps=np.where(cruise, 0.0, 1.0), ψ_nb=rng.uniform(-3, 3, n))
    ref = _world(fb, n, False, kin)
    ref.set_params(h_terrain=h_trn)
    fb.f_init(ref, tp)
    x0, s0, u0, ui0 = ref.x, ref.s, ref.u.copy(), ref.ui
    ok = ref.trim_success
    u0[fb.K["FB_U_ELEVATOR"]] += rng.uniform(-0.03, 0.03, n)      # not a steady state
    u0[fb.K["FB_U_AILERON"]] += rng.uniform(-0.03, 0.03, n)
    u0[fb.K["FB_U_M_PILOT"]] = rng.uniform(50, 100, n)             # per-aircraft payload: the mass-property sums differ lane by lane
    x0 = x0.copy()
    far = np.nonzero(cruise & ok)[0][:8]
    x0[he, far[:4]] = 90e3                                          # above the ISA model (FB_ST_ISA_RANGE)
    x0[he, far[4:]] = -1500.0                                       # below the altitude range (FB_ST_ALT_RANGE)
    st0 = np.zeros(n, np.int32)
    st0[rng.random(n) < 0.05] = fb.K["FB_ST_NAN"]                  # terminated before the launch: must be left alone
    if n >= 512:
        st0[128:256] = fb.K["FB_ST_NAN"]                           # two whole wave pairs of workgroup 0 have nothing to do: they leave at
        cruise[320:384] = True                                     # once, and the workgroup's barriers go on without them
    out = {}
    for duo in (False, True):
        w = _world(fb, n, duo, kin)
        w.set_params(h_terrain=h_trn)
        w.set_state(x0, s0); w.u = u0; w.ui = ui0
        fb._lib.check(fb.lib.fb_set_status(w._h, st0.ctypes.data_as(fb._lib.C.POINTER(fb._lib.C.c_int32))))
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=spl)
        fb.step(sim, 3.0); w.sync()
        out[duo] = (w.x, w.s, w.status)
        w.close()
    (xa, sa, sta), (xd, sd, std) = out[False], out[True]
    assert np.array_equal(sta, std) and np.array_equal(sa, sd)
    hi, lo = far[:4][st0[far[:4]] == 0], far[4:][st0[far[4:]] == 0]   # (those not already terminated before the launch)
    assert len(hi) + len(lo) > 0 and (std[hi] & fb.K["FB_ST_ISA_RANGE"]).all() and (std[lo] & fb.K["FB_ST_ALT_RANGE"]).all()
    dead0 = st0 != 0
    assert np.array_equal(xd[:, dead0], x0[:, dead0])               # untouched
    live = ok & (sta == 0)
    err = np.abs(xd - xa) / _scale(xa)
    agl = xa[he] - h_trn                                            # (ellipsoidal, good enough to tell who came near the ground)
    low = agl < 12 + 20                                             # these spent launches in the ground-capable pass, which both share
    print(kin, "duo vs air after 300 steps: max scaled difference %.2e (aircraft that stayed high), %.2e (went low); handed over: %d" % (
        err[:, live & ~low].max(), err[:, live & low].max() if (live & low).any() else 0.0, int((live & low).sum())))
    assert live.sum() > n // 3
    assert err[:, live & ~low].max() < 1e-10
    if (live & low).any():
        assert err[:, live & low].max() < 1e-6                      # (contact amplifies rounding, see test_approach_crosses_the_air_ground_handover)


def test_wave_pairs_leaving_at_different_steps(fb, oracle):
    """One workgroup of k_step_duo serves 256 aircraft with four wave pairs, and a pair whose lanes have all left (handed over to the
    ground-capable pass) leaves the kernel on its own while the others go on through the workgroup's barriers — which relies on gfx950's
    s_barrier counting only the waves that have not ended. Here every pair of two workgroups runs dry at a DIFFERENT step of one launch
    (its 64 aircraft zoom through the ISA ceiling, pair by pair; the last pair of each workgroup flies on), with a few lanes terminated
    before the launch mixed in: the survivors must agree with the one-wave stepper, the terminated ones with the oracle."""
    from test_gpu_termination import flying_batch, geoid
    n = 512
    rng = np.random.default_rng(77)
    lat = rng.uniform(-1.0, 1.0, n); lon = rng.uniform(-3.0, 3.0, n)
    a = 6378137.0
    ceiling = 84852.0 * a / (a - 84852.0) + geoid(oracle, lat, lon)
    pair = (np.arange(n) // 64) % 4
    below = np.where(pair < 3, 1.0 + 1.7 * pair + rng.uniform(0.0, 0.4, n), 500.0)      # pair p crosses around step 5 + 8 p; pair 3 never
    x, s, u, ui = flying_batch(fb, oracle, n, 77, lat, lon, ceiling - below, np.full(n, 20.0), {})
    st0 = np.zeros(n, np.int32); st0[rng.random(n) < 0.04] = fb.K["FB_ST_NAN"]
    out = {}
    for duo in (False, True):
        w = _world(fb, n, duo)
        w.set_state(x, s); w.u = u; w.ui = ui
        fb._lib.check(fb.lib.fb_set_status(w._h, st0.ctypes.data_as(fb._lib.C.POINTER(fb._lib.C.c_int32))))
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
        fb.step(sim, 1.0); w.sync()
        out[duo] = (w.x, w.s, w.status, w.termination)
        w.close()
    (xa, sa, sta, (tsa, twa)), (xd, sd, std, (tsd, twd)) = out[False], out[True]
    assert np.array_equal(sta, std) and np.array_equal(sa, sd) and np.array_equal(tsa, tsd) and np.array_equal(twa, twd)
    fresh = st0 == 0
    gone = fresh & (pair < 3)
    assert (std[gone] == fb.K["FB_ST_ISA_RANGE"]).all() and (std[fresh & (pair == 3)] == 0).all()
    steps = [np.median(tsd[gone & (pair == p)]) for p in range(3)]
    assert steps[0] < steps[1] < steps[2] < 40, steps                      # the pairs ran dry one after the other, inside the first launch
    assert np.array_equal(xd[:, ~fresh], x[:, ~fresh])                     # terminated before the launch: untouched
    assert np.array_equal(xd[:, gone], xa[:, gone])                        # ended by the same ground-capable pass in both worlds
    err = np.abs(xd - xa)[:, fresh & (pair == 3)] / _scale(xa[:, fresh & (pair == 3)])
    assert err.max() < 1e-10, err.max()
    xo, so, sto, tso, two = oracle.step_term(x, u, ui, s, oracle.default_env(), 0.01, 100, status=st0)
    assert np.array_equal(sto, std) and np.array_equal(tso[gone], tsd[gone]) and np.array_equal(two[gone], twd[gone])
    assert (np.abs(xd - xo) / np.maximum(np.abs(xo), 1e-3)).max() < 1e-9


@pytest.mark.parametrize("n,spl,kin", [(1000, 50, "WA"), (333, 7, "ECEF"), (577, 25, "NED")])
def test_x2_duo_and_air_steppers_agree(fb, n, spl, kin):
    """Cessna172Xv2 under its autopilot on the two airborne steppers: the wave pair (k_step_duo<KIN, true>: actuators and aerodynamic sums on
    role P, the control update in two halves side by side with streamed LQR gains) and the one-wave stepper (k_step_air<KIN, true>,
    FLIGHTBATCH_DUO=0: the update as one call with the record cached). Same physics and control laws, another order of evaluation: state,
    control-law record and discrete states agree to rounding on every aircraft — ragged batch, every pair of control modes, aircraft
    terminated before the launch, aircraft on short final that cross into the ground-capable pass in the middle of a launch (their
    control-law record is put back from ctl_bak and the lane stepped again from the launch-start state)."""
    K = fb.K
    gains = fb.ctl_gains.ctl_gains_blob()
    rng = np.random.default_rng(77 + n)
    h_trn = 250.0
    cruise = rng.random(n) < 0.7
    h = np.where(cruise, h_trn + rng.uniform(300, 3000, n), h_trn + rng.uniform(12, 30, n))
    tp = fb.TrimParameters(EAS=np.where(cruise, rng.uniform(40, 55, n), rng.uniform(35, 42, n)), h_e=h,
                           γ_wb_n=np.where(cruise, 0.0, -np.deg2rad(rng.uniform(2, 4, n))), flaps=np.where(cruise, 0.0, 1.0), ψ_nb=rng.uniform(-3, 3, n))
    lon = rng.integers(0, 9, n); lat = rng.integers(0, 5, n)
    st0 = np.zeros(n, np.int32); st0[rng.random(n) < 0.05] = K["FB_ST_NAN"]
    out = {}
    for duo in (False, True):
        old = os.environ.get("FLIGHTBATCH_DUO")
        os.environ["FLIGHTBATCH_DUO"] = "1" if duo else "0"
        try:
            w = fb.Cessna172Xv2World(n, gains=gains, kinematics=kin)
        finally:
            if old is None: del os.environ["FLIGHTBATCH_DUO"]
            else: os.environ["FLIGHTBATCH_DUO"] = old
        w.set_params(h_terrain=h_trn, wind_ned=(2.0, -1.0, 0.0))
        sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=spl)
        fb.init(sim, tp)
        ok = w.trim_success
        cu = w.cu
        cu[K["FB_CU_LON_MODE_REQ"]] = np.where(cruise, lon, float(fb.ModeControlLon.EAS_clm))
        cu[K["FB_CU_LAT_MODE_REQ"]] = np.where(cruise, lat, float(fb.ModeControlLat.φ_β))
        cu[K["FB_CU_CLM_REF"]] = np.where(cruise, cu[K["FB_CU_CLM_REF"]] + 1.0, -3.0)
        cu[K["FB_CU_EAS_REF"]] += 2.0; cu[K["FB_CU_PHI_REF"]] += 0.2; cu[K["FB_CU_H_REF"]] += 30.0
        w.cu = cu
        fb._lib.check(fb.lib.fb_set_status(w._h, st0.ctypes.data_as(fb._lib.C.POINTER(fb._lib.C.c_int32))))
        fb.step(sim, 2.0)
        uu = w.u                                            # flaps on the move for half of the cruising aircraft (the flap actuator follows its command
        uu[K["FB_U_FLAPS"]] = np.where(cruise & (np.arange(n) % 2 == 0), 0.5, uu[K["FB_U_FLAPS"]])   # with a 50 ms lag: role P re-forms the flap-dependent sums every stage)
        w.u = uu
        fb.step(sim, 2.0); w.sync()
        out[duo] = dict(x=w.x, cs=w.cs, s=w.s, status=w.status, ok=ok)
        w.close()
    a, b = out[False], out[True]
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["s"], b["s"])
    live = a["ok"] & (a["status"] == 0)
    # C ABI row (reference order: the actuators behind the power plant, the mechanisation's unused rows dropped) of the device's altitude row
    perm = [k if k < K["FB_X2_ACT"] else (27 + k - K["FB_X2_ACT"] if k < K["FB_X2_KIN"] else k - K["FB_NACT"]) for k in range(34)]
    perm = [r for r in perm if r not in {"WA": (), "ECEF": (20,), "NED": (18, 19, 20)}[kin]]
    perm_he = perm.index({"WA": 20, "ECEF": 19, "NED": 17}[kin])
    assert a["x"].shape[0] == len(perm)
    landed = live & (a["x"][perm_he] - h_trn < 8.0)
    flying = live & ~landed
    assert flying.sum() > 0.4 * n and (~cruise & live).sum() > 0.1 * n
    ex = np.abs(a["x"] - b["x"]) / np.maximum(np.abs(a["x"]), 1.0)
    ec = np.abs(a["cs"] - b["cs"]) / np.maximum(np.abs(a["cs"]), 1.0)
    print(f"Xv2({kin}) duo vs air after 400 steps ({spl} per launch): flying {int(flying.sum())}: state {ex[:, flying].max():.2e}, record {ec[:, flying].max():.2e}; "
          f"handed to the ground-capable pass / rolling {int(landed.sum())}: state {ex[:, landed].max() if landed.any() else 0.0:.2e}")
    assert ex[:, flying].max() < 1e-9 and ec[:, flying].max() < 1e-9
    assert not landed.any() or np.quantile(ex[:, landed].max(0), 0.9) < 1e-6   # (on their wheels: ill-conditioned, see tests/conditioning.py; the two differ by rounding only)
    dead = a["status"] != 0
    assert np.array_equal(a["x"][:, dead], b["x"][:, dead])   # terminated before the launch: left alone by both
