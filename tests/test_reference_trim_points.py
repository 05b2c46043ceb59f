"""The oracle AND the HIP path against trim solutions the REFERENCE computed and stored.

The reference's autopilot design sweep trims Cessna172Xv0(NED) with NLopt/BOBYQA at 28 flight conditions — EAS 25...55 m/s,
h 50...3050 m, flaps 1...0 by `flaps_schedule` — and saves the trimmed (q, θ, EAS, α, α_filt, n_eng, throttle, elevator) and
(p, r, φ, EAS, β, β_filt, aileron, rudder) beside the LQR gains (lib/FlightApps/design/c172/c172x_design.jl:87-130,151-216,
549-671; tests/reference_fixtures.py has the row maps). Those are OUTPUTS OF THE REFERENCE'S OWN force / moment / engine /
atmosphere / mass model and trim cost (FA/c172/c172.jl:857-942) at three ISA densities and the whole flap range; the files are
shipped byte-identical (hash-checked below). A trim point is a zero of seven accelerations, so matching the seven unknowns to
1e-6 pins the aerodynamic coefficients, the propeller map, the engine map, the ISA model and the mass properties behind them
two orders tighter than any of the reference's tolerance tests does. Measured (round 5, in the log of every run):
oracle vs reference <= 4.5e-7 (throttle), the size of BOBYQA's stopval = 1e-16 on the squared accelerations."""
import numpy as np
import pytest

import reference_fixtures as rf

TOL = 1e-6   # absolute, every row (radians, normalised engine speed, normalised actuator positions; EAS in m/s)


def _trim_parameters_packed(n_copies=1):
    """C172.TrimParameters(; Ob = Geographic(LatLon(), HEllip(h)), EAS, flaps) — c172x_design.jl:107-112; LatLon() is ϕ = λ = 0,
    i.e. n_e = (1, 0, 0); everything else at its default (c172.jl:806-818)."""
    EAS, h, flaps = (np.tile(a, n_copies) for a in rf.design_nodes())
    n = EAS.size
    tp = np.zeros((18, n)); tp[0] = 1.0; tp[3] = h; tp[5] = EAS; tp[10] = 0.5; tp[11] = 0.5; tp[12] = flaps
    tp[13:18] = np.array([75.0, 75.0, 0.0, 0.0, 50.0])[:, None]
    return tp


def test_shipped_data_files_are_the_references():
    import json
    for rel in json.load(open(rf.HASHES)):
        rf.assert_shipped_copy_is_the_references(rel)


def test_flaps_schedule_and_nodes():
    EAS, h, flaps = rf.design_nodes()
    assert EAS.size == 28 and EAS[0] == 25 and EAS[6] == 55 and h[0] == 50 and h[7] == 1050 and h[27] == 3050
    assert np.array_equal(flaps[:7], [1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0])      # 25, 30 -> 1;  35 and above -> 0
    assert rf.flaps_schedule(32.5) == 0.5
    # the stored EAS / h rows are the node values themselves
    vh = rf.stored("vh2te")
    assert np.allclose(vh["x_trim"][2], EAS, atol=1e-9) and np.array_equal(vh["x_trim"][4], h)


def test_oracle_trim_reproduces_the_references_stored_trim_points(oracle, capsys):
    tp = _trim_parameters_packed()
    env = oracle.default_env()
    ts0 = np.tile(np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])[:, None], (1, 28))   # TrimState(), c172.jl:796-804
    r = oracle.trim(tp, ts0, env)
    assert r["ok"].all(), "the reference trims all 28 design points (the gains exist); so must the oracle"
    xd, y, st = oracle.f_ode(r["x"], r["u"], r["ui"], r["s"], env)
    assert (st == 0).all()
    lon, lat = rf.trim_point_rows(r["ts"], r["x"], y)
    with capsys.disabled():
        worst = rf.compare_with_stored(lon, lat, y, TOL, log=lambda s: print("\n[oracle] " + s))
    assert max(worst.values()) <= TOL
    # the residual of the reference's own solution in the oracle's model: put the stored TrimState in and evaluate the cost
    te, ar = rf.stored("te2te"), rf.stored("ar2ar")
    ts_ref = np.stack([te["x_trim"][3], ar["x_trim"][2], te["x_trim"][5], te["x_trim"][6], ar["x_trim"][6], te["x_trim"][7], ar["x_trim"][7]])
    import ctypes as C
    from oracle_binding import _p
    cost = np.array([oracle.lib.fo_c172_trim_cost(_p(np.ascontiguousarray(tp[:, k])), _p(np.ascontiguousarray(ts_ref[:, k])), _p(env))
                     for k in range(28)])
    with capsys.disabled():
        print(f"[oracle] trim cost of the REFERENCE's stored solutions in the oracle's model: max {cost.max():.2e} (NLopt's stopval 1e-16)")
    assert cost.max() < 3e-15     # BOBYQA stopped at <= 1e-16 in its own arithmetic; the 8-digit agreement of the model is what this bounds


@pytest.mark.gpu
@pytest.mark.parametrize("model, kin", [("Sv0", "WA"), ("Sv0", "NED"), ("Xv2", "NED"), ("Xv2", "WA")])
def test_device_trim_reproduces_the_references_stored_trim_points(fb, model, kin, capsys):
    """fb_trim + fb_f_ode through the C ABI — Cessna172Sv0 and Cessna172Xv2 (the reference's Xv0 airframe + actuators), in the
    reference's design mechanisation (NED) and the default one (WA). 64 copies of the 28 points: every lane of a wave, and the
    persistent trim kernel's queue, see them."""
    copies = 64
    EAS, h, flaps = (np.tile(a, copies) for a in rf.design_nodes())
    n = EAS.size
    K = fb.K
    if model == "Sv0":
        w = fb.BatchedWorld(n, kinematics=kin)
    else:
        w = fb.Cessna172Xv2World(n, kinematics=kin)
    fb.f_init(w, fb.TrimParameters(h_e=h, EAS=EAS, flaps=flaps))
    assert w.trim_success.all()
    fb.f_ode(w)
    assert (w.status == 0).all()
    x, y, ts = w.x, w.y, w.trim_state
    lon, lat = rf.trim_point_rows(ts, x, y)
    # every copy equals the first one bit for bit (lane independence), then the 28 against the reference
    for a in (lon, lat, y[[rf.Y_GAMMA, rf.Y_VD]]):
        assert np.array_equal(a.reshape(a.shape[0], copies, 28), np.broadcast_to(a[:, None, :28], (a.shape[0], copies, 28)))
    if model == "Xv2":   # actuator positions are states here: they must equal the TrimState's commands (assign!, c172x.jl:296-323)
        act = K["FB_X2_ACT"]
        assert np.array_equal(x[act + 0], ts[3]) and np.array_equal(x[act + 1], ts[4]) and np.array_equal(x[act + 2], ts[5]) and np.array_equal(x[act + 3], ts[6])
    with capsys.disabled():
        worst = rf.compare_with_stored(lon[:, :28], lat[:, :28], y[:, :28], TOL, log=lambda s: print(f"\n[HIP {model} {kin}] " + s))
    assert max(worst.values()) <= TOL
    w.close()
