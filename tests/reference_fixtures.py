"""Numbers the REFERENCE itself holds for the hot path (test infrastructure; no oracle, no product code in here).

1. The trim solutions of its autopilot design sweep. `generate_lookups` (lib/FlightApps/design/c172/c172x_design.jl:87-130)
   trims `Cessna172Xv0(NED)` with NLopt's BOBYQA at every node of EAS = 25:5:55 m/s x h = 50:1000:3050 m,
       design_point = C172.TrimParameters(; Ob = Geographic(LatLon(), HEllip(h)), EAS, flaps = flaps_schedule(EAS))     (:107-112)
   and stores `x_trim = lss.x0`, `u_trim = lss.u0`, `z_trim = lss.y0[z_labels]` of the linearised model next to the gains
   (:151-160, :216 for te2te; :549-671 lateral; written by save_lookup_data, lib/FlightPhysics/src/control.jl:855-877).
   Those files ship byte-identical in flight.jl_amd/data/c172x_ctl/ (hashes: tests/golden/reference_data_sha256.json, taken
   from the reference by tests/golden/hash_reference_data.py).
   Row meaning (the label lists of the design script, which it asserts against C172XControl's types):
       te2te  x_trim (q, θ, EAS, α, α_filt, n_eng, thr_p, ele_p)        u_trim = z_trim = (throttle_cmd, elevator_cmd)
       tv2te  x_trim as te2te                                             z_trim = (throttle_cmd, EAS)
       vh2te  x_trim (q, θ, EAS, α, h, α_filt, n_eng, thr_p, ele_p)      z_trim = (EAS, h)
       ar2ar  x_trim (p, r, φ, EAS, β, β_filt, ail_p, rud_p)             u_trim = z_trim = (aileron_cmd, rudder_cmd)
       φβ2ar  x_trim as ar2ar                                             z_trim = (φ, β)
   where the values are those of YStateSpace at the trim point (lib/FlightApps/src/c172/c172x/c172x.jl:354-371,407-448):
   θ, φ = e_nb; α, β = aero.α, aero.β; EAS = airflow.EAS; n_eng = engine.n; *_p = actuator positions; p, q, r = ω_eb_b.

2. The printed linearisation of Robot2D.Vehicle: tests/golden/robot2d_linearization.json (see that test).
"""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HASHES = os.path.join(ROOT, "tests", "golden", "reference_data_sha256.json")
CTL_DIR = os.path.join(ROOT, "flight.jl_amd", "data", "c172x_ctl")

EAS_RANGE = np.arange(25.0, 55.0 + 1e-9, 5.0)          # range(25, 55, length = 7), c172x_design.jl:88 ("#7")
H_RANGE = np.arange(50.0, 3050.0 + 1e-9, 1000.0)       # range(50, 3050, length = 4), :89 ("#4")


def assert_shipped_copy_is_the_references(rel_path: str) -> None:
    """sha256 of the shipped data file == the sha256 hash_reference_data.py took from the reference's file."""
    rec = json.load(open(HASHES))[rel_path]
    with open(os.path.join(ROOT, rel_path), "rb") as f:
        blob = f.read()
    assert len(blob) == rec["bytes"] and hashlib.sha256(blob).hexdigest() == rec["sha256"], \
        f"{rel_path} is not the reference's {rec['reference']}"


def flaps_schedule(EAS):
    """C172XControl.flaps_schedule, lib/FlightApps/src/c172/c172x/control/c172x_ctl.jl:18-24"""
    EAS = np.asarray(EAS, dtype=np.float64)
    return np.where(EAS < 30.0, 1.0, np.where(EAS > 35.0, 0.0, 1.0 - (EAS - 30.0) / 5.0))


def design_nodes():
    """EAS, h, flaps of the 28 design points, flattened EAS fastest (Iterators.product(EAS_range, h_range), :106):
    node k = iE + 7 iH, matching the [.., iE, iH] layout of the stored arrays."""
    E, H = np.meshgrid(EAS_RANGE, H_RANGE, indexing="ij")
    EAS, h = E.ravel(order="F"), H.ravel(order="F")
    return EAS, h, flaps_schedule(EAS)


def stored(name: str) -> dict:
    """x_trim / u_trim / z_trim / K_fbk / K_fwd / K_int of one LQR lookup, node index last ([rows, 28] / [.., .., 28])."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd", "flightbatch"))
    import hdf5_min
    assert_shipped_copy_is_the_references(f"flight.jl_amd/data/c172x_ctl/{name}.h5")
    d = hdf5_min.read_all(os.path.join(CTL_DIR, name + ".h5"))
    assert np.array_equal(d["bounds"], np.array([[25.0, 50.0], [55.0, 3050.0]]))
    out = {}
    for k in ("x_trim", "u_trim", "z_trim", "K_fbk", "K_fwd", "K_int"):
        a = d["data/" + k]
        assert a.shape[-2:] == (7, 4)
        out[k] = a.reshape(a.shape[:-2] + (28,), order="F")
    return out


# rows of the 174-double output record (include/flightbatch.h FB_Y_*) the comparison reads
Y_THETA, Y_PHI = 1, 2
Y_P, Y_Q, Y_R = 28, 29, 30
Y_VD = 36
Y_GAMMA = 39
Y_EAS = 40 + 20
Y_ALPHA, Y_BETA = 62, 63


def trim_point_rows(ts, x, y, x_alpha_filt=0, x_beta_filt=1, n_eng_row=2):
    """The reference's stored rows rebuilt from a trim solution: ts [7, n] = TrimState (α_a, φ_nb, n_eng, throttle, aileron,
    elevator, rudder — c172.jl:796-804), x the trimmed continuous state, y the output record of f_ode! at it.
    Actuator positions equal their commands at a trim point (assign!, lib/FlightApps/src/c172/c172x/c172x.jl:296-323), and
    the commands are the TrimState's."""
    lon = np.stack([y[Y_Q], y[Y_THETA], y[Y_EAS], y[Y_ALPHA], x[x_alpha_filt], ts[n_eng_row], ts[3], ts[5]])
    lat = np.stack([y[Y_P], y[Y_R], y[Y_PHI], y[Y_EAS], y[Y_BETA], x[x_beta_filt], ts[4], ts[6]])
    return lon, lat


LON_LABELS = ("q", "θ", "EAS", "α", "α_filt", "n_eng", "thr_p", "ele_p")
LAT_LABELS = ("p", "r", "φ", "EAS", "β", "β_filt", "ail_p", "rud_p")


def compare_with_stored(lon, lat, y, tol, log=print):
    """|rebuilt − stored| <= tol for every stored trim row of the five LQR lookups; returns the per-label maxima."""
    te, tv, vh, ar, pb = (stored(n) for n in ("te2te", "tv2te", "vh2te", "ar2ar", "phibeta2ar"))
    EAS, h, _ = design_nodes()
    worst = {}

    def hold(label, got, want):
        d = np.abs(got - want).max()
        worst[label] = max(worst.get(label, 0.0), d)
        assert d <= tol, f"{label}: |ours − reference| = {d:.3e} > {tol:g} at node {int(np.abs(got - want).argmax())}"

    for k, lab in enumerate(LON_LABELS):
        hold(lab, lon[k], te["x_trim"][k])
        hold(lab, lon[k], tv["x_trim"][k])
        hold(lab, lon[k], vh["x_trim"][k if k < 4 else k + 1])
    hold("h", h, vh["x_trim"][4])
    for k, lab in enumerate(LAT_LABELS):
        hold(lab, lat[k], ar["x_trim"][k])
        hold(lab, lat[k], pb["x_trim"][k])
    # u_trim / z_trim: commands = positions at trim; tracked outputs
    hold("thr_p", lon[6], te["u_trim"][0]); hold("ele_p", lon[7], te["u_trim"][1])
    hold("thr_p", lon[6], te["z_trim"][0]); hold("ele_p", lon[7], te["z_trim"][1])
    hold("thr_p", lon[6], tv["z_trim"][0]); hold("EAS", lon[2], tv["z_trim"][1])
    hold("EAS", lon[2], vh["z_trim"][0]); hold("h", h, vh["z_trim"][1])
    hold("ail_p", lat[6], ar["u_trim"][0]); hold("rud_p", lat[7], ar["u_trim"][1])
    hold("ail_p", lat[6], ar["z_trim"][0]); hold("rud_p", lat[7], ar["z_trim"][1])
    hold("φ", lat[2], pb["z_trim"][0]); hold("β", lat[4], pb["z_trim"][1])
    # the trim constraints themselves: level flight (γ_wb_n = 0 in still air = ground flight-path angle), no climb
    hold("γ", y[Y_GAMMA], 0.0); hold("climb_rate", -y[Y_VD], 0.0)
    log("max |ours − reference's stored trim| over the 28 design points: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    return worst
