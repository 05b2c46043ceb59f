"""Cessna172Xv2 (fly-by-wire actuators + gain-scheduled control laws) on the GPU, through the C ABI, against the CPU oracle
and against the reference's closed-loop tolerances (lib/FlightApps/test/c172/test_c172x1.jl)."""
import os
import sys

import numpy as np
import pytest

import conditioning

from oracle_binding import OracleX
from test_gpu_parity import lattice_trim_params, state_scale

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_to_dev_rows(K):
    """row k of the C ABI state (reference order) -> row of the oracle / device order (27 Sv0 rows, then the actuators)"""
    return np.array([k if k < K["FB_X2_ACT"] else (27 + k - K["FB_X2_ACT"] if k < K["FB_X2_KIN"] else k - K["FB_NACT"]) for k in range(34)])


def x_scale(x):
    sc = np.ones_like(x)
    sc[:27] = state_scale(x[:27])
    return sc


@pytest.fixture(scope="module")
def gains(fb):
    return fb.ctl_gains.ctl_gains_blob()


def make_pair(fb, oracle, gains, n, seed, dt=0.01, ratio=2):
    """The same randomised-trim batch initialised on the GPU and on the oracle."""
    tp = lattice_trim_params(fb, n, seed=seed)
    w = fb.Cessna172Xv2World(n, gains=gains)
    sim = fb.Simulation(w, dt=dt, Δt=dt * ratio, save_on=False, steps_per_launch=50)
    fb.init(sim, tp)
    X = OracleX(oracle, gains)
    env = oracle.default_env()
    o = X.trim_init(tp.pack(n), fb.TrimState(n), env, dt * ratio)
    o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
    return w, sim, X, env, o


def test_x2_init_matches_oracle(fb, oracle, gains):
    K = fb.K
    n = 1024
    w, sim, X, env, o = make_pair(fb, oracle, gains, n, seed=21)
    perm = ref_to_dev_rows(K)
    ok = w.trim_success & o["ok"]
    assert (w.trim_success == o["ok"]).all() and ok.mean() > 0.6
    xo = o["x"][perm]                             # oracle rows re-ordered to the reference order (w.x is in that order)
    assert np.max(np.abs(w.x - xo)[:, ok] / x_scale(o["x"])[perm][:, ok]) < 1e-7
    # actuator states = commands = trim values; control-law inputs aligned with the vehicle; both channels in direct
    assert np.array_equal(w.x[K["FB_X2_ACT"] + K["FB_ACT_THROTTLE"]], w.ctl.y("THROTTLE_CMD"))
    assert np.max(np.abs(w.cu - o["cu"])[:, ok]) < 1e-6 and np.max(np.abs(w.cs - o["cs"])[:, ok]) < 1e-6
    assert (w.ctl.y("LON_MODE") == 0).all() and (w.ctl.y("LAT_MODE") == 0).all() and (w.ctl.y("H_STATE") == K["FB_ALT_HOLD"]).all()
    # f_ode!: ẋ (34 rows, reference order) and y against the oracle evaluated at the GPU's own state
    xd = np.zeros((34, n)); fb.f_ode(w, xd)
    od = dict(o); od["x"] = np.ascontiguousarray(np.empty((34, n))); od["x"][perm] = w.x; od["cs"] = w.cs; od["u"] = w.u; od["ui"] = w.ui; od["s"] = w.s
    xdo, yo, _ = X.f_ode(od, env)
    err = np.abs(xd - xdo[perm]) / np.maximum(np.abs(xdo[perm]), 1.0)
    assert err.max() < 1e-9, (err.max(), np.unravel_index(err.argmax(), err.shape))
    sc_y = np.maximum(np.abs(yo), 1.0); sc_y[22:25] = 6.4e6
    assert (np.abs(w.y - yo) / sc_y).max() < 1e-9
    w.close()


@pytest.mark.parametrize("same_grid", ["1", "0"])
def test_x2_closed_loop_trajectory_matches_oracle(fb, oracle, gains, same_grid, monkeypatch):
    """README example 2 configuration (dt = 0.01, Δt = 0.02): every aircraft in its own pair of control modes with its own
    references, 10 s; state, control-law record and status against the oracle at the north-star tolerance. Both forms of the
    gain lookup inside the stepping kernel: the cell located once per update (the ten lookups of the reference's data share one
    grid: CtlOffsets::same_grid) and once per lookup from its own header (FLIGHTBATCH_CTL_SAME_GRID=0, read when the blob is set)."""
    monkeypatch.setenv("FLIGHTBATCH_CTL_SAME_GRID", same_grid)
    K = fb.K
    n = 2048
    w, sim, X, env, o = make_pair(fb, oracle, gains, n, seed=22)
    rng = np.random.default_rng(7)
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, n)
    cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, n)
    cu[K["FB_CU_Q_REF"]] += rng.uniform(-0.005, 0.005, n); cu[K["FB_CU_THETA_REF"]] += rng.uniform(-0.03, 0.03, n)
    cu[K["FB_CU_EAS_REF"]] += rng.uniform(-3, 3, n); cu[K["FB_CU_CLM_REF"]] += rng.uniform(-1.5, 1.5, n)
    cu[K["FB_CU_H_REF"]] += rng.choice([-60.0, -5.0, 0.0, 5.0, 60.0], n)
    cu[K["FB_CU_P_REF"]] += rng.uniform(-0.01, 0.01, n); cu[K["FB_CU_BETA_REF"]] += rng.uniform(-0.03, 0.03, n)
    cu[K["FB_CU_PHI_REF"]] += rng.uniform(-0.3, 0.3, n); cu[K["FB_CU_CHI_REF"]] += rng.uniform(-0.5, 0.5, n)
    w.cu = cu
    o["cu"] = np.ascontiguousarray(o["cu"]); o["cu"][:] = cu
    # start both from the GPU's initial condition so that only the stepping is compared
    perm = ref_to_dev_rows(K)
    o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
    fb.step(sim, 10.0); w.sync()
    X.step(o, env, 0.01, 2, 1000)
    st, sto = w.status, o["status"]
    assert np.array_equal(st != 0, sto != 0)
    ok = (st == 0)
    assert ok.mean() > 0.9
    xo = o["x"][perm]
    sc = x_scale(o["x"])[perm]
    err = (np.abs(w.x - xo) / sc)[:, ok]
    print("X2 closed loop, max scaled state error after 1000 steps:", err.max(), "terminated:", int((~ok).sum()))
    assert err.max() < 1e-6
    cerr = np.abs(w.cs - o["cs"])[:, ok] / np.maximum(np.abs(o["cs"][:, ok]), 1.0)
    print("max control-law record error:", cerr.max())
    assert cerr.max() < 1e-6
    assert np.array_equal(w.ctl.y("LON_MODE")[ok], o["cs"][K["FB_CS_LON_MODE"], ok]) and np.array_equal(w.s[:, ok], o["s"][:, ok])
    w.close()


def test_x2_reference_mode_tracking(fb, gains):
    """test_c172x1.jl:300-470 on the GPU, one scenario per lane, dt = Δt = 0.01 like the reference's test:
    thr_θ (θ_ref = 5°, atol 1e-4), thr_EAS (45 m/s, 1e-1), EAS_clm (2 m/s & 45 m/s), φ_β (φ = π/12, β = 3°, 1e-3)."""
    K = fb.K
    n = 256
    w = fb.Cessna172Xv2World(n, gains=gains)
    sim = fb.Simulation(w, dt=0.01, Δt=0.01, save_on=False, steps_per_launch=1)
    fb.init(sim, fb.TrimParameters())
    assert w.trim_success.all()
    lane = np.arange(n) % 4
    w.ctl.lon.mode_req = np.choose(lane, [ModeLon(fb).thr_θ, ModeLon(fb).thr_EAS, ModeLon(fb).EAS_clm, ModeLon(fb).sas]).astype(float)
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β)
    fb.step(sim, 1.0)
    y = _y(fb, w)
    assert np.array_equal(w.ctl.y("LON_MODE"), w.ctl.lon.mode_req) and (w.ctl.y("LAT_MODE") == fb.ModeControlLat.φ_β).all()
    w.ctl.lat.φ_ref = np.where(lane == 3, np.pi / 12, np.pi / 6)
    w.ctl.lat.β_ref = np.where(lane == 3, np.deg2rad(3), 0.0)
    w.ctl.lon.θ_ref = np.where(lane == 0, np.deg2rad(5), w.ctl.lon.θ_ref)
    w.ctl.lon.EAS_ref = np.where((lane == 1) | (lane == 2), 45.0, w.ctl.lon.EAS_ref)
    w.ctl.lon.clm_ref = np.where(lane == 2, 2.0, w.ctl.lon.clm_ref)
    fb.step(sim, 10.0); w.sync()
    y = _y(fb, w)
    assert np.all(np.abs(y[1, lane == 0] - np.deg2rad(5)) < 1e-4)                    # thr_θ after 10 s
    assert np.all(np.abs(y[2, lane == 3] - np.pi / 12) < 1e-3) and np.all(np.abs(y[K["FB_Y_AERO"] + 1, lane == 3] - np.deg2rad(3)) < 1e-3)
    fb.step(sim, 20.0); w.sync()
    y = _y(fb, w)
    assert np.all(np.abs(y[K["FB_Y_AIR"] + 20, lane == 1] - 45) < 1e-1)               # thr_EAS after 30 s
    assert np.all(np.abs(y[K["FB_Y_KIN"] + 36, lane == 2] + 2) < 1e-1) and np.all(np.abs(y[K["FB_Y_AIR"] + 20, lane == 2] - 45) < 2e-1)
    assert (w.status == 0).all()
    w.close()


def ModeLon(fb):
    return fb.ModeControlLon


def _y(fb, w):
    fb.f_ode(w)
    return w.y


def test_checkpoint_restore_resumes_bitwise(fb, gains):
    """Checkpoint in the middle of a control-law period (odd step count at Δt = 2 dt), keep going, restore into a NEW world
    and repeat: identical bits. Same for Cessna172Sv0 and an fp32 Robot2D batch."""
    import io
    n = 512
    tp = lattice_trim_params(fb, n, seed=5)

    def exercise(make, prepare, T0, T1, **simkw):
        w = make(); sim = fb.Simulation(w, save_on=False, **simkw); prepare(w, sim)
        fb.step(sim, T0)
        ck = fb.checkpoint(sim)
        buf = io.BytesIO(); np.savez(buf, **ck); buf.seek(0); ck = dict(np.load(buf))      # survives serialisation
        fb.step(sim, T1); w.sync()
        a = w.checkpoint()
        w2 = make(); sim2 = fb.Simulation(w2, save_on=False, **simkw)
        fb.restore(sim2, ck)
        assert sim2.t == pytest.approx(T0)
        fb.step(sim2, T1); w2.sync()
        b = w2.checkpoint()
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        w.close(); w2.close()

    def prep_x2(w, sim):
        fb.init(sim, tp)
        w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 1.0
        w.ctl.lat.mode_req = float(fb.ModeControlLat.χ_β); w.ctl.lat.χ_ref = 0.3

    exercise(lambda: fb.Cessna172Xv2World(n, gains=gains), prep_x2, 0.37, 1.0, dt=0.01, Δt=0.02, steps_per_launch=50)
    exercise(lambda: fb.Cessna172Xv2World(n, gains=gains, kinematics="NED"), prep_x2, 0.37, 1.0, dt=0.01, Δt=0.02, steps_per_launch=50)   # 31 rows
    exercise(lambda: fb.BatchedWorld(n), lambda w, sim: fb.init(sim, tp), 0.37, 1.0, dt=0.01, steps_per_launch=50)

    def prep_r2(w, sim):
        fb.init(sim, fb.InitParameters(u_m=0.05))
        u = w.u; u[0] = 2; u[3] = 1.0; w.u = u

    exercise(lambda: fb.Robot2DWorld(n, dtype="f32"), prep_r2, 0.37, 1.0, dt=0.01, Δt=0.02, steps_per_launch=50)


def test_x2_segment_guidance_matches_oracle(fb, oracle, gains):
    """Segment guidance (c172x_gdc.jl:232-329) in the loop: every aircraft gets its own target segment (left / right of its
    course, above / below), horizontal and vertical guidance requested; 30 s against the oracle, then the same run continued
    to 120 s must have captured the segment."""
    from test_oracle_c172x import seg_end
    K = fb.K
    n = 512
    w, sim, X, env, o = make_pair(fb, oracle, gains, n, seed=23)
    perm = ref_to_dev_rows(K)
    fb.f_ode(w)
    y = w.y
    rng = np.random.default_rng(11)
    p1 = np.zeros((3, n)); p2 = np.zeros((3, n))
    for i in range(n):
        ob = np.array([y[K["FB_Y_KIN"] + 15, i], y[K["FB_Y_KIN"] + 16, i], y[K["FB_Y_KIN"] + 20, i]])
        chi = y[K["FB_Y_KIN"] + 38, i]
        p1[:, i] = seg_end(oracle, ob, rng.uniform(100, 600), chi + rng.choice([-1, 1]) * np.pi / 2, rng.uniform(-40, 40))
        p2[:, i] = seg_end(oracle, p1[:, i], 3e4, chi + rng.uniform(-0.3, 0.3), rng.uniform(-100, 100))
    w.ctl.set_target(p1, p2)
    w.ctl.gdc.mode_req = float(fb.ModeGuidance.segment); w.ctl.gdc.hor_gdc_req = 1.0; w.ctl.gdc.vrt_gdc_req = 1.0
    o["cu"] = np.ascontiguousarray(w.cu); o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
    fb.step(sim, 30.0); w.sync()
    X.step(o, env, 0.01, 2, 3000)
    ok = (w.status == 0) & (o["status"] == 0) & w.trim_success
    assert ok.mean() > 0.6 and np.array_equal(w.status != 0, o["status"] != 0)
    err = (np.abs(w.x - o["x"][perm]) / x_scale(o["x"])[perm])[:, ok]
    cerr = (np.abs(w.cs - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0))[:, ok]
    uerr = (np.abs(w.cu - o["cu"]) / np.maximum(np.abs(o["cu"]), 1.0))[:, ok]
    print("guidance closed loop after 3000 steps: state", err.max(), "record", cerr.max(), "inputs", uerr.max())
    assert err.max() < 1e-6 and cerr.max() < 1e-6 and uerr.max() < 1e-6
    assert (w.ctl.y("GDC_MODE")[ok] == fb.ModeGuidance.segment).all() and (w.ctl.y("LAT_MODE")[ok] == fb.ModeControlLat.χ_β).all()
    fb.step(sim, 90.0); w.sync()
    good = ok & (w.status == 0)
    e_sb = np.abs(w.ctl.y("SEG_E_SB")[good])
    print("cross-track error after 120 s: median", np.median(e_sb), "max", e_sb.max())
    assert np.median(e_sb) < 2.0 and (e_sb < 20.0).mean() > 0.95
    w.close()


def refine_lookup(blob, offsets, recs, k, along_EAS=True):
    """The blob with lookup k resampled on a finer grid (midpoints inserted along h, and along EAS too if asked): the same
    piecewise-bilinear function on a DIFFERENT grid from the other lookups."""
    o = offsets[k]; nE, nH = int(blob[o]), int(blob[o + 1]); rec = recs[k]
    body = blob[o + 6:o + 6 + nE * nH * rec].reshape(nH, nE, rec)
    if along_EAS:
        wide = np.zeros((nH, 2 * nE - 1, rec))
        wide[:, ::2] = body; wide[:, 1::2] = 0.5 * (body[:, :-1] + body[:, 1:])
        body = wide
    fine = np.zeros((2 * nH - 1, body.shape[1], rec))
    fine[::2] = body; fine[1::2] = 0.5 * (body[:-1] + body[1:])
    hdr = blob[o:o + 6].copy(); hdr[0] = fine.shape[1]; hdr[1] = fine.shape[0]
    return np.concatenate([blob[:o], hdr, fine.reshape(-1), blob[o + 6 + nE * nH * rec:]])


@pytest.mark.parametrize("variant,kin", [("same_grid", "WA"), ("per_lookup_headers", "WA"), ("one_lookup_refined", "WA"), ("same_grid", "ECEF"), ("same_grid", "NED")])
def test_x2_control_laws_fuzz(fb, oracle, gains, variant, kin, monkeypatch):
    """f_periodic!(Unconditional(), world) — guidance + control laws — from 16 384 random controller records: every pair of
    previous / requested modes (so every bumpless-transfer branch), arbitrary compensator states and saturation flags, references
    all over the place, gain lookups inside, on the edge of and outside the (EAS, h) grid; record and inputs against the oracle.
    Variants: the shared-cell lookup (the reference's ten lookups sit on one grid), the per-lookup headers forced on the same blob,
    and a blob whose q2e and v2t lookups are resampled on finer grids (the library must notice that the grids differ; the oracle
    runs on the original blob: the functions are the same. The blob must still fit the 6144 doubles fb_f_periodic stages in LDS). The
    first variant also for Cessna172Xv2(ECEF()) / (NED()): k_x2_ctl<KIN> takes its inputs from that mechanisation's evaluation."""
    K = fb.K
    n = 16384
    dyn0 = K["FB_X2_DYN"] - {"WA": 0, "ECEF": 1, "NED": 3}[kin]      # first row of w_eb_b in the C ABI's state of this mechanisation
    gains_gpu = gains
    if variant == "per_lookup_headers": monkeypatch.setenv("FLIGHTBATCH_CTL_SAME_GRID", "0")
    if variant == "one_lookup_refined":
        recs = [K["FB_CTL_LQR8_REC"], K["FB_CTL_LQR8_REC"], K["FB_CTL_LQR9_REC"]] + [K["FB_CTL_PID_REC"]] * 3 + [K["FB_CTL_LQR8_REC"]] * 2 + [K["FB_CTL_PID_REC"]] * 2
        def offsets_of(b):
            offs, o = [], 0
            for r in recs:
                offs.append(o); o += 6 + int(b[o]) * int(b[o + 1]) * r
            assert o == b.size
            return offs
        gains_gpu = refine_lookup(gains, offsets_of(gains), recs, 3)
        gains_gpu = refine_lookup(gains_gpu, offsets_of(gains_gpu), recs, 5, along_EAS=False)
        assert gains_gpu.size <= 6144
    rng = np.random.default_rng(99)
    tp = fb.TrimParameters(EAS=rng.uniform(36, 56, n), h_e=rng.uniform(100, 3300, n), ψ_nb=rng.uniform(-3, 3, n))
    w = fb.Cessna172Xv2World(n, gains=gains_gpu, kinematics=kin)
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False)
    fb.init(sim, tp)
    x = w.x
    x[dyn0:dyn0 + 3] += rng.normal(0, 0.05, (3, n)); x[dyn0 + 3:] += rng.normal(0, 2.0, (3, n))
    x[K["FB_X2_ACT"]:K["FB_X2_ACT"] + 4] += rng.normal(0, 0.2, (4, n))
    w.set_state(x, w.s)
    cu = w.cu
    cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, n); cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, n)
    for k in ("THROTTLE_AXIS", "THROTTLE_OFFSET", "ELEVATOR_AXIS", "ELEVATOR_OFFSET", "AILERON_AXIS", "AILERON_OFFSET", "RUDDER_AXIS", "RUDDER_OFFSET"):
        cu[K["FB_CU_" + k]] = rng.uniform(-1.3, 1.3, n)
    cu[K["FB_CU_Q_REF"]] = rng.normal(0, 0.05, n); cu[K["FB_CU_THETA_REF"]] = rng.normal(0, 0.2, n); cu[K["FB_CU_EAS_REF"]] = rng.uniform(20, 70, n)
    cu[K["FB_CU_CLM_REF"]] = rng.normal(0, 3, n); cu[K["FB_CU_H_REF"]] += rng.choice([-200.0, -10.5, -9.5, 0.0, 8.9, 11.2, 300.0], n)
    cu[K["FB_CU_P_REF"]] = rng.normal(0, 0.1, n); cu[K["FB_CU_BETA_REF"]] = rng.normal(0, 0.1, n); cu[K["FB_CU_PHI_REF"]] = rng.normal(0, 0.6, n)
    cu[K["FB_CU_CHI_REF"]] = rng.uniform(-7, 7, n)
    cu[K["FB_CU_GDC_MODE_REQ"]] = rng.integers(0, 3, n); cu[K["FB_CU_SEG_HOR_REQ"]] = rng.integers(0, 2, n); cu[K["FB_CU_SEG_VRT_REQ"]] = rng.integers(0, 2, n)
    fb.f_ode(w); y = w.y
    cu[K["FB_CU_SEG_P1"]] = y[K["FB_Y_KIN"] + 15] + rng.normal(0, 2e-4, n); cu[K["FB_CU_SEG_P1"] + 1] = y[K["FB_Y_KIN"] + 16] + rng.normal(0, 2e-4, n)
    cu[K["FB_CU_SEG_P1"] + 2] = y[K["FB_Y_KIN"] + 20] + rng.normal(0, 50, n)
    cu[K["FB_CU_SEG_P2"]] = cu[K["FB_CU_SEG_P1"]] + rng.normal(0, 3e-3, n); cu[K["FB_CU_SEG_P2"] + 1] = cu[K["FB_CU_SEG_P1"] + 1] + rng.normal(0, 3e-3, n)
    cu[K["FB_CU_SEG_P2"] + 2] = cu[K["FB_CU_SEG_P1"] + 2] + rng.normal(0, 100, n)
    cs = rng.normal(0, 0.3, (K["FB_NCS"], n))
    cs[K["FB_CS_LON_MODE"]] = rng.integers(0, 9, n); cs[K["FB_CS_LAT_MODE"]] = rng.integers(0, 5, n); cs[K["FB_CS_H_STATE"]] = rng.integers(0, 2, n)
    for blk, cnt in (("TE2TE", 2), ("TV2TE", 2), ("VH2TE", 2), ("AR2AR", 2), ("PHIBETA2AR", 2)):
        cs[K["FB_CS_" + blk] + 2:K["FB_CS_" + blk] + 4] = rng.integers(-1, 2, (2, n))
    for blk, off in (("Q2E_INT", 1), ("P2PHI_INT", 1), ("Q2E_PID", 2), ("C2THETA_PID", 2), ("V2T_PID", 2), ("P2PHI_PID", 2), ("CHI2PHI_PID", 2)):
        cs[K["FB_CS_" + blk] + off] = rng.integers(-1, 2, n)
    for k in ("THROTTLE_CMD",):
        cs[K["FB_CS_" + k]] = rng.uniform(0, 1, n)
    w.cu = cu; w.cs = cs
    perm = abi_to_dev_rows(K, kin)
    X = OracleX(oracle, gains)
    st = dict(x=np.zeros((34, n)), u=w.u, ui=w.ui, s=w.s, cu=cu.copy(), cs=cs.copy())
    st["x"][perm] = w.x
    fb.f_periodic(w); w.sync()
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        X.f_periodic(st, oracle.default_env(), 0.02)
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    dcs = np.abs(w.cs - st["cs"]) / np.maximum(np.abs(st["cs"]), 1.0)
    dcu = np.abs(w.cu - st["cu"]) / np.maximum(np.abs(st["cu"]), 1.0)
    print("control-law fuzz: record %.2e (row %d), inputs %.2e" % (dcs.max(), dcs.max(1).argmax(), dcu.max()))
    assert dcs.max() < 1e-9 and dcu.max() < 1e-9
    for row in ("LON_MODE", "LAT_MODE", "H_STATE", "GDC_MODE", "SEG_HOR_GDC", "SEG_VRT_GDC"):
        assert np.array_equal(w.cs[K["FB_CS_" + row]], st["cs"][K["FB_CS_" + row]]), row
    w.close()


def abi_to_dev_rows(K, kin):
    """row k of the C ABI state of Cessna172Xv2(kinematics) -> row of the oracle / device order, in which every mechanisation keeps the
    nine kinematic rows 12..20 and leaves the ones it does not use at zero (ECEF: row 20; NED: rows 18-20)"""
    unused = {"WA": (), "ECEF": (20,), "NED": (18, 19, 20)}[kin]
    return np.array([r for r in ref_to_dev_rows(K) if r not in unused])


@pytest.mark.parametrize("kin", ["WA", "ECEF", "NED"])
def test_x2_ecef_and_ned_mechanisations(fb, oracle, gains, kin):
    """Cessna172Xv2(ECEF()) / Cessna172Xv2(NED()) (FA/c172/c172x/c172x2.jl:57-59 over FP/kinematics.jl:250-425) against the oracle
    with the same mechanisation: f_init! (trim, actuators, control laws), f_ode!, ten seconds of closed-loop flight with every aircraft
    in its own pair of modes, and — for the ground-capable instance of that mechanisation — steep autopilot descents onto a runway:
    hard landings that end in GroundCrash and softer ones that roll out, status word, step and place of every termination included;
    the descent is run twice and must repeat bit for bit (what a spill-placement fault of the compiler would break: docs/design/k_step_air.md)."""
    K = fb.K
    perm = abi_to_dev_rows(K, kin)
    nx = 34 - {"WA": 0, "ECEF": 1, "NED": 3}[kin]
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        # ---- f_init!, f_ode!
        n = 1024
        tp = lattice_trim_params(fb, n, seed=41)
        w = fb.Cessna172Xv2World(n, gains=gains, kinematics=kin)
        sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
        fb.init(sim, tp)
        X = OracleX(oracle, gains)
        env = oracle.default_env()
        o = X.trim_init(tp.pack(n), fb.TrimState(n), env, 0.02)
        o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
        ok = w.trim_success & o["ok"]
        assert w.x.shape[0] == nx and perm.size == nx and (w.trim_success == o["ok"]).all() and ok.mean() > 0.6
        unused = np.setdiff1d(np.arange(34), perm)
        assert (o["x"][unused] == 0).all()
        sc = x_scale(o["x"])[perm]
        assert np.max(np.abs(w.x - o["x"][perm])[:, ok] / sc[:, ok]) < 1e-7
        assert np.max(np.abs(w.cu - o["cu"])[:, ok]) < 1e-6 and np.max(np.abs(w.cs - o["cs"])[:, ok]) < 1e-6
        xd = np.zeros((nx, n)); fb.f_ode(w, xd)
        od = dict(o); od["x"] = np.zeros((34, n)); od["x"][perm] = w.x; od["cs"] = w.cs; od["u"] = w.u; od["ui"] = w.ui; od["s"] = w.s
        xdo, yo, _ = X.f_ode(od, env)
        assert (xdo[unused] == 0).all()
        assert (np.abs(xd - xdo[perm]) / np.maximum(np.abs(xdo[perm]), 1.0)).max() < 1e-9
        sc_y = np.maximum(np.abs(yo), 1.0); sc_y[22:25] = 6.4e6
        assert (np.abs(w.y - yo) / sc_y).max() < 1e-9
        # ---- closed loop, airborne: modes and references per aircraft
        rng = np.random.default_rng(8)
        cu = w.cu
        cu[K["FB_CU_LON_MODE_REQ"]] = rng.integers(0, 9, n)
        cu[K["FB_CU_LAT_MODE_REQ"]] = rng.integers(0, 5, n)
        cu[K["FB_CU_THETA_REF"]] += rng.uniform(-0.03, 0.03, n); cu[K["FB_CU_EAS_REF"]] += rng.uniform(-3, 3, n)
        cu[K["FB_CU_CLM_REF"]] += rng.uniform(-1.5, 1.5, n); cu[K["FB_CU_H_REF"]] += rng.choice([-60.0, -5.0, 0.0, 5.0, 60.0], n)
        cu[K["FB_CU_PHI_REF"]] += rng.uniform(-0.3, 0.3, n); cu[K["FB_CU_CHI_REF"]] += rng.uniform(-0.5, 0.5, n)
        cu[K["FB_CU_BETA_REF"]] += rng.uniform(-0.03, 0.03, n)
        w.cu = cu
        o["cu"] = np.ascontiguousarray(cu.copy())
        o["x"][:] = 0; o["x"][perm] = w.x; o["cs"] = w.cs; o["u"] = w.u; o["ui"] = w.ui; o["s"] = w.s
        fb.step(sim, 10.0); w.sync()
        X.step(o, env, 0.01, 2, 1000)
        st, sto = w.status, o["status"]
        assert np.array_equal(st, sto)
        fine = st == 0
        assert fine.mean() > 0.9
        err = (np.abs(w.x - o["x"][perm]) / x_scale(o["x"])[perm])[:, fine]
        cerr = (np.abs(w.cs - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0))[:, fine]
        print(f"Xv2({kin}) closed loop after 1000 steps: max scaled state error {err.max():.2e}, control-law record {cerr.max():.2e}")
        assert err.max() < 1e-6 and cerr.max() < 1e-6 and np.array_equal(w.s[:, fine], o["s"][:, fine])
        w.close()
        # ---- the ground-capable instance: autopilot descents onto the runway
        from test_gpu_termination import geoid
        n = 2048
        rng = np.random.default_rng(61)
        N0 = geoid(oracle, np.zeros(1), np.zeros(1))[0]
        tp = fb.TrimParameters(EAS=rng.uniform(42.0, 55.0, n), h_e=N0 + 2.0 + rng.uniform(15.0, 50.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n), γ_wb_n=-0.05)
        runs = []
        for rep in range(2):
            w = fb.Cessna172Xv2World(n, gains=gains, kinematics=kin)
            sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
            fb.init(sim, tp)
            assert w.trim_success.all()
            cu = w.cu
            cu[K["FB_CU_LON_MODE_REQ"]] = float(fb.ModeControlLon.EAS_clm)
            cu[K["FB_CU_LAT_MODE_REQ"]] = float(fb.ModeControlLat.φ_β)
            cu[K["FB_CU_CLM_REF"]] = -np.where(np.arange(n) % 2 == 0, rng.uniform(7.0, 14.0, n), rng.uniform(1.0, 3.0, n)) if rep == 0 else runs[0]["clm"]
            w.cu = cu
            start = dict(x=w.x, cs=w.cs, u=w.u, ui=w.ui, s=w.s, cu=cu.copy())
            fb.step(sim, 12.0); w.sync()
            runs.append(dict(x=w.x, cs=w.cs, s=w.s, status=w.status, term=w.termination, clm=cu[K["FB_CU_CLM_REF"]].copy(), start=start))
            w.close()
        a, b = runs
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["cs"], b["cs"]) and np.array_equal(a["status"], b["status"]), "not reproducible"
        o = X.trim_init(tp.pack(n), fb.TrimState(n), env, 0.02)
        o["status"] = np.zeros(n, np.int32); o["nstep"] = 0
        s0 = a["start"]
        o["cu"] = np.ascontiguousarray(s0["cu"]); o["x"][:] = 0; o["x"][perm] = s0["x"]; o["cs"] = s0["cs"]; o["u"] = s0["u"]; o["ui"] = s0["ui"]; o["s"] = s0["s"]
        o_start = {k: np.array(v, copy=True) for k, v in o.items() if isinstance(v, np.ndarray)}
        X.step_term(o, env, 0.01, 2, 1200)
        # the oracle against itself (tests/conditioning.py): v_eb_b nudged once per aircraft as it comes within wheel reach of the runway
        h_row_o = {"WA": 20, "ECEF": 19, "NED": 17}[kin]
        pert_ulp = conditioning.x2_perturbed_runs(X, o_start, env, 1200, h_row_o, N0, None, K=2, seed=1, threads=16)
        pert_rel = conditioning.x2_perturbed_runs(X, o_start, env, 1200, h_row_o, N0, 1e-12, K=4, jitter=conditioning.ULP_R, seed=2, threads=16)
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    st, sto = a["status"], o["status"]
    term = sto != 0
    print(f"Xv2({kin}) descents: {int(term.sum())} of {n} crashed; status words {np.unique(sto)}, places {np.unique(o['term_where'])}")
    assert np.array_equal(st, sto) and term.sum() >= 300 and (~term).sum() >= 300
    tstep, twhere = a["term"]
    assert np.array_equal(twhere, o["term_where"]) and np.array_equal(tstep, o["term_step"])
    xo = o["x"][perm]
    err = np.abs(a["x"] - xo) / x_scale(o["x"])[perm]
    he_row = int(np.where(perm == {"WA": 20, "ECEF": 19, "NED": 17}[kin])[0][0])
    flying = ~term & (xo[he_row] - N0 > 8.0)
    rolling = ~term & ~flying
    print(f"Xv2({kin}): max scaled state error, crashed {err[:, term].max():.2e} | still flying {err[:, flying].max() if flying.any() else 0.0:.2e} "
          f"({int(flying.sum())}) | rolling {err[:, rolling].max() if rolling.any() else 0.0:.2e} ({int(rolling.sum())})")
    assert err[:, term].max() < 1e-6                      # (tolerances: see test_x2_crash_under_autopilot)
    assert not flying.any() or err[:, flying].max() < 1e-6
    # The survivors have spent up to ten seconds bouncing and rolling under an autopilot that still demands a descent: stick-slip on six
    # friction regulators (device rows 2-7: integrators with k_i = 400 1/s behind a sign-tested anti-windup halt, landinggear.jl:411-476).
    # How ill-conditioned that is, on these very aircraft, is MEASURED (tests/conditioning.py): the oracle is run again with v_eb_b of every
    # aircraft nudged once as it reaches the runway — by one ulp (quoted in the log), and by 1e-12 (the GPU's own distance from the oracle
    # after an airborne approach) with the altitude jittered by one ulp of the geocentric radius per step, the resolution of the contact
    # geometry — and the GPU is held, aircraft by aircraft, to max(1e-6, 10 x |oracle − oracle'|): an aircraft whose envelope is below
    # 1e-7 holds the north star's 1e-6. State and control-law record together.
    cerr = np.abs(a["cs"] - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)
    assert cerr[:, term | flying].max() < 1e-6

    def lane_err(xx, cc):   # per-aircraft max scaled distance from the nominal oracle run, state rows (oracle order) and control-law record
        return np.maximum((np.abs(xx - o["x"]) / x_scale(o["x"])).max(0), (np.abs(cc - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)).max(0))
    assert rolling.sum() >= 100
    per_lane = np.maximum(err.max(0), cerr.max(0))[rolling]
    E_ulp = np.stack([lane_err(p["x"], p["cs"])[rolling] for p in pert_ulp])
    E_rel = np.stack([lane_err(p["x"], p["cs"])[rolling] for p in pert_rel])
    assert all(p["nudged"][rolling].mean() > 0.95 for p in pert_ulp + pert_rel)   # (the rest came within 8 m of the runway in the last steps only)
    print(f"Xv2({kin}) rolling ({int(rolling.sum())} aircraft): oracle vs oracle' with ONE ULP on v_eb_b at touchdown: per-aircraft quantiles 50/90/99/100 % "
          f"{np.quantile(E_ulp.ravel(), [0.5, 0.9, 0.99, 1.0])}; beyond 1e-6: {int((E_ulp.max(0) > 1e-6).sum())}, beyond 1e-4: {int((E_ulp.max(0) > 1e-4).sum())}")
    conditioning.check_against_envelope(per_lane, E_rel, f"Xv2({kin}) rolling")
    assert np.array_equal(a["s"], o["s"])


@pytest.mark.parametrize("kin", ["ECEF", "NED"])
def test_x2_mechanisations_device_log(fb, gains, kin):
    """The on-device TimeSeries log of Cessna172Xv2(ECEF()) / (NED()): 33 / 31 state rows in the reference's order, equal to what the
    same run gives when it is stopped by hand at the same instants (cb_save, FC/sim.jl:210-217)."""
    n = 512
    nx = {"ECEF": 33, "NED": 31}[kin]
    tp = lattice_trim_params(fb, n, seed=6)
    rows = [fb.K["FB_Y_KIN"] + 1, fb.K["FB_Y_AIR"] + 20]
    runs = []
    for by_hand in (False, True):
        w = fb.Cessna172Xv2World(n, gains=gains, kinematics=kin)
        sim = fb.Simulation(w, dt=0.01, Δt=0.02, t_end=2.0, saveat=0.5, save_rows=rows, save_on=not by_hand, steps_per_launch=40)
        fb.init(sim, tp)
        w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 1.5
        w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = 0.2
        if not by_hand:
            fb.run(sim)
            ts = fb.TimeSeries(sim)
            assert len(ts) == 5 and ts.x.shape == (5, nx, n) and ts.y.shape == (5, 2, n) and np.array_equal(ts.x[-1], w.x)
            runs.append(ts)
        else:
            ts = runs[0]
            for k in range(5):
                if k:
                    fb.step(sim, 0.5)
                fb.f_ode(w)
                assert np.array_equal(ts.x[k], w.x), k
                assert np.array_equal(ts.y[k], w.y[rows]), k
        assert (w.status == 0).all()
        w.close()


@pytest.mark.parametrize("where", ["air", "ground"])
def test_x2_launch_partition_is_invisible(fb, gains, where):
    """A run cut into launches of 1, 7 or 50 steps is the same run, bit for bit — state, control-law record, discrete states — as long as
    every aircraft stays with one pass (the airborne and the ground-capable pass are different code and agree to rounding only). What it
    pins: the control-law schedule across launches (ctl_phase), the actuators' closed form (the first RK4 stage IS the state: act_stage_pos,
    csrc/c172_kernels.hpp), and on the ground the derivative carried from launch to launch (k1 / k1_valid) against the one a launch
    evaluates for itself — round 6 found the carried one had never been used in the two-pass flow, and was not the same numbers."""
    K = fb.K
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    n = 512
    rng = np.random.default_rng(12)
    out = []
    for spl in (1, 7, 50):
        if where == "air":
            w = fb.Cessna172Xv2World(n, gains=gains)
            w.set_params(wind_ned=(2.0, -1.0, 0.0))
            sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=spl)
            r = np.random.default_rng(5)
            fb.init(sim, fb.TrimParameters(EAS=r.uniform(38, 55, n), h_e=r.uniform(300, 2500, n), ψ_nb=r.uniform(-3, 3, n)))
            assert w.trim_success.all()
            cu = w.cu
            cu[K["FB_CU_LON_MODE_REQ"]] = r.integers(0, 9, n); cu[K["FB_CU_LAT_MODE_REQ"]] = r.integers(0, 5, n)
            cu[K["FB_CU_EAS_REF"]] += 2.0; cu[K["FB_CU_CLM_REF"]] += 1.0; cu[K["FB_CU_PHI_REF"]] += 0.2
            w.cu = cu
        else:
            import ground_launch_anatomy as gla
            w = gla.parked(n)                      # parked, brakes set, engine off ...
            x = w.x; s = w.s
            roll = np.arange(n) % 2 == 0           # ... and every other one rolling at 3-12 m/s when the brakes bite
            x[K["FB_X2_DYN"] + 3] = np.where(roll, np.random.default_rng(6).uniform(3, 12, n), 0.0)
            w.set_state(x, s)
            fb.f_init(w, None)
            sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=spl)
        fb.step(sim, 1.4); w.sync()               # (140 steps: 1, 7 and 50 all end launches at other instants on the way)
        out.append(dict(x=w.x, cs=w.cs, s=w.s, status=w.status))
        w.close()
    a = out[0]
    assert (a["status"] == 0).all()
    if where == "ground":
        v = a["x"][K["FB_X2_DYN"] + 3]
        assert v[::2].min() > 0.5 and np.abs(v[1::2]).max() < 0.1, "the rolling half is still rolling (friction regulators active), the parked half stands (settling on its struts)"
    for b in out[1:]:
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["cs"], b["cs"]) and np.array_equal(a["s"], b["s"]) and np.array_equal(a["status"], b["status"])
