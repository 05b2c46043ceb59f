"""End-to-end scripted scenario on the GPU: the reference's crosswind-landing demo (lib/FlightApps/demos/c172_demos.jl:406-497)
for a batch — trim on final, segment guidance, flare, touchdown, braking — through everything the path contains: Cessna172Xv2
actuators, control laws, guidance, landing gear ground contact, the air / ground kernel hand-over, wind, user callback."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hold_decrab", [False, True])
def test_crosswind_landing_batch(fb, hold_decrab):
    """hold_decrab = False is the demo's callback to the letter (it leaves the horizontal guidance request set in the flare, so the
    de-crab lasts one control period); True keeps the bank + sideslip mode to touchdown — both must land every aircraft."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    out = demo.run(n=32, t_end=150.0, seed=3, hold_decrab=hold_decrab)
    td = out["touchdown"]
    print("touchdown: %.1f-%.1f s, %.0f..%.0f m past the threshold, |cross-track| <= %.2f m; final ground speed <= %.3f m/s"
          % (td[0].min(), td[0].max(), td[1].min(), td[1].max(), np.abs(td[2]).max(), out["v_gnd"].max()))
    assert (out["status"] == 0).all(), "an aircraft crashed or left the envelope"
    assert (out["phase"] == 3).all(), "every aircraft must reach the ground roll"
    assert np.isfinite(td).all() and (td[1] > -50).all() and (td[1] < 300).all()       # on the runway, near the aiming point
    assert np.abs(td[2]).max() < 3.0                                                    # on the centreline in a 6 m/s crosswind
    assert out["v_gnd"].max() < 0.5 and np.abs(out["h_agl"] - 1.85).max() < 0.2         # stopped, sitting on its wheels


def test_crosswind_landing_wind_dispersion(fb):
    """Every aircraft lands in its OWN crosswind (per-aircraft environment rows, fb_set_env): rows that repeat the demo's 6 m/s give the
    batch-wide run's touchdown points (the PERENV instances of the stepping kernels read the rows — k_step_duo<WA, X, true> in the air, the
    ground-capable k_step_air<.., true, true> near the runway — where the batch-wide run's instances read the one block: same arithmetic, to
    rounding), and over a 0 ... 9 m/s distribution every aircraft still lands on the runway — with the crab angle at
    the flare growing with the crosswind it flies in."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    n = 32
    ref = demo.run(n=n, t_end=150.0, seed=3)
    same = demo.run(n=n, t_end=150.0, seed=3, crosswind=np.full(n, 6.0))
    d = np.abs(same["touchdown"] - ref["touchdown"])
    print("rows = the batch-wide block: touchdown differs by %.2e s, %.2e m along, %.2e m across" % (d[0].max(), d[1].max(), d[2].max()))
    assert (same["status"] == 0).all() and (same["phase"] == 3).all()
    assert d[0].max() < 0.05 and d[1].max() < 1.0 and d[2].max() < 0.1
    cw = np.linspace(0.0, 9.0, n)
    out = demo.run(n=n, t_end=150.0, seed=3, crosswind=cw)
    td = out["touchdown"]
    assert (out["status"] == 0).all() and (out["phase"] == 3).all()
    assert np.isfinite(td).all() and (td[1] > -50).all() and (td[1] < 300).all() and np.abs(td[2]).max() < 4.0
    assert out["v_gnd"].max() < 0.5
    # the wind each aircraft flew in is its own: touchdown times differ with it (ground speed on final = airspeed along track),
    # and the aircraft in no crosswind touched down closest to the centreline
    assert np.abs(td[2][:4]).mean() < np.abs(td[2][-4:]).mean() + 0.5
    assert not np.allclose(td[:, 0], td[:, -1])


@pytest.mark.parametrize("kin", ["ECEF", "NED"])
def test_crosswind_landing_in_the_other_mechanisations(fb, kin):
    """The same scripted landing with Cessna172Xv2(ECEF()) / Cessna172Xv2(NED()): guidance on the tapped latitude / longitude, the flare,
    touchdown, nose-wheel steering and braking through the ground-capable instance of that mechanisation. The touchdown points must be
    those of the WA run to within a metre (the mechanisations integrate the same motion)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    ref = demo.run(n=32, t_end=150.0, seed=3)
    out = demo.run(n=32, t_end=150.0, seed=3, kinematics=kin)
    td = out["touchdown"]
    assert (out["status"] == 0).all() and (out["phase"] == 3).all()
    assert np.isfinite(td).all() and np.abs(td[2]).max() < 3.0 and out["v_gnd"].max() < 0.5 and np.abs(out["h_agl"] - 1.85).max() < 0.2
    d = np.abs(td - ref["touchdown"])
    print(kin, "touchdown against the WA run: time %.3f s, along track %.3f m, cross track %.3f m" % (d[0].max(), d[1].max(), d[2].max()))
    assert d[0].max() < 0.1 and d[1].max() < 2.0 and d[2].max() < 0.2


def test_crosswind_landing_is_reproducible(fb):
    """The same scenario twice in one process, bit for bit — through the airborne pass, the hand-over, the ground-capable pass and the
    in-kernel control laws. (A kernel miscompiled at full register pressure — spill code before the exec restore of a join block,
    tools/check_isa_spills.py — reloads whatever the scratch memory held: the two runs then differ, as they did during round 2.)"""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    a = demo.run(n=64, t_end=120.0, seed=0)
    b = demo.run(n=64, t_end=120.0, seed=0)
    assert (a["phase"] >= 2).any(), "the horizon must include the flare and the ground roll"
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["cs"], b["cs"]) and np.array_equal(a["status"], b["status"])


def test_traffic_pattern_batch(fb):
    """c172_demos.jl:502-645: cold start on the runway, engine start, takeoff, four guided legs, final, flare, landing, full stop."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import traffic_pattern as demo
    out = demo.run(n=16, t_end=420.0, seed=1)
    e = out["entered"]
    assert (out["status"] == 0).all(), "an aircraft crashed or left the envelope"
    assert (out["phase"] == demo.GROUND).all(), "every aircraft must complete the pattern"
    assert np.isfinite(e[1:]).all() and (np.diff(e[1:], axis=0) > 0).all()            # every phase visited, in order
    assert (e[demo.TAKEOFF] - e[demo.STARTUP] < 3).all()                              # the starter brings the engine up within seconds
    assert (e[demo.DEPARTURE] - e[demo.TAKEOFF] < 30).all()                           # airborne after a ground roll at full throttle
    td = out["touchdown"]
    assert (td[0] > -100).all() and (td[0] < 400).all() and np.abs(td[1]).max() < 5   # touchdown near the runway point, on the centreline
    assert out["v_gnd"].max() < 0.5                                                   # braked to a stop


# ---- the scenarios as TABLES, interpreted on the device (flightbatch/scenario.py, csrc/scenario_kernels.hpp) ----
def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_crosswind_landing_device_table_equals_host_callback(fb):
    """The demo's closure as a host callback after every step (PCIe both ways, every step) and as a table interpreted by k_scenario between the
    stepping launches (nothing crosses to the host during the run): same phases, same touchdown records, same final state — bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    for hold in (False, True):
        a = demo.run(n=32, t_end=150.0, seed=3, hold_decrab=hold)
        b = demo.run(n=32, t_end=150.0, seed=3, hold_decrab=hold, mode="device")
        assert (a["phase"] == 3).all() and _same(a["phase"], b["phase"])
        assert _same(a["touchdown"], b["touchdown"]), np.nanmax(np.abs(a["touchdown"] - b["touchdown"]))
        assert _same(a["x"], b["x"]) and _same(a["cs"], b["cs"]) and _same(a["status"], b["status"])


def test_traffic_pattern_device_table_equals_host_callback(fb):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import traffic_pattern as demo
    a = demo.run(n=32, t_end=420.0, seed=1)
    b = demo.run(n=32, t_end=420.0, seed=1, mode="device")
    assert (a["phase"] == demo.GROUND).all() and _same(a["phase"], b["phase"])
    assert _same(a["entered"][1:], b["entered"][1:]) and _same(a["touchdown"], b["touchdown"])
    assert _same(a["status"], b["status"]) and _same(a["v_gnd"], b["v_gnd"])


def test_scenario_table_is_validated_on_the_host(fb):
    """A table is data from outside: every index the kernel would follow is checked when it is loaded (an out-of-range row on the device is a
    fault), other models refuse it, and the period must be positive."""
    from flightbatch import scenario as sc
    K = fb.K
    w = fb.Cessna172Xv2World(64)
    scn = sc.Scenario(n_par=2, n_rec=1)
    A, B = scn.phase("a"), scn.phase("b")
    scn.when(A, sc.src.T >= 1.0, [sc.cu("EAS_REF", sc.par(1)), sc.rec(0, sc.src.T)], then=B)
    w.set_scenario(scn, params=np.ones((2, 64)))
    good = scn.pack()

    def load(blob):
        import ctypes as C
        blob = np.ascontiguousarray(blob, dtype=np.float64)
        return fb.lib.fb_set_table(w._h, K["FB_TABLE_SCENARIO"], blob.ctypes.data_as(C.c_void_p), (C.c_int64 * 1)(blob.size), 1)

    assert load(good) == 0
    hdr, nph = K["FB_SCN_HDR"], 2
    rule0 = hdr + K["FB_SCN_PHASE_REC"] * nph
    act0 = rule0 + K["FB_SCN_RULE_REC"] * 1
    for pos, val, what in ((0, 7.0, b"layout version"), (rule0 + 7, 5.0, b"next phase"), (rule0 + 1, 3.0, None), (act0 + 1, 1000.0, b"destination row"),
                           (act0 + 6, 9.0, b"source row"), (act0 + 4, 4.0, b"term count"), (rule0 + 5, 99.0, b"first action"), (1, 3.0, b"header needs")):
        bad = good.copy(); bad[pos] = val
        rc = load(bad)
        if what is None:
            continue   # (a source row of a kind without rows: accepted)
        assert rc != 0 and what in fb.lib.fb_last_error(), (pos, val, fb.lib.fb_last_error())
    assert load(good[:-1]) != 0
    assert load(good) == 0, "a rejected table must leave the handle usable"
    assert fb.lib.fb_scenario_configure(w._h, -1) != 0
    sv = fb.BatchedWorld(64)
    import ctypes as C
    assert fb.lib.fb_set_table(sv._h, K["FB_TABLE_SCENARIO"], good.ctypes.data_as(C.c_void_p), (C.c_int64 * 1)(good.size), 1) != 0 and b"Cessna172Xv2" in fb.lib.fb_last_error()
    sv.close(); w.close()


def test_scenario_state_survives_a_checkpoint(fb):
    """checkpoint -> np.savez -> restore in a fresh world carries the table, its period, the parameters and every aircraft's phase / entry step /
    records: the resumed run ends bit for bit where the uninterrupted one does."""
    import io
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    K = fb.K
    n = 32
    ref = demo.run(n=n, t_end=120.0, seed=5, mode="device")
    # the same run, interrupted at t = 100 s (final / flare for most aircraft)
    from flightbatch.guidance import Segment  # noqa: F401
    w = fb.Cessna172Xv2World(n)
    w.set_params(h_terrain=demo.H_ORTH, wind_ned=(0.0, 6.0, 0.0))
    ic = ref["ic"]
    w.set_state(ic["x"], ic["s"]); w.u = ic["u"]; w.ui = ic["ui"]; w.cu = ic["cu"]; w.cs = ic["cs"]
    par_rows = np.concatenate([ic["far"], ic["p2"], ic["EAS"][None], np.full((1, n), ic["p_rwy"][2]), ic["s0"][None]])
    w.set_scenario(demo.scenario_table(False), params=par_rows, every=1, rec_init=np.nan)
    sim = fb.Simulation(w, dt=0.02, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.step(sim, 100.0); w.sync()
    buf = io.BytesIO(); np.savez(buf, **fb.checkpoint(sim)); buf.seek(0)
    w.close()
    ck = dict(np.load(buf))
    w2 = fb.Cessna172Xv2World(n)
    w2.set_params(h_terrain=demo.H_ORTH, wind_ned=(0.0, 6.0, 0.0))
    sim2 = fb.Simulation(w2, dt=0.02, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.restore(sim2, ck)
    fb.step(sim2, 20.0); w2.sync()
    st = w2.scenario_state()
    assert _same(st["phase"], ref["phase"]) and _same(st["rec"], ref["touchdown"]) and _same(w2.x, ref["x"]) and _same(w2.cs, ref["cs"])
    w2.close()


def test_crosswind_landing_scenario_against_the_oracles_phase_machine(fb, oracle):
    """256 aircraft fly the scripted crosswind landing on the device (table) and on the CPU: the oracle's Cessna172Xv2 stepped one step at a time,
    the SAME table interpreted by flightbatch.scenario.evaluate_on_host on the oracle's own outputs (ψ, h_e, weight on wheels from its f_ode!).
    Airborne (t = 40 s, everybody on final under segment guidance): state to 1e-6. Through flare, touchdown and ground roll: the same phases, the
    touchdown within two steps and a metre (ground contact is ill-conditioned: docs/design/ground.md)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import crosswind_landing as demo
    from flightbatch import scenario as sc
    from oracle_binding import OracleX
    from test_gpu_c172x import ref_to_dev_rows, x_scale
    K = fb.K
    n, dt, t_end, t_probe = 256, 0.02, 150.0, 40.0
    gpu = demo.run(n=n, t_end=t_end, seed=11, mode="device", probe_t=t_probe)
    ic = gpu["ic"]
    perm = ref_to_dev_rows(K)
    X = OracleX(oracle, fb.ctl_gains.ctl_gains_blob())
    env = oracle.default_env(); env[3] = 6.0; env[5] = demo.H_ORTH     # T_sl p_sl wind N E D h_trn surface
    o = dict(x=np.zeros((34, n)), u=ic["u"].copy(), ui=ic["ui"].copy(), s=ic["s"].copy(), cu=ic["cu"].copy(), cs=ic["cs"].copy(),
             status=np.zeros(n, np.int32), nstep=0)
    o["x"][perm] = ic["x"]
    blob = demo.scenario_table(False).pack()
    st = dict(phase=np.zeros(n, np.int64), since=np.zeros(n, np.int64), step=0, rec=np.full((3, n), np.nan),
              par=np.concatenate([ic["far"], ic["p2"], ic["EAS"][None], np.full((1, n), ic["p_rwy"][2]), ic["s0"][None]]))
    x_probe = None
    nsteps = int(round(t_end / dt))
    for k in range(1, nsteps + 1):
        X.step(o, env, dt, 1, 1, threads=16)
        _, y, _ = X.f_ode(o, env)
        st.update(step=k, cu=o["cu"], cs=o["cs"], u=o["u"], ui=o["ui"], s=o["s"], active=o["status"] == 0, h_e=y[K["FB_Y_KIN"] + 20], psi=y[0],
                  theta=y[1], phi=y[2], chi=y[K["FB_Y_KIN"] + 38], EAS=y[K["FB_Y_AIR"] + 20], clm=-y[K["FB_Y_KIN"] + 36],
                  on_gnd=((y[K["FB_Y_LDG"] + 1] + y[K["FB_Y_LDG"] + 12] + y[K["FB_Y_LDG"] + 23]) > 0).astype(np.float64))
        sc.evaluate_on_host(blob, st, k * dt, dt)
        if k == int(round(t_probe / dt)):
            x_probe = o["x"][perm].copy()
    assert (gpu["status"] == 0).all() and (o["status"] == 0).all()
    err = np.abs(gpu["probe"]["x"] - x_probe) / x_scale(o["x"])[perm]
    print("airborne, t = %.0f s: max scaled state error %.2e" % (t_probe, err.max()))
    assert err.max() < 1e-6
    assert (gpu["phase"] == 3).all() and np.array_equal(gpu["phase"], st["phase"])
    d = np.abs(gpu["touchdown"] - st["rec"])
    print("touchdown against the oracle's phase machine: time %.3f s, along track %.3f m, cross track %.4f m" % (d[0].max(), d[1].max(), d[2].max()))
    assert d[0].max() <= 2 * dt + 1e-9 and d[1].max() < 1.5 and d[2].max() < 0.05


def test_elevator_doublet_device_table_equals_host_callback_and_the_oracle(fb, oracle):
    """c172_demos.jl:286-316: `elevator_offset` = +a for 5 <= t < 7, -a for 7 <= t < 9, else 0, from a user callback — as a host callback, as a table on the
    device (bitwise the same run), and on the oracle stepped one step at a time with the closure applied to ITS inputs (1e-6 after 20 s: the doublet excites
    the short-period and phugoid modes of 64 aircraft at different speeds and altitudes)."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import elevator_doublet as demo
    from oracle_binding import OracleX
    from test_gpu_c172x import ref_to_dev_rows, x_scale
    K = fb.K
    a = demo.run(n=64, seed=4)
    b = demo.run(n=64, seed=4, mode="device")
    assert (b["phase"] == 3).all() and _same(a["x"], b["x"]) and _same(a["cs"], b["cs"]) and _same(a["cu"], b["cu"]) and (a["status"] == 0).all()
    # the oracle: same trims (the GPU's initial condition is reproduced by running the example's set-up again), the closure on its inputs
    n, dt = 64, 0.02
    rng = np.random.default_rng(4)
    w = fb.Cessna172Xv2World(n)
    sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters(EAS=rng.uniform(40.0, 52.0, n), h_e=rng.uniform(500.0, 2500.0, n)))
    amp = rng.uniform(0.05, 0.1, n)
    assert np.array_equal(amp, a["amp"])
    perm = ref_to_dev_rows(K)
    X = OracleX(oracle, fb.ctl_gains.ctl_gains_blob())
    env = oracle.default_env()
    o = dict(x=np.zeros((34, n)), u=w.u, ui=w.ui, s=w.s, cu=w.cu, cs=w.cs, status=np.zeros(n, np.int32), nstep=0)
    o["x"][perm] = w.x
    w.close()
    for k in range(1, 1001):
        X.step(o, env, dt, 1, 1, threads=16)
        t = k * dt
        o["cu"][K["FB_CU_ELEVATOR_OFFSET"]] = amp if 5 <= t < 7 else (-amp if 7 <= t < 9 else 0.0)
    err = np.abs(b["x"] - o["x"][perm]) / x_scale(o["x"])[perm]
    print("elevator doublet, 64 aircraft x 1000 steps, device table against the oracle with the closure: max scaled error %.2e" % err.max())
    assert err.max() < 1e-6 and (o["status"] == 0).all()
    dq = np.abs(b["x"][K["FB_X2_KIN"] + 8] - a["x"][K["FB_X2_KIN"] + 8]).max()
    assert dq == 0.0
