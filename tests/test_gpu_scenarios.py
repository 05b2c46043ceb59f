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
    batch-wide run's touchdown points (the rows are read by the one-wave kernels, the batch-wide block by the wave-pair kernel: same
    arithmetic, to rounding), and over a 0 ... 9 m/s distribution every aircraft still lands on the runway — with the crab angle at
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
