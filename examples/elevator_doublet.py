#!/usr/bin/env python3
"""The reference's elevator doublet through a user callback (lib/FlightApps/demos/c172_demos.jl:286-316) for a BATCH of fly-by-wire
Cessna172Xv2: trimmed level flight, `elevator_offset` = +0.1 for 5 <= t < 7, -0.1 for 7 <= t < 9, 0 otherwise, the longitudinal channel in
`direct` mode (the demo flies Cessna172Xv1, whose avionics are the direct channel alone; here the aircraft differ in airspeed and altitude and
each gets its own doublet amplitude).

Two forms of the same logic: `mode="callback"` — the closure after every step on the host — and `mode="device"` — the same as a scenario table
(flightbatch.scenario) evaluated on the device between the stepping launches. `python examples/elevator_doublet.py [n] [device]`."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402


def scenario_table():
    """if 5 <= t < 7: offset = +a elseif 7 <= t < 9: offset = -a else offset = 0 — as phases: before, up, down, after (parameter row 0: the amplitude a)."""
    from flightbatch import scenario as sc
    scn = sc.Scenario(n_par=1, n_rec=0)
    BEFORE, UP, DOWN, AFTER = (scn.phase(p) for p in ("before", "up", "down", "after"))
    scn.always(BEFORE, [sc.cu("ELEVATOR_OFFSET", 0.0)])
    scn.when(BEFORE, sc.src.T >= 5.0, [sc.cu("ELEVATOR_OFFSET", sc.par(0))], then=UP)
    scn.always(UP, [sc.cu("ELEVATOR_OFFSET", sc.par(0))])
    scn.when(UP, sc.src.T >= 7.0, [sc.cu("ELEVATOR_OFFSET", -sc.par(0))], then=DOWN)
    scn.always(DOWN, [sc.cu("ELEVATOR_OFFSET", -sc.par(0))])
    scn.when(DOWN, sc.src.T >= 9.0, [sc.cu("ELEVATOR_OFFSET", 0.0)], then=AFTER)
    scn.always(AFTER, [sc.cu("ELEVATOR_OFFSET", 0.0)])
    return scn


def run(n=64, t_end=20.0, dt=0.02, seed=0, mode="callback", verbose=False):
    K = fb.K
    rng = np.random.default_rng(seed)
    w = fb.Cessna172Xv2World(n)
    sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters(EAS=rng.uniform(40.0, 52.0, n), h_e=rng.uniform(500.0, 2500.0, n)))
    assert w.trim_success.all()
    amp = rng.uniform(0.05, 0.1, n)
    theta0 = None
    q_peak = np.zeros(n)
    if mode == "device":
        w.set_scenario(scenario_table(), params=amp[None], every=1)
        fb.step(sim, t_end); w.sync()
        phase = w.scenario_state()["phase"].astype(int)
    else:
        def callback(mdl):
            t = mdl.t
            cu = mdl.cu
            cu[K["FB_CU_ELEVATOR_OFFSET"]] = amp if 5 <= t < 7 else (-amp if 7 <= t < 9 else 0.0)
            mdl.cu = cu
        sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, user_callback=callback)
        fb.step(sim, t_end); w.sync()
        phase = np.full(n, 3)
    out = dict(x=w.x, cs=w.cs, cu=w.cu, status=w.status, phase=phase, amp=amp)
    if verbose:
        print(f"n = {n}, mode {mode}: terminated {int((w.status != 0).sum())}, final elevator offset {np.abs(out['cu'][K['FB_CU_ELEVATOR_OFFSET']]).max():.3f}, "
              f"phases {np.bincount(phase, minlength=4)}")
    w.close()
    return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 64, mode="device" if "device" in sys.argv[2:] else "callback", verbose=True)
