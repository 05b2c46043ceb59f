#!/usr/bin/env python3
"""The reference's scripted crosswind landing (lib/FlightApps/demos/c172_demos.jl:406-497) for a BATCH of Cessna172Xv2:
every aircraft flies the final leg to runway 15 of LOWS under segment guidance in a 6 m/s crosswind, flares 6 m above the
runway (climb-rate hold at -0.3 m/s, crab turned into a sideslip), closes the throttle at touchdown and brakes to a stop.
The phase logic is the demo's user callback, vectorised over the batch; the aircraft differ in their starting distance and
approach speed — and, with `crosswind=` (an array, or `python examples/crosswind_landing.py n disperse`), in the WIND each of them
lands in: the demo sets `world.atmosphere.wind.u.E = 6` on its one simulation (c172_demos.jl:424), a batch is N simulations, each with
its own world, so a touchdown-dispersion study over a wind distribution is one launch sequence (BatchedWorld.set_env, fb_set_env).
`python examples/crosswind_landing.py [n]` prints a summary; tests/test_gpu_scenarios.py asserts on it.

Two forms of the same phase logic: `mode="callback"` — a host `user_callback!` after every step (numpy over the batch; every step crosses PCIe with
the output record: fine for n = 32) — and `mode="device"` — the logic as a table (flightbatch.scenario) interpreted on the device between the
stepping launches, with nothing crossing to the host during the run: what a touchdown-dispersion study at N = 10^6 uses
(`python examples/crosswind_landing.py 1048576 disperse device`). Both give the same phases and touchdown points, bit for bit."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402
from flightbatch.guidance import Segment  # noqa: E402

LOC = (np.deg2rad(47.80433), np.deg2rad(12.997)); H_ORTH = 427.2; PSI = np.deg2rad(157.0)   # c172_demos.jl:17-19


def scenario_table(hold_decrab):
    """The demo's callback (c172_demos.jl:423-486) as a table. Parameter rows (per aircraft): 0-2 / 3-5 the final leg's end points, 6 approach
    EAS, 7 the runway's ellipsoidal altitude, 8 the starting distance. Record rows: 0 touchdown time, 1 distance past the threshold, 2 cross-track."""
    from flightbatch import scenario as sc
    scn = sc.Scenario(n_par=9, n_rec=3)
    INIT, FINAL, FLARE, GROUND = (scn.phase(p) for p in ("init", "final", "flare", "ground"))
    scn.when(INIT, sc.ALWAYS, [sc.cu("GDC_MODE_REQ", float(fb.ModeGuidance.segment))] + sc.target(0, 3) +
             [sc.cu("SEG_HOR_REQ", 1), sc.cu("SEG_VRT_REQ", 1), sc.cu("EAS_REF", sc.par(6)), sc.u("FLAPS", 1.0)], then=FINAL)
    flare = [sc.cu("SEG_VRT_REQ", 0), sc.cu("LON_MODE_REQ", float(fb.ModeControlLon.EAS_clm)), sc.cu("CLM_REF", -0.3),
             sc.cu("LAT_MODE_REQ", float(fb.ModeControlLat.φ_β)),
             sc.cu("BETA_REF", sc.wrap_to_pi(sc.src.PSI - sc.cs_("SEG_CHI_REF") + sc.cs_("SEG_DCHI"))),   # wrap_to_π(ψ - χ_12)
             sc.cu("PHI_REF", 0.0)]
    if hold_decrab:
        flare.append(sc.cu("SEG_HOR_REQ", 0))
    scn.when(FINAL, sc.src.H_E - sc.par(7) < 6.0, flare, then=FLARE)   # vehicle.y.kinematics.h_e - final_leg.p2.h < 6
    scn.when(FLARE, sc.src.ON_GND > 0.5, [sc.cu("THROTTLE_AXIS", 0.0), sc.cu("RUDDER_AXIS", -0.04), sc.u("FLAPS", 0.0),
                                           sc.rec(0, sc.src.T), sc.rec(1, sc.cs_("SEG_S_1B") - sc.par(8)), sc.rec(2, sc.cs_("SEG_E_SB"))], then=GROUND)
    scn.always(GROUND, [sc.cu("THROTTLE_AXIS", 0.0), sc.u("BRAKE_LEFT", 1.0), sc.u("BRAKE_RIGHT", 1.0)])
    return scn


def run(n=64, t_end=150.0, dt=0.02, seed=0, verbose=False, hold_decrab=False, kinematics="WA", crosswind=None, mode="callback", every=1, probe_t=None):
    """hold_decrab = False: the demo's callback to the letter — it leaves `seg.u.hor_gdc_req` set in the flare, so the guidance law puts the
    lateral channel back on track hold (χ_β) at its next update and the de-crab (φ_β with β_ref = ψ − χ_12) lasts one control period.
    hold_decrab = True: the request is dropped at the flare, so the bank + sideslip mode stays in force until touchdown (a variant, not
    the demo).
    crosswind: None — the demo's 6 m/s from the east for every aircraft (batch-wide fb_params) — or [n] east-wind components, one per
    aircraft (per-aircraft environment rows).
    probe_t: also return the state and the control-law record at that time (out["probe"]), and the initial condition (out["ic"])."""
    K = fb.K
    rng = np.random.default_rng(seed)
    w = fb.Cessna172Xv2World(n, kinematics=kinematics)               # Cessna172Xv2(kinematics): WA (the demo's), ECEF or NED
    w.set_params(h_terrain=H_ORTH, wind_ned=(0.0, 6.0, 0.0))          # HorizontalTerrain(h_LOWS15); atmosphere.wind.u.E = 6
    if crosswind is not None:   # every simulation's own atmosphere.wind.u (FP/atmosphere.jl:156-165); terrain and sea level from the block above
        cw = np.asarray(crosswind, dtype=np.float64).reshape(n)
        w.set_env(wind_ned=np.stack([np.zeros(n), cw, np.zeros(n)]))
    # ellipsoidal altitude of the runway: orthometric + geoid height at the threshold (asked from the device model itself)
    probe = fb.TrimParameters(n_e=np.array([np.cos(LOC[0]) * np.cos(LOC[1]), np.cos(LOC[0]) * np.sin(LOC[1]), np.sin(LOC[0])]), h_e=1000.0)
    sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=1)
    fb.init(sim, probe)
    fb.f_ode(w)
    y = w.y
    geoid = float((y[K["FB_Y_KIN"] + 20] - y[K["FB_Y_KIN"] + 21])[0])
    p_rwy = np.array([LOC[0], LOC[1], H_ORTH + geoid])
    # final leg: from s metres out on the extended centreline, 3 degrees above the threshold, down to the threshold
    s0 = rng.uniform(2500.0, 3500.0, n)
    p2 = np.repeat(p_rwy[:, None], n, axis=1)
    far = Segment.from_origin(p2, s0, PSI + np.pi, γ=np.deg2rad(3)).p2        # [3, n]: vectorised over the aircraft
    EAS = rng.uniform(29.0, 32.0, n)
    n_e = np.array([np.cos(far[0]) * np.cos(far[1]), np.cos(far[0]) * np.sin(far[1]), np.sin(far[0])])
    fb.init(sim, fb.TrimParameters(n_e=n_e, h_e=far[2], EAS=EAS, ψ_nb=PSI, γ_wb_n=-np.deg2rad(3), flaps=1.0, fuel_load=0.5))
    assert w.trim_success.all(), "approach trim failed"
    ic = dict(x=w.x, s=w.s, u=w.u, ui=w.ui, cu=w.cu, cs=w.cs, far=far, p2=p2, EAS=EAS, s0=s0, p_rwy=p_rwy)
    probe = None
    phase = np.zeros(n, dtype=int)          # 0 init, 1 final, 2 flare, 3 ground
    touchdown = np.full((3, n), np.nan)     # time, along-track distance past the threshold, cross-track error
    if mode == "device":
        # the same logic as a table, interpreted on the device behind every `every` steps: no host traffic inside the run
        par_rows = np.concatenate([far, p2, EAS[None], np.full((1, n), p_rwy[2]), s0[None]])
        w.set_scenario(scenario_table(hold_decrab), params=par_rows, every=every, rec_init=np.nan)
        sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=50)
        if probe_t is not None:
            fb.step(sim, probe_t); w.sync(); probe = dict(x=w.x, cs=w.cs, t=sim.t)
        fb.step(sim, t_end - sim.t); w.sync()
        st_ = w.scenario_state()
        phase, touchdown = st_["phase"].astype(int), st_["rec"]

    def callback(mdl):
        """the demo's closure, vectorised: every aircraft takes the branch of the phase it ENTERED the call in (if / elseif: one branch per call)"""
        fb.f_ode(mdl)
        yy = mdl.y
        h_e, psi = yy[K["FB_Y_KIN"] + 20], yy[0]
        on_gnd = (yy[K["FB_Y_LDG"] + 1] + yy[K["FB_Y_LDG"] + 12] + yy[K["FB_Y_LDG"] + 23]) > 0
        cu, u, cs = mdl.cu, mdl.u, mdl.cs
        ph0 = np.where(mdl.status == 0, phase, -1)     # (an aircraft whose simulation has ended gets no callback)
        init = ph0 == 0
        if init.any():
            cu[K["FB_CU_GDC_MODE_REQ"], init] = fb.ModeGuidance.segment
            cu[K["FB_CU_SEG_P1"]:K["FB_CU_SEG_P1"] + 3, init] = far[:, init]; cu[K["FB_CU_SEG_P2"]:K["FB_CU_SEG_P2"] + 3, init] = p2[:, init]
            cu[K["FB_CU_SEG_HOR_REQ"], init] = 1; cu[K["FB_CU_SEG_VRT_REQ"], init] = 1
            cu[K["FB_CU_EAS_REF"], init] = EAS[init]
            u[K["FB_U_FLAPS"], init] = 1.0
            phase[init] = 1
        flare = (ph0 == 1) & (h_e - p_rwy[2] < 6)
        if flare.any():
            cu[K["FB_CU_SEG_VRT_REQ"], flare] = 0
            cu[K["FB_CU_LON_MODE_REQ"], flare] = fb.ModeControlLon.EAS_clm; cu[K["FB_CU_CLM_REF"], flare] = -0.3
            cu[K["FB_CU_LAT_MODE_REQ"], flare] = fb.ModeControlLat.φ_β
            d = (psi - cs[K["FB_CS_SEG_CHI_REF"]]) + cs[K["FB_CS_SEG_DCHI"]]     # ψ - χ_12, χ_12 = χ_ref - Δχ (summed in the table's order)
            cu[K["FB_CU_BETA_REF"], flare] = (d + 2 * np.pi * np.floor((np.pi - d) / (2 * np.pi)))[flare]     # wrap_to_π
            cu[K["FB_CU_PHI_REF"], flare] = 0.0
            if hold_decrab:   # (the demo leaves hor_gdc_req set: c172_demos.jl:447-461)
                cu[K["FB_CU_SEG_HOR_REQ"], flare] = 0
            phase[flare] = 2
        touch = (ph0 == 2) & on_gnd
        if touch.any():
            cu[K["FB_CU_THROTTLE_AXIS"], touch] = 0.0; cu[K["FB_CU_RUDDER_AXIS"], touch] = -0.04
            u[K["FB_U_FLAPS"], touch] = 0.0
            touchdown[0, touch] = mdl.t
            touchdown[1, touch] = cs[K["FB_CS_SEG_S_1B"], touch] - s0[touch]
            touchdown[2, touch] = cs[K["FB_CS_SEG_E_SB"], touch]
            phase[touch] = 3
        gnd = ph0 == 3
        if gnd.any():
            cu[K["FB_CU_THROTTLE_AXIS"], gnd] = 0.0
            u[K["FB_U_BRAKE_LEFT"], gnd] = 1.0; u[K["FB_U_BRAKE_RIGHT"], gnd] = 1.0
        mdl.cu = cu; mdl.u = u

    if mode == "callback":
        sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, user_callback=callback)
        sim._nstep = 0
        if probe_t is not None:
            fb.step(sim, probe_t); w.sync(); probe = dict(x=w.x, cs=w.cs, t=sim.t)
        fb.step(sim, t_end - sim.t); w.sync()
    fb.f_ode(w)
    y = w.y
    out = dict(phase=phase.copy(), status=w.status, v_gnd=y[K["FB_Y_KIN"] + 37], touchdown=touchdown, h_agl=y[K["FB_Y_KIN"] + 21] - H_ORTH,
               e_sb=w.cs[K["FB_CS_SEG_E_SB"]], x=w.x, cs=w.cs, ic=ic, probe=probe)
    if verbose:
        print(f"n = {n}: phases {np.bincount(phase, minlength=4)}, terminated {int((w.status != 0).sum())}, touchdown at "
              f"{np.nanmin(touchdown[0]):.1f}-{np.nanmax(touchdown[0]):.1f} s, {np.nanmin(touchdown[1]):.0f}..{np.nanmax(touchdown[1]):.0f} m past the threshold, "
              f"cross-track {np.nanmax(np.abs(touchdown[2])):.2f} m max, final ground speed {out['v_gnd'].max():.2f} m/s")
    w.close()
    return out


if __name__ == "__main__":
    n_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    mode_ = "device" if "device" in sys.argv[2:] else "callback"
    every_ = next((int(a_.split("=")[1]) for a_ in sys.argv[2:] if a_.startswith("every=")), 1)   # scenario period in steps (device mode)
    if len(sys.argv) > 2 and sys.argv[2] == "disperse":   # touchdown dispersion over a crosswind distribution: 0 ... 9 m/s from the east
        import time
        cw_ = np.random.default_rng(1).uniform(0.0, 9.0, n_)
        t0_ = time.perf_counter()
        o_ = run(n_, verbose=True, crosswind=cw_, hold_decrab=True, mode=mode_, every=every_)
        el_ = time.perf_counter() - t0_
        print(f"{mode_} (scenario evaluated every {every_} step(s)): {n_} aircraft x 7500 steps in {el_:.1f} s (trim and set-up included): {n_ * 7500 / el_:.3e} aircraft-steps/s")
        td_ = o_["touchdown"]
        for lo_ in range(0, 9, 3):
            m_ = (cw_ >= lo_) & (cw_ < lo_ + 3) & np.isfinite(td_[0])
            if m_.any():
                print(f"crosswind {lo_}-{lo_ + 3} m/s: {int(m_.sum())} aircraft, touchdown {td_[1][m_].mean():.0f} ± {td_[1][m_].std():.0f} m past the threshold, "
                      f"cross-track {td_[2][m_].mean():+.2f} ± {td_[2][m_].std():.2f} m")
    else:
        run(n_, verbose=True, hold_decrab="hold" in sys.argv[2:], mode=mode_, every=every_)
