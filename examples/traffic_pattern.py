#!/usr/bin/env python3
"""The reference's scripted traffic pattern (lib/FlightApps/demos/c172_demos.jl:502-645) for a BATCH of Cessna172Xv2: parked on
runway 15 of LOWS with the engine off, each aircraft starts its engine, takes off at full throttle, flies departure, crosswind,
downwind, base and final under segment guidance (next leg 200 m before the end of the current one), flares, lands and brakes.
The phase logic is the demo's user callback, vectorised over the batch; the aircraft differ in payload and fuel load.
`python examples/traffic_pattern.py [n] [device]` prints a summary; tests/test_gpu_scenarios.py asserts on it.

Two forms of the same logic (see crosswind_landing.py): `mode="callback"`, a host callback after every step, and `mode="device"`, the table of
`scenario_table()` interpreted on the device between the stepping launches — same phases, entry times and touchdown points, bit for bit."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402
from flightbatch.guidance import Segment  # noqa: E402

LOC = (np.deg2rad(47.80433), np.deg2rad(12.997)); H_ORTH = 427.2; PSI = np.deg2rad(157.0)   # c172_demos.jl:17-19
STANDBY, STARTUP, TAKEOFF, DEPARTURE, CROSSWIND, DOWNWIND, BASE, FINAL, FLARE, GROUND = range(10)


LEG_PAR = {DEPARTURE: 0, CROSSWIND: 6, DOWNWIND: 12, BASE: 18, FINAL: 24}   # parameter rows of the legs' end points (p1: +0..2, p2: +3..5)
PAR_H_RWY, N_PAR = 30, 31
REC_TD_S2B, REC_TD_ESB, N_REC = 10, 11, 12                                  # record rows 0-9: the time each phase was entered


def scenario_table():
    """The demo's callback (c172_demos.jl:525-642) as a table (flightbatch.scenario)."""
    from flightbatch import scenario as sc
    scn = sc.Scenario(n_par=N_PAR, n_rec=N_REC)
    for name in "standby startup takeoff departure crosswind downwind base final flare ground".split():
        scn.phase(name)
    go = lambda ph: [sc.rec(ph, sc.src.T)]
    leg = lambda ph: sc.target(LEG_PAR[ph], LEG_PAR[ph] + 3)
    scn.when(STANDBY, sc.src.T >= 5.0, go(STARTUP), then=STARTUP)
    scn.always(STARTUP, [sc.ui("ENG_START", True)])
    scn.when(STARTUP, sc.s_("ENG_STATE").eq(2.0), [sc.ui("ENG_START", False)] + go(TAKEOFF), then=TAKEOFF)
    scn.always(TAKEOFF, [sc.cu("GDC_MODE_REQ", float(fb.ModeGuidance.segment))] + leg(DEPARTURE) +
               [sc.cu("SEG_HOR_REQ", 1), sc.cu("SEG_VRT_REQ", 1), sc.cu("EAS_REF", 35.0), sc.cu("THROTTLE_AXIS", 1.0)])
    scn.when(TAKEOFF, sc.src.ON_GND < 0.5, go(DEPARTURE), then=DEPARTURE)
    for cur, nxt in ((DEPARTURE, CROSSWIND), (CROSSWIND, DOWNWIND), (DOWNWIND, BASE), (BASE, FINAL)):
        if cur == DOWNWIND:
            scn.always(cur, [sc.cu("EAS_REF", 50.0)])
        if cur == BASE:
            scn.always(cur, [sc.cu("EAS_REF", 30.0), sc.u("FLAPS", 1.0)])
        scn.when(cur, sc.cs_("SEG_S_2B") > -200.0, leg(nxt) + go(nxt), then=nxt)   # capture_threshold
    scn.when(FINAL, sc.src.H_E - sc.par(PAR_H_RWY) < 6.0,
             [sc.cu("SEG_VRT_REQ", 0), sc.cu("SEG_HOR_REQ", 0), sc.cu("LON_MODE_REQ", float(fb.ModeControlLon.EAS_clm)), sc.cu("CLM_REF", -0.3),
              sc.cu("LAT_MODE_REQ", float(fb.ModeControlLat.φ_β)),
              sc.cu("BETA_REF", sc.wrap_to_pi(sc.src.PSI - sc.cs_("SEG_CHI_REF") + sc.cs_("SEG_DCHI"))), sc.cu("PHI_REF", 0.0)] + go(FLARE), then=FLARE)
    scn.when(FLARE, sc.src.ON_GND > 0.5, [sc.cu("THROTTLE_AXIS", 0.0), sc.cu("RUDDER_AXIS", -0.04), sc.u("FLAPS", 0.0),
                                           sc.rec(REC_TD_S2B, sc.cs_("SEG_S_2B")), sc.rec(REC_TD_ESB, sc.cs_("SEG_E_SB"))] + go(GROUND), then=GROUND)
    scn.always(GROUND, [sc.cu("THROTTLE_AXIS", 0.0), sc.u("BRAKE_LEFT", 1.0), sc.u("BRAKE_RIGHT", 1.0)])
    return scn


def run(n=32, t_end=700.0, dt=0.02, seed=0, verbose=False, mode="callback", every=1):
    K = fb.K
    rng = np.random.default_rng(seed)
    w = fb.Cessna172Xv2World(n)
    w.set_params(h_terrain=H_ORTH)
    sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=1)
    n_e = np.array([np.cos(LOC[0]) * np.cos(LOC[1]), np.cos(LOC[0]) * np.sin(LOC[1]), np.sin(LOC[0])])
    fb.init(sim, fb.TrimParameters(n_e=n_e, h_e=1000.0))          # only to ask the device model for the geoid height here
    fb.f_ode(w)
    geoid = float((w.y[K["FB_Y_KIN"] + 20] - w.y[K["FB_Y_KIN"] + 21])[0])
    p_rwy = np.array([LOC[0], LOC[1], H_ORTH + geoid])
    # the pattern (c172_demos.jl:508-514)
    final_leg = -Segment.from_origin(p_rwy, 3e3, PSI + np.pi, γ=np.deg2rad(3))
    base_leg = -Segment.from_origin(final_leg.p1, 1e3, PSI - np.pi / 2, γ=0.0)
    downwind_leg = -Segment.from_origin(base_leg.p1, 6e3, PSI, γ=0.0)
    crosswind_leg = -Segment.from_origin(downwind_leg.p1, 1e3, PSI + np.pi / 2, γ=0.0)
    departure_leg = Segment(p_rwy, crosswind_leg.p1)
    legs = {DEPARTURE: departure_leg, CROSSWIND: crosswind_leg, DOWNWIND: downwind_leg, BASE: base_leg, FINAL: final_leg}
    # C172.Init(KinInit(location, h = h_LOWS15 + Δh_to_gnd, q_nb = REuler(ψ, 0, 0), ω = 0, v = 0)): parked, engine off
    x = np.zeros((K["FB_X2_NX"], n))
    x[K["FB_X_FUEL"]] = rng.uniform(0.3, 0.9, n)
    kq = K["FB_X2_KIN"]
    x[kq:kq + 4] = np.array([np.cos(PSI / 2), 0, 0, np.sin(PSI / 2)])[:, None]                    # q_wb = q_nb (ψ_nw = 0)
    a = -(LOC[0] + np.pi / 2)
    qz = np.array([np.cos(LOC[1] / 2), 0, 0, np.sin(LOC[1] / 2)]); qy = np.array([np.cos(a / 2), 0, np.sin(a / 2), 0])
    q_ew = np.array([qz[0] * qy[0] - qz[3] * 0, -qz[3] * qy[2], qz[0] * qy[2], qz[3] * qy[0]])   # Rz(λ) ∘ Ry(-(ϕ + π/2))
    x[kq + 4:kq + 8] = q_ew[:, None]
    x[kq + 8] = p_rwy[2] + 1.81                                                                   # C172.Δh_to_gnd
    s = np.zeros((2, n), dtype=np.int32)
    u = np.zeros((K["FB_NU"], n)); u[K["FB_U_MIXTURE"]] = 0.5
    u[K["FB_U_M_PILOT"]] = rng.uniform(60, 95, n); u[K["FB_U_M_COPILOT"]] = rng.uniform(0, 95, n); u[K["FB_U_M_BAGGAGE"]] = rng.uniform(0, 50, n)
    w.set_state(x, s); w.u = u
    w.ui = np.full(n, K["FB_UI_MIXTURE_AUTO"] | K["FB_UI_STEERING_ENGAGED"], dtype=np.int32)
    fb.f_init(w, None)                                                                            # avionics initialisation on this state
    phase = np.full(n, STANDBY)
    entered = np.full((10, n), np.nan)
    touchdown = np.full((2, n), np.nan)

    def set_leg(cu, mask, leg):
        cu[K["FB_CU_SEG_P1"]:K["FB_CU_SEG_P1"] + 3, mask] = leg.p1[:, None]; cu[K["FB_CU_SEG_P2"]:K["FB_CU_SEG_P2"] + 3, mask] = leg.p2[:, None]

    def go(mask, new, t):
        phase[mask] = new; entered[new, mask] = t

    def callback(mdl):
        """the demo's closure, vectorised: every aircraft takes the branch of the phase it ENTERED the call in (if / elseif: one branch per call)"""
        t = mdl.t
        fb.f_ode(mdl)
        yy = mdl.y
        h_e, psi = yy[K["FB_Y_KIN"] + 20], yy[0]
        on_gnd = (yy[K["FB_Y_LDG"] + 1] + yy[K["FB_Y_LDG"] + 12] + yy[K["FB_Y_LDG"] + 23]) > 0
        cu, uu, cs, ui, ss = mdl.cu, mdl.u, mdl.cs, mdl.ui, mdl.s
        alive = mdl.status == 0            # (an aircraft whose simulation has ended gets no callback)
        ph0 = np.where(alive, phase, -1)
        go(( ph0 == STANDBY) & (t >= 5), STARTUP, t)
        m = ph0 == STARTUP
        ui[m] |= K["FB_UI_ENG_START"]
        running = m & (ss[K["FB_S_ENG_STATE"]] == 2)
        ui[running] &= ~K["FB_UI_ENG_START"]
        go(running, TAKEOFF, t)
        m = ph0 == TAKEOFF
        if m.any():
            cu[K["FB_CU_GDC_MODE_REQ"], m] = fb.ModeGuidance.segment; set_leg(cu, m, departure_leg)
            cu[K["FB_CU_SEG_HOR_REQ"], m] = 1; cu[K["FB_CU_SEG_VRT_REQ"], m] = 1
            cu[K["FB_CU_EAS_REF"], m] = 35.0; cu[K["FB_CU_THROTTLE_AXIS"], m] = 1.0
            go(m & ~on_gnd, DEPARTURE, t)
        for cur, nxt in ((DEPARTURE, CROSSWIND), (CROSSWIND, DOWNWIND), (DOWNWIND, BASE), (BASE, FINAL)):
            m = ph0 == cur
            if cur == DOWNWIND: cu[K["FB_CU_EAS_REF"], m] = 50.0
            if cur == BASE: cu[K["FB_CU_EAS_REF"], m] = 30.0; uu[K["FB_U_FLAPS"], m] = 1.0
            sw = m & (cs[K["FB_CS_SEG_S_2B"]] > -200.0)
            if sw.any():
                set_leg(cu, sw, legs[nxt]); go(sw, nxt, t)
        m = (ph0 == FINAL) & (h_e - p_rwy[2] < 6)
        if m.any():
            cu[K["FB_CU_SEG_VRT_REQ"], m] = 0; cu[K["FB_CU_SEG_HOR_REQ"], m] = 0    # (see crosswind_landing.py)
            cu[K["FB_CU_LON_MODE_REQ"], m] = fb.ModeControlLon.EAS_clm; cu[K["FB_CU_CLM_REF"], m] = -0.3
            cu[K["FB_CU_LAT_MODE_REQ"], m] = fb.ModeControlLat.φ_β
            d = (psi - cs[K["FB_CS_SEG_CHI_REF"]]) + cs[K["FB_CS_SEG_DCHI"]]          # ψ - χ_12, χ_12 = χ_ref - Δχ
            cu[K["FB_CU_BETA_REF"], m] = (d + 2 * np.pi * np.floor((np.pi - d) / (2 * np.pi)))[m]; cu[K["FB_CU_PHI_REF"], m] = 0.0
            go(m, FLARE, t)
        m = (ph0 == FLARE) & on_gnd
        if m.any():
            cu[K["FB_CU_THROTTLE_AXIS"], m] = 0.0; cu[K["FB_CU_RUDDER_AXIS"], m] = -0.04; uu[K["FB_U_FLAPS"], m] = 0.0
            touchdown[0, m] = cs[K["FB_CS_SEG_S_2B"], m]; touchdown[1, m] = cs[K["FB_CS_SEG_E_SB"], m]
            go(m, GROUND, t)
        m = ph0 == GROUND
        if m.any():
            cu[K["FB_CU_THROTTLE_AXIS"], m] = 0.0; uu[K["FB_U_BRAKE_LEFT"], m] = 1.0; uu[K["FB_U_BRAKE_RIGHT"], m] = 1.0
        mdl.cu = cu; mdl.u = uu; mdl.ui = ui

    if mode == "device":
        par_rows = np.zeros((N_PAR, n))
        for ph, leg_ in legs.items():
            par_rows[LEG_PAR[ph]:LEG_PAR[ph] + 3] = leg_.p1[:, None]; par_rows[LEG_PAR[ph] + 3:LEG_PAR[ph] + 6] = leg_.p2[:, None]
        par_rows[PAR_H_RWY] = p_rwy[2]
        w.set_scenario(scenario_table(), params=par_rows, every=every, rec_init=np.nan)
        sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, steps_per_launch=50)
        fb.step(sim, t_end); w.sync()
        st_ = w.scenario_state()
        phase = st_["phase"].astype(int); entered = st_["rec"][:10]; touchdown = st_["rec"][REC_TD_S2B:REC_TD_ESB + 1]
    if mode == "callback":
        sim = fb.Simulation(w, dt=dt, Δt=dt, save_on=False, user_callback=callback)
        fb.step(sim, t_end); w.sync()
    fb.f_ode(w)
    y = w.y
    out = dict(phase=phase.copy(), entered=entered, status=w.status, v_gnd=y[K["FB_Y_KIN"] + 37], touchdown=touchdown)
    if verbose:
        names = "standby startup takeoff departure crosswind downwind base final flare ground".split()
        print(f"n = {n}: final phases {dict(zip(*np.unique([names[p] for p in phase], return_counts=True)))}, terminated {int((w.status != 0).sum())}")
        for k in range(1, 10):
            if np.isfinite(entered[k]).any():
                print(f"  {names[k]:10s} entered at {np.nanmin(entered[k]):6.1f} .. {np.nanmax(entered[k]):6.1f} s")
        print(f"  touchdown {np.nanmin(touchdown[0]):.0f}..{np.nanmax(touchdown[0]):.0f} m from the far end of the final leg (s_2b), cross-track <= {np.nanmax(np.abs(touchdown[1])):.2f} m; "
              f"final ground speed <= {out['v_gnd'].max():.2f} m/s")
    w.close()
    return out


if __name__ == "__main__":
    import time
    n_cli = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    mode_cli = "device" if "device" in sys.argv[2:] else "callback"
    t0 = time.time()
    run(n_cli, verbose=True, mode=mode_cli)
    wall = time.time() - t0
    print(f"{mode_cli}: {n_cli} aircraft x 35000 steps (700 s of flight at 50 Hz, the scenario after every step) in {wall:.1f} s wall, set-up included: {n_cli * 35000 / wall:.3e} aircraft-steps/s")
