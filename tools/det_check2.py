"""Run-to-run determinism probes for the Cessna172Xv2 stepper (same process, fresh worlds): cruise, and a ground roll."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb

def cruise(spl, ratio, T=2.0, n=512):
    w = fb.Cessna172Xv2World(n)
    sim = fb.Simulation(w, dt=0.01, Δt=0.01 * ratio, save_on=False, steps_per_launch=spl)
    fb.init(sim, fb.TrimParameters(EAS=np.linspace(40, 50, n), h_e=np.linspace(500, 2000, n)))
    w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 1.0
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = 0.2
    fb.step(sim, T); w.sync()
    out = (w.x.copy(), w.cs.copy(), w.status.copy()); w.close(); return out

def ground(spl, ratio, T=float(os.environ.get("DET_T", "15")), n=512):
    w = fb.Cessna172Xv2World(n)
    sim = fb.Simulation(w, dt=0.01, Δt=0.01 * ratio, save_on=False, steps_per_launch=spl)
    fb.init(sim, fb.TrimParameters(EAS=np.linspace(30, 34, n), h_e=np.full(n, 8.0), γ_wb_n=-0.05, flaps=1.0))   # 8 m above the terrain (h_trn = 0), descending: touches down
    fb.step(sim, T); w.sync()
    fb.f_ode(w); y = w.y
    out = (w.x.copy(), w.cs.copy(), w.status.copy(), ((y[fb.K["FB_Y_LDG"] + 1] + y[fb.K["FB_Y_LDG"] + 12] + y[fb.K["FB_Y_LDG"] + 23]) > 0).astype(float)); w.close(); return out

for name, fn in (("ground", ground),):
    for spl, ratio in ((1, 1), (1, 2), (50, 2), (7, 2)):
        r = [fn(spl, ratio) for _ in range(3)]
        same = [all(np.array_equal(a, b, equal_nan=True) for a, b in zip(r[0], r[k])) for k in (1, 2)]
        print(f"{name:7s} steps/launch {spl:2d} ratio {ratio}: runs identical {same}, terminated {[int((q[2] != 0).sum()) for q in r]}, on wheels {[int(q[3].sum()) for q in r] if len(r[0]) > 3 else None}, max|dx| {[float(np.nanmax(np.abs(r[0][0] - r[k][0]))) for k in (1, 2)]}", flush=True)
