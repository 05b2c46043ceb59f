#!/usr/bin/env python3
"""What a launch of the ground-capable Cessna172Xv2 pass costs besides its steps, by batch size: parked aircraft (tools/bench_ground_x2.py's set-up),
1 / 2 / 4 / 8 steps per launch, a + b k fitted per size. One workgroup (256 aircraft) alone shows the latency chain of a launch; one workgroup per
CU (65 536) adds what 256 prologues at once cost the memory system; 4 and 16 per CU add the dispatch of the later rounds.
python3 tools/ground_launch_anatomy.py [Δt/dt]"""
import ctypes as C
import os
import sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, os.path.join(R, "examples"))
import flightbatch as fb
import traffic_pattern as tpat
K = fb.K


def parked(n, ratio=1):
    w = fb.Cessna172Xv2World(n)
    w.set_params(h_terrain=tpat.H_ORTH)
    sim = fb.Simulation(w, dt=0.02, Δt=0.02 * ratio, save_on=False, steps_per_launch=1)
    LOC, PSI = tpat.LOC, tpat.PSI
    n_e = np.array([np.cos(LOC[0]) * np.cos(LOC[1]), np.cos(LOC[0]) * np.sin(LOC[1]), np.sin(LOC[0])])
    fb.init(sim, fb.TrimParameters(n_e=n_e, h_e=1000.0)); fb.f_ode(w)
    geoid = float((w.y[K["FB_Y_KIN"] + 20] - w.y[K["FB_Y_KIN"] + 21])[0])
    x = np.zeros((K["FB_X2_NX"], n)); x[K["FB_X_FUEL"]] = 0.5
    kq = K["FB_X2_KIN"]
    x[kq:kq + 4] = np.array([np.cos(PSI / 2), 0, 0, np.sin(PSI / 2)])[:, None]
    a = -(LOC[0] + np.pi / 2)
    qz = np.array([np.cos(LOC[1] / 2), 0, 0, np.sin(LOC[1] / 2)]); qy = np.array([np.cos(a / 2), 0, np.sin(a / 2), 0])
    x[kq + 4:kq + 8] = np.array([qz[0] * qy[0], -qz[3] * qy[2], qz[0] * qy[2], qz[3] * qy[0]])[:, None]
    x[kq + 8] = tpat.H_ORTH + geoid + 1.81
    u = np.zeros((K["FB_NU"], n)); u[K["FB_U_MIXTURE"]] = 0.5; u[K["FB_U_M_PILOT"]] = 75; u[K["FB_U_BRAKE_LEFT"]] = 1; u[K["FB_U_BRAKE_RIGHT"]] = 1
    w.set_state(x, np.zeros((2, n), dtype=np.int32)); w.u = u
    w.ui = np.full(n, K["FB_UI_MIXTURE_AUTO"] | K["FB_UI_STEERING_ENGAGED"], dtype=np.int32)
    fb.f_init(w, None)
    return w


def main():
    ratio = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    print(f"# control laws every {ratio} step(s); per-launch HIP events (fb_timing_begin_per_launch), median of the launches of 1.6 s of flight; both passes of a launch")
    for n in (256, 16384, 65536, 262144, 1048576):
        w = parked(n, ratio)
        row = []
        for k in (1, 2, 4, 8):
            sim = fb.Simulation(w, dt=0.02, Δt=0.02 * ratio, save_on=False, steps_per_launch=k)
            fb.step(sim, 1.6); w.sync()
            nl = 80 // k
            fb.lib.fb_timing_begin_per_launch(w._h, nl)
            fb.step(sim, 1.6); w.sync()
            tot = C.c_float(); cnt = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(tot), C.byref(cnt))
            ms = (C.c_float * nl)(); got = C.c_int64()
            fb._lib.check(fb.lib.fb_timing_launches(w._h, ms, nl, C.byref(got)))
            row.append(float(np.median(np.array(ms[:got.value]))))
        b, a = np.polyfit([1, 2, 4, 8], row, 1)
        wg = (n + 255) // 256
        print(f"n = {n:8d} ({wg:5d} workgroups, {wg / 256:6.2f} per CU): " + "  ".join(f"k={k}: {m * 1e3:8.1f} us" for k, m in zip((1, 2, 4, 8), row)) +
              f"   => {a * 1e3:7.1f} us per launch + {b * 1e3:7.1f} us per step")
        w.close()


if __name__ == "__main__":
    main()
