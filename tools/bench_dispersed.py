#!/usr/bin/env python3
"""Times the headline workload (bench.lattice(0): 1 048 576 Cessna172Sv0, 50 steps per launch) with the aircraft PLACED
(a) uniformly over the sphere, (b) over a 10 deg x 10 deg box, (c) all at one point (phi = lambda = 0: what bench.py does, per
SURVEY.md §8d). The EGM96 gather (FP/geodesy.jl:186-211: bilinear on a 721 x 1441 Float32 grid, 4.2 MB) is the only
data-dependent global-memory access of the airborne path; in (c) every gather of every wave hits one 0.25 deg cell.

    python tools/bench_dispersed.py [launches] [placement ...]     placements: sphere box point

Prints ms per launch for each placement (HIP events on the stepping stream) and the trim time. Under rocprofv3 run one
placement per process (tools/collect_dispersed.sh)."""
import ctypes as C
import os
import sys
import time
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb
from bench import lattice, N_TOTAL, DT


def placement(kind, n, seed=11):
    rng = np.random.default_rng(seed)
    if kind == "sphere":
        lat = np.arcsin(rng.uniform(-1.0, 1.0, n)); lon = rng.uniform(-np.pi, np.pi, n)
    elif kind == "box":
        lat = np.deg2rad(rng.uniform(40.0, 50.0, n)); lon = np.deg2rad(rng.uniform(0.0, 10.0, n))
    elif kind == "point":
        lat = np.zeros(n); lon = np.zeros(n)
    else:
        raise SystemExit(f"unknown placement {kind}")
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    kinds = sys.argv[2:] or ["point", "box", "sphere"]
    n = int(os.environ.get("DISPERSED_N", N_TOTAL))
    EAS, h, psi, _ = lattice(0, n)
    for kind in kinds:
        w = fb.BatchedWorld(n)
        t0 = time.perf_counter()
        fb.f_init(w, fb.TrimParameters(n_e=placement(kind, n), EAS=EAS, h_e=h, ψ_nb=psi))
        trim_s = time.perf_counter() - t0
        ok = float(w.trim_success.mean())
        sim = fb.Simulation(w, dt=DT, save_on=False, steps_per_launch=50)
        for _ in range(3):
            fb.step(sim, 50 * DT)
        w.sync()
        fb.lib.fb_timing_begin(w._h)
        for _ in range(launches):
            fb.step(sim, 50 * DT)
        w.sync()
        ms = C.c_float(); nl = C.c_int64()
        fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
        bad = int((w.status != 0).sum())
        print(f"{kind:7s} n={n}: {ms.value / max(nl.value, 1):8.3f} ms per 50-step launch ({nl.value} launches), "
              f"{n * 50 * nl.value / (ms.value * 1e-3):.4e} aircraft-steps/s; trim {trim_s:.2f} s (success {ok:.4f}); terminated {bad}", flush=True)
        w.close()


if __name__ == "__main__":
    main()
