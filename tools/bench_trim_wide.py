"""fb_trim on an envelope wider than the aircraft can fly (EAS 24-62 m/s, h 100-4500 m, flaps, climb / descent: a quarter of the points have no trim
and go through the continuation fallback): wall time per batch.   python tools/bench_trim_wide.py [n]"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
rng = np.random.default_rng(5)
lat = rng.uniform(-1.2, 1.2, n); lon = rng.uniform(-np.pi, np.pi, n)
tp = fb.TrimParameters(n_e=np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)]), h_e=rng.uniform(100.0, 4500.0, n),
                       EAS=rng.uniform(24.0, 62.0, n), ψ_nb=rng.uniform(-np.pi, np.pi, n), γ_wb_n=rng.uniform(-0.05, 0.05, n),
                       ψ_wb_dot=rng.uniform(-0.03, 0.03, n), flaps=rng.choice([0.0, 0.33, 1.0], n), fuel_load=rng.uniform(0.1, 1.0, n))
w = fb.BatchedWorld(n)
for rep in range(2):
    t0 = time.time(); fb.f_init(w, tp); w.sync(); dt = time.time() - t0
    print("wide envelope, %d aircraft: fb_trim %.3f s wall, success %.4f, checksum %.12e" % (n, dt, w.trim_success.mean(), float(np.nansum(w.trim_state))))
