#!/usr/bin/env python3
"""f_ode!(world) with the full 174-double output record for a batch: calls/s and effective write bandwidth (python3 tools/bench_f_ode.py [n])."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flight.jl_amd"))
import flightbatch as fb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=np.linspace(35, 55, n), h_e=np.linspace(200, 3000, n)))
for _ in range(3): fb.f_ode(w)
w.sync()
t0 = time.perf_counter()
for _ in range(20): fb.f_ode(w)
w.sync()
dt = (time.perf_counter() - t0) / 20
print(f"f_ode!: {dt*1e3:.3f} ms per call for {n} aircraft = {n/dt:.3e} evaluations/s, {n*174*8/dt/1e12:.2f} TB/s of output records")
