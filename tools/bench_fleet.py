#!/usr/bin/env python3
"""Config 5 of BASELINE.json on one GPU: a mixed fleet of N vehicles, 50 % Cessna172Sv0 / 50 % Robot2D interleaved in the input
order, dt = 0.01, Δt = 0.02. The packer (flightbatch/fleet.py) sorts by model into two homogeneous batches on
two HIP streams. Both batches run their fp32 steppers by default (argv[3] = f64 for the fp64 aircraft kernel).
Prints one JSON line: vehicle-steps/s of the whole fleet and of each batch."""
import json
import os
import sys
import time
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
KC, KR = fb.K["FB_MODEL_C172S0"], fb.K["FB_MODEL_ROBOT2D"]
types = np.where(np.arange(n) % 2 == 0, KC, KR)
t0 = time.perf_counter()
DT = sys.argv[3] if len(sys.argv) > 3 else "f32"
fleet = fb.MixedFleet(types, {KC: lambda m: fb.BatchedWorld(m, dtype=DT), KR: lambda m: fb.Robot2DWorld(m, dtype="f32")})
pack_s = time.perf_counter() - t0
fleet.simulate(dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
fleet.init({KC: fb.TrimParameters(), KR: fb.InitParameters()})
rw = fleet.worlds[KR]
u = rw.u; u[0] = 1; u[2] = 0.3; rw.u = u
fleet.step(1.0); fleet.sync()
t0 = time.perf_counter(); fleet.step(T); fleet.sync(); el = time.perf_counter() - t0
steps = int(round(T / 0.01))
# each batch alone, for comparison
t0 = time.perf_counter(); fb.step(fleet.sims[KC], T); fleet.worlds[KC].sync(); el_c = time.perf_counter() - t0
t0 = time.perf_counter(); fb.step(fleet.sims[KR], T); rw.sync(); el_r = time.perf_counter() - t0
st = fleet.gather("status", fill=-1)
print(json.dumps({"metric": "vehicle-steps/sec", "value": n * steps / el, "unit": "vehicle-steps/s", "n_gpus": 1,
                  "config": {"workload": f"mixed fleet N={n}: 50% Cessna172Sv0 ({DT}) / 50% Robot2D (fp32), interleaved input order, dt=0.01, Δt=0.02"},
                  "fleet_s": el, "c172_alone_s": el_c, "robot2d_alone_s": el_r, "overlap_gain": (el_c + el_r) / el, "pack_s": pack_s,
                  "terminated": int((st != 0).sum())}))
fleet.close()
