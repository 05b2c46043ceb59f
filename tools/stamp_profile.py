#!/usr/bin/env python3
"""Per-phase cycle profile of ONE wave of the airborne stepper (diagnostic build with -DFB_STAMP, see c172_device_impl.inc):

    hipcc ... -DFB_STAMP -o flight.jl_amd/libflightbatch_stamp.so   (tools/build_variant.sh stamp -DFB_STAMP)
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_stamp.so python tools/stamp_profile.py

Prints, per fence k, the shader cycles wave 0 of workgroup 0 spent between the previous fence and fence k, per evaluation."""
import ctypes as C
import json
import os
import sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=EAS[:n], h_e=h[:n], ψ_nb=psi[:n]))
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 1.0); w.sync()
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
fb.lib.fb_debug_stamps(None, None, 1)
fb.step(sim, 2.0); w.sync()
fb.lib.fb_debug_stamps(acc, cnt, 0)
names = {0: "loop tail + emit setup (20 -> 0)", 11: "kinematics head: attitude, n_e, lat/lon atan2, geoid", 1: "kinematics rest + emits (9 rows)",
         2: "air data", 12: "aero: airflow angles, filters, knot location", 3: "aero: lookups, coefficients, wrench", 4: "gear unit (x3)",
         5: "gear tail", 9: "propeller", 10: "engine head: PI, mixture, locate n / f, pi_ratio", 6: "engine chain: mu_wot .. SFC, emit", 7: "fuel",
         8: "mass properties", 20: "dynamics + emits (6 rows)"}
order = [0, 11, 1, 2, 12, 3, 4, 5, 9, 10, 6, 7, 8, 20]
evals = cnt[0]
tot = sum(acc[k] for k in order)
out = {}
for k in order:
    per = acc[k] / max(evals, 1)
    out[names[k]] = per
    print(f"{k:3d} {names[k]:58s} {per:9.1f} cycles/eval  {100 * acc[k] / tot:5.1f} %")
print(f"total {tot / evals:.1f} cycles per evaluation over {evals} evaluations (stamps included: each costs ~ one memory round trip outside the intervals)")
json.dump(out, open(os.path.join(R, "gpurun_out", "stamp_profile.json"), "w"), indent=1)
w.close()
