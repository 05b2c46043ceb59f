#!/usr/bin/env python3
"""Config 4 of BASELINE.json on ONE GPU's share (524 288 Cessna172Xv2, dt = 0.01, Δt = 0.02, README example 2): bench.py's extra.x2 leg
alone, for quick A/B runs and rocprofv3:
    python3 tools/bench_x2.py [rk4 steps per launch] [--no-parity] [--pre-sleep SECONDS] [--warm-cap N] [--blocks N]
--no-parity leaves the 512-aircraft parity sample out, so that a `rocprofv3 --kernel-trace --stats` summary of this command holds ONE
population of k_step_duo<0, true, false> launches (the warm-up + timed launches of the 524 288-aircraft batch). Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("inner", nargs="?", type=int, default=50)
ap.add_argument("--no-parity", action="store_true")
ap.add_argument("--pre-sleep", type=float, default=0.0, help="seconds of idle GPU in front of the warm-up (the driver's bench run had ~20 s of CPU legs there)")
ap.add_argument("--warm-cap", type=int, default=30)
ap.add_argument("--blocks", type=int, default=12)
a = ap.parse_args()
args = types.SimpleNamespace(x2_inner=a.inner, x2_pre_sleep=a.pre_sleep, x2_warm_cap=a.warm_cap, x2_blocks=a.blocks)
timed = bench.time_x2(fb, None, None, C, args)
print(json.dumps(timed if a.no_parity else bench.extra_x2(fb, C, args, timed)))
