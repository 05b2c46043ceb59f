#!/usr/bin/env python3
"""Config 4 of BASELINE.json on ONE GPU's share (524 288 Cessna172Xv2, dt = 0.01, Δt = 0.02, README example 2): bench.py's extra_x2 leg
alone, for quick A/B runs and rocprofv3 (`python3 tools/bench_x2.py [rk4 steps per launch]`). Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402

args = types.SimpleNamespace(x2_inner=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print(json.dumps(bench.extra_x2(fb, C, args)))
