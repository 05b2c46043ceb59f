#!/usr/bin/env python3
"""configs[3]'s per-GPU share (524 288 Cessna172Xv2, 50-step launches, autopilot every 2 steps) timed on four batches:
identical aircraft in one mode pair (bench.py's extra.x2: the SURVEY's scenario), randomised trims in that one mode pair,
identical trims with every aircraft in its own mode pair, and both (bench.py's extra.x2_lattice).
    [X2_RATIO=50] python tools/bench_x2_divergence.py [which ...]       which: identical trim modes both"""
import ctypes as C
import os
import sys
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402

ratio = int(os.environ.get("X2_RATIO", "2"))       # control period in steps (50: the launch holds one update — stepping alone)
args = types.SimpleNamespace(x2_inner=50, x2_ratio=ratio)
print(f"control period {ratio} dt")
for which in (sys.argv[1:] or ["identical", "trim", "modes", "both"]):
    d = bench.time_x2(fb, None, None, C, args, divergent={"identical": False, "both": True}.get(which, which))
    print(f"{which:9s} {d['kernel_ms']:8.3f} ms per launch  {d['value']:.4e} aircraft-steps/s  terminated {d['config']['terminated_aircraft']}", flush=True)
