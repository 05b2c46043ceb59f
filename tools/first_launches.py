"""What the first stepping launches after fb_trim cost (k_step_duo<0> max 16.4 ms against 14.3 in the kernel stats).
Workload, run under `rocprofv3 --kernel-trace` by tools/first_launches.sh: trim 1 048 576 aircraft, 30 back-to-back 50-step launches,
one second of idle, 30 more, a second trim (the trim workspace, ~100 MB, is allocated by the FIRST fb_trim and stays resident until fb_destroy: the second trim allocates nothing), 30 more.
    python tools/first_launches.py run                 the workload
    python tools/first_launches.py report TRACE.csv    the launches in order: start time, duration, what ran before"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flight.jl_amd")); sys.path.insert(0, ROOT)

if sys.argv[1] == "run":
    import numpy as np
    import flightbatch as fb
    import bench
    EAS, h, psi, _ = bench.lattice(0)
    w = fb.BatchedWorld(bench.N_TOTAL)
    tp = fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi)
    fb.f_init(w, tp)
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    for _ in range(30):
        fb.step(sim, 0.5)
    w.sync()
    time.sleep(1.0)
    for _ in range(30):
        fb.step(sim, 0.5)
    w.sync()
    fb.f_init(w, tp)
    for _ in range(30):
        fb.step(sim, 0.5)
    w.sync()
    assert (w.status == 0).all()
else:
    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end, k = None, 0
    print("#  launch  start_ms  duration_ms  gap_before_ms  previous kernel")
    prev_name = "-"
    for r in rows:
        name = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "k_step_duo" in name:
            k += 1
            gap = (s - prev_end) / 1e6 if prev_end else 0.0
            print(f"{k:4d}  {(s - t0) / 1e6:10.2f}  {(e - s) / 1e6:8.3f}  {gap:10.3f}  {prev_name[:60]}")
        elif "k_trim" in name:
            print(f"      {(s - t0) / 1e6:10.2f}  {(e - s) / 1e6:8.3f}  (k_trim)")
        prev_end, prev_name = e, name
