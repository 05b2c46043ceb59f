"""Time of fb_trim on the bench lattice (1,048,576 aircraft, neighbouring lanes in different (EAS, h) cells) and on a smooth ramp.
    python tools/bench_trim.py            wall times (host copies of 26 doubles per aircraft included)
    rocprofv3 --kernel-trace --stats -d gpurun_out/trim -- python3 tools/bench_trim.py      the kernel's own time (k_trim)"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
w = fb.BatchedWorld(n)
EAS, h, psi, _ = bench.lattice(0, n)
for name, tp in (("lattice", fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi)),
                 ("ramp", fb.TrimParameters(EAS=np.linspace(35, 55, n), h_e=np.linspace(200, 3000, n), ψ_nb=np.linspace(-3, 3, n)))):
    for rep in range(2):
        t0 = time.time(); fb.f_init(w, tp); w.sync(); dt = time.time() - t0
        print("%-8s fb_trim %.4f s wall, success %.6f" % (name, dt, w.trim_success.mean()))
