#!/bin/bash
# How long each role of k_step_duo takes with its SIMD to itself (timing diagnostic, results are NOT physical): ON THE GPU BOX, after
#   python __graft_entry__.py --diagnostic-variant onlyd -DFB_DUO_ONLY=2 ; python __graft_entry__.py --diagnostic-variant onlyp -DFB_DUO_ONLY=1
# prints the stepping kernel's launch time for the shipped library and for the two variants (the partner role only meets the barriers).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out
for tag in main onlyd onlyp; do
  lib=flight.jl_amd/libflightbatch_$tag.so; [ "$tag" = main ] && lib=flight.jl_amd/libflightbatch.so
  FLIGHTBATCH_LIB=$ROOT/$lib timeout -k 10 200 python - <<PY
import ctypes as C, os, sys
sys.path.insert(0, "flight.jl_amd"); sys.path.insert(0, ".")
import flightbatch as fb, bench
n = 1 << 20
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 0.5); w.sync()
fb.lib.fb_timing_begin(w._h)
fb.step(sim, 2.0); w.sync()
ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
print("%-6s %.3f ms per launch (%d launches); aircraft still stepping: %d of %d" % ("$tag", ms.value / nl.value, nl.value, int((w.status == 0).sum()), n))
PY
done
