"""The fixed cost of a k_step_duo<KIN, true> launch (diagnostic build): when role D's first wave of workgroup 0 passes the phases of a launch.
    python __graft_entry__.py --diagnostic-variant phases -DFB_STAMP -DFB_DUO_PHASES
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_phases.so python tools/duo_phases_x2.py [steps_per_launch=1] [Δt/dt=1] [n=524288]"""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ratio = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 524288
w = fb.Cessna172Xv2World(n)
w.set_params(wind_ned=(1.0, 0.5, 0.0))
sim = fb.Simulation(w, dt=0.02, Δt=0.02 * ratio, save_on=False, steps_per_launch=k)
fb.init(sim, fb.TrimParameters())
K = fb.K
cu = w.cu
cu[K["FB_CU_LON_MODE_REQ"]] = float(fb.ModeControlLon.EAS_clm); cu[K["FB_CU_LAT_MODE_REQ"]] = float(fb.ModeControlLat.φ_β)
w.cu = cu
fb.step(sim, 0.4 * k); w.sync()
rows = []
for rep in range(6):   # the stamps are those of the LAST launch: six of them, one per call
    fb.lib.fb_timing_begin(w._h)
    fb.step(sim, 0.02 * k * (3 + rep % 2)); w.sync()     # (an odd and an even number of launches in turn: with Δt = 2 dt both kinds of last launch)
    ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
    acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
    fb.lib.fb_debug_stamps(acc, cnt, 0)
    t = [acc[8 + j] for j in range(8)]
    rows.append((ms.value / nl.value, t[1] - t[0], t[5] - t[1], t[2] - t[5], t[3] - t[2], t[4] - t[3], t[4] - t[0]))
print("n = %d Cessna172Xv2, %d step(s) per launch, control laws every %d step(s); workgroup 0, role D's first wave, shader-clock cycles of the LAST launch of each call:" % (n, k, ratio))
print("  ms/launch   tables staged   state rows loaded   constants formed + stored   evaluations (+ update)   exit: rows written back   entry -> exit")
for r in rows:
    print("  %8.3f %14d %19d %27d %24d %25d %15d" % r)
