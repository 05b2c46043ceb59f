"""The fixed cost of a k_step_duo launch (diagnostic build): when role D's first wave of workgroup 0 passes the phases of a launch.
    python __graft_entry__.py --diagnostic-variant phases -DFB_STAMP -DFB_DUO_PHASES
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_phases.so python tools/duo_phases.py [steps_per_launch=1]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = 1 << 20
EAS, h, psi, _ = bench.lattice(0)
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
fb.step(sim, 0.1 * k); w.sync()
fb.lib.fb_timing_begin(w._h)
fb.step(sim, 0.1 * k); w.sync()
ms = C.c_float(); nl = C.c_int64(); fb.lib.fb_timing_end(w._h, C.byref(ms), C.byref(nl))
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
t = [acc[8 + j] for j in range(8)]
print("launch: %.3f ms per %d step(s) of %d aircraft; workgroup 0, role D's first wave (the LAST launch), shader-clock cycles:" % (ms.value / nl.value, k, n))
for j, name in enumerate(["tables staged (copies, reciprocal spacings, atan table, two barriers)", "state loaded, launch constants formed and stored", "the evaluations", "exit: rows written back"]):
    print("   %-72s %9d" % (name, t[j + 1] - t[j]))
print("   inside the second: state rows loaded %d, input rows loaded %d, aerodynamic / payload sums formed %d, stored + the rest %d" % (t[5] - t[1], t[6] - t[5], t[7] - t[6], t[2] - t[7]))
