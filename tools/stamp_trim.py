"""Where k_trim's first wave spends its cycles (diagnostic build):
    python __graft_entry__.py --diagnostic-variant trimstamp -DFB_TRIM_STAMP
    FLIGHTBATCH_LIB=flight.jl_amd/libflightbatch_trimstamp.so python tools/stamp_trim.py"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd")); sys.path.insert(0, R)
import flightbatch as fb  # noqa: E402
import bench  # noqa: E402
fb.lib.fb_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int32]
n = 1 << 20
w = fb.BatchedWorld(n)
EAS, h, psi, _ = bench.lattice(0, n)
fb.lib.fb_debug_stamps(None, None, 1)
fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi)); w.sync()
acc = (C.c_ulonglong * 32)(); cnt = (C.c_ulonglong * 32)()
fb.lib.fb_debug_stamps(acc, cnt, 0)
names = ["other (Jacobian columns, candidate, bookkeeping)", "residual evaluations", "active-set solver", "serving finished lanes (results, next aircraft, first residual)"]
tot = sum(acc[24 + k] for k in range(4))
for k in range(4):
    print("%-62s %12d cycles %5.1f %%   %7d intervals" % (names[k], acc[24 + k], 100.0 * acc[24 + k] / max(tot, 1), cnt[24 + k]))
print("success %.6f" % w.trim_success.mean())
