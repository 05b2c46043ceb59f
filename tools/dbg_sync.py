import ctypes as C, os, sys, time, traceback
import numpy as np
sys.path.insert(0, "flight.jl_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import flightbatch as fb
import test_gpu_duo as T
t0 = time.time()
try:
    T.test_duo_and_air_steppers_agree(fb, int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 50)
    print("test passed")
except BaseException:
    traceback.print_exc(limit=2)
print("took", time.time() - t0)
out = (C.c_uint * 40)()
fb.lib.fb_debug_duo_sync(out)
o = list(out)
print("failures", o[0])
for k in range(0, min(o[1], 28), 4):
    print("block %d thread %d (role %s pair %d) count %d partner %d" % (o[2 + k], o[3 + k], "P" if o[3 + k] < 256 else "D", (o[3 + k] & 255) >> 6, o[4 + k], o[5 + k]))
