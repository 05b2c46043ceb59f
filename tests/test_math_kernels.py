"""The one self-contained math routine of the stepping kernels that the library's own does not stand behind: sincos_step
(c172_device_impl.inc), used by the NED mechanisation and the guidance in place of sincos / tan / cos. Its TEXT is cut out of the device
source, compiled for the host (the builtins it uses are gcc's too) and measured in ulps against 80-bit sinl / cosl — so the accuracy its
comment claims is that of the shipped code, not of a copy."""
import ctypes
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "flight.jl_amd", "csrc", "c172_device_impl.inc")

HARNESS = r"""
#include <cmath>
#define FBD static inline
%s
extern "C" void sincos_step_ulps(const double* x, long n, double* es, double* ec) {
    for (long i = 0; i < n; i++) {
        double s, c;
        sincos_step(x[i], s, c);
        const long double rs = sinl((long double)x[i]), rc = cosl((long double)x[i]);
        es[i] = (double)(fabsl((long double)s - rs) / (long double)std::ldexp(1.0, std::ilogb((double)fabsl(rs)) - 52));
        ec[i] = (double)(fabsl((long double)c - rc) / (long double)std::ldexp(1.0, std::ilogb((double)fabsl(rc)) - 52));
    }
}
"""


def test_sincos_step_accuracy(tmp_path):
    text = open(SRC, encoding="utf-8").read()
    m = re.search(r"FBD void sincos_step\(double x, double& s, double& c\) \{.*?\n\}\n", text, flags=re.S)
    assert m, "sincos_step not found in c172_device_impl.inc"
    src = tmp_path / "h.cpp"
    src.write_text(HARNESS % m.group(0))
    so = tmp_path / "h.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    lib = ctypes.CDLL(str(so))
    P = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(5)
    worst = {}
    for name, x in (("|x| <= pi (the Euler angles, latitude, longitude)", rng.uniform(-np.pi, np.pi, 400000)),
                    ("|x| <= 1e3", rng.uniform(-1e3, 1e3, 400000)),
                    ("|x| <= 1e6 (a heading wound up through 160 000 turns)", rng.uniform(-1e6, 1e6, 400000)),
                    ("near the multiples of pi/2", np.concatenate([k * (np.pi / 2) + rng.uniform(-1e-6, 1e-6, 2000) for k in range(-40, 41)])),
                    ("quadrant boundaries", np.concatenate([(k + 0.5) * (np.pi / 2) + rng.uniform(-1e-9, 1e-9, 500) for k in range(-40, 41)]))):
        x = np.ascontiguousarray(x)
        es = np.empty_like(x); ec = np.empty_like(x)
        lib.sincos_step_ulps(x.ctypes.data_as(P), ctypes.c_long(x.size), es.ctypes.data_as(P), ec.ctypes.data_as(P))
        worst[name] = (es.max(), ec.max())
        print("%-58s max error: sin %.2f ulp, cos %.2f ulp" % (name, es.max(), ec.max()))
        assert es.max() <= 1.5 and ec.max() <= 1.5, (name, es.max(), ec.max())
    # NaN and infinities give NaN, like the library
    x = np.array([np.nan, np.inf, -np.inf]); es = np.empty(3); ec = np.empty(3)
    lib.sincos_step_ulps(x.ctypes.data_as(P), ctypes.c_long(3), es.ctypes.data_as(P), ec.ctypes.data_as(P))
    assert np.isnan(es).all() and np.isnan(ec).all()
