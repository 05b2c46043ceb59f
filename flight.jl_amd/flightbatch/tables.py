"""Host-side construction of the lookup tables the reference builds at model-construction time.

The kernels only *evaluate* tables; building them is init-time host work, exactly as in the reference
(SURVEY.md Appendix B).  A Julia host would hand over its own arrays (`aero_lookup`, `PistonEngineLookup`,
`Propellers.Lookup(...).data`, `egm96` grid); this module is the Python host's equivalent so that the
library can be driven without Julia.  Layouts are those of ``csrc/tables.h``.

Reference (paths relative to the reference repo, lib/...):
  * aero:      FlightApps/src/c172/c172.jl:51-199 (JSBSim C172R data as transcribed there)
  * piston:    FlightPhysics/src/piston.jl:70-195
  * propeller: FlightPhysics/src/propellers.jl:50-107 (airfoil/blade), :131-208 (blade-element
               coefficients), :235-276 (21 x 21 x 1 fixed-pitch lookup)
  * geoid:     FlightPhysics/src/geodesy.jl:163-198 (ww15mgh_le.bin, sha256 checked)
"""
from __future__ import annotations

import hashlib
import os
import numpy as np

_DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
EGM96_SHA256 = "9d190e021672769b508547021bcaebcc7d13558d66d215d019675a5f595f5cae"  # geodesy.jl:169

# ---- layout constants mirrored from csrc/tables.h --------------------------------------------------
AT = dict(GE_K=0, CD_GE_V=13, CL_GE_V=26, DF4_K=39, CD_DF_V=43, CL_DF_V=47, CM_DF_V=51, UNIT3_K=55, CD_DE_V=58,
          CD_BETA_V=61, CD_ALPHA_K=64, CD_ALPHA_DF_V=90, CY_BETA_K=194, DF2_K=197, CY_BETA_DF_V=199, ALPHA2_K=205,
          CY_P_V=207, CY_R_V=211, CL_R_V=215, CL_ALPHA_K=219, CL_ALPHA_V=236, SCALARS=270, SIZE=291)
PT = dict(DELTA_WOT_V=0, MU_WOT_V=18, PISTD_N_K=36, PISTD_MU_K=49, PISTD_V=52, PIWOT_N_K=91, PIWOT_D_K=96, PIWOT_V=99,
          F_K=114, PI_RATIO_V=125, SFC_RATIO_V=136, SFC_N_K=147, SFC_PI_K=152, SFC_POW_V=160, SIZE=200)


def load_egm96(path: str | None = None) -> np.ndarray:
    """721 x 1441 float32 geoid heights, returned in Julia (column-major) memory order as a flat-compatible
    Fortran array ``A[lat, lon]`` (geodesy.jl:163-185)."""
    path = path or os.path.join(_DATA_DIR, "ww15mgh_le.bin")
    with open(path, "rb") as f:
        raw = f.read()
    if hashlib.sha256(raw).hexdigest() != EGM96_SHA256:
        raise ValueError("Wrong file hash")  # geodesy.jl:170
    return np.frombuffer(raw, dtype="<f4").reshape((721, 1441), order="F").copy(order="F")


# ===================================== aerodynamics =====================================================
def _deg(v):
    return np.asarray(v, dtype=np.float64) * (np.pi / 180)  # Base.deg2rad


def aero_blob() -> np.ndarray:
    """c172.jl:51-199 packed per csrc/tables.h (2-D tables column-major [n1 x n2])."""
    b = np.zeros(AT["SIZE"])

    def put(key, arr):
        a = np.asarray(arr, dtype=np.float64)
        b[AT[key]:AT[key] + a.size] = a.ravel(order="F")

    ge_x = [0.0000, 0.1000, 0.1500, 0.2000, 0.3000, 0.4000, 0.5000, 0.6000, 0.7000, 0.8000, 0.9000, 1.0000, 1.1000]
    put("GE_K", ge_x)
    put("CD_GE_V", [0.4800, 0.5150, 0.6290, 0.7090, 0.8150, 0.8820, 0.9280, 0.9620, 0.9880, 1.0000, 1.0000, 1.0000, 1.0000])
    put("CL_GE_V", [1.2030, 1.1270, 1.0900, 1.0730, 1.0460, 1.0550, 1.0190, 1.0130, 1.0080, 1.0060, 1.0030, 1.0020, 1.0000])
    put("DF4_K", _deg([0, 10, 20, 30]))
    put("CD_DF_V", [0.0000, 0.0070, 0.0120, 0.0180])
    put("CL_DF_V", [0.0000, 0.2, 0.3, 0.35])
    put("CM_DF_V", [0.0000, -0.0654, -0.0981, -0.1140])
    put("UNIT3_K", [-1.0, 0.0, 1.0])
    put("CD_DE_V", [0.06, 0.0, 0.06])
    put("CD_BETA_V", [0.17, 0.0, 0.17])
    put("CD_ALPHA_K", [-0.0873, -0.0698, -0.0524, -0.0349, -0.0175, 0.0000, 0.0175, 0.0349, 0.0524, 0.0698, 0.0873, 0.1047,
                       0.1222, 0.1396, 0.1571, 0.1745, 0.1920, 0.2094, 0.2269, 0.2443, 0.2618, 0.2793, 0.2967, 0.3142, 0.3316, 0.3491])
    cd_alpha_df = np.array([
        [0.0041, 0.0013, 0.0001, 0.0003, 0.0020, 0.0052, 0.0099, 0.0162, 0.0240, 0.0334, 0.0442, 0.0566, 0.0706, 0.0860, 0.0962, 0.1069, 0.1180, 0.1298, 0.1424, 0.1565, 0.1727, 0.1782, 0.1716, 0.1618, 0.1475, 0.1097],
        [0.0000, 0.0004, 0.0023, 0.0057, 0.0105, 0.0168, 0.0248, 0.0342, 0.0452, 0.0577, 0.0718, 0.0874, 0.1045, 0.1232, 0.1353, 0.1479, 0.1610, 0.1746, 0.1892, 0.2054, 0.2240, 0.2302, 0.2227, 0.2115, 0.1951, 0.1512],
        [0.0005, 0.0025, 0.0059, 0.0108, 0.0172, 0.0251, 0.0346, 0.0457, 0.0583, 0.0724, 0.0881, 0.1053, 0.1240, 0.1442, 0.1573, 0.1708, 0.1849, 0.1995, 0.2151, 0.2323, 0.2521, 0.2587, 0.2507, 0.2388, 0.2214, 0.1744],
        [0.0014, 0.0041, 0.0084, 0.0141, 0.0212, 0.0299, 0.0402, 0.0521, 0.0655, 0.0804, 0.0968, 0.1148, 0.1343, 0.1554, 0.1690, 0.1830, 0.1975, 0.2126, 0.2286, 0.2464, 0.2667, 0.2735, 0.2653, 0.2531, 0.2351, 0.1866],
    ]).T  # 26 x 4
    put("CD_ALPHA_DF_V", cd_alpha_df)
    put("CY_BETA_K", [-0.3490, 0, 0.3490])
    put("DF2_K", _deg([0, 30]))
    put("CY_BETA_DF_V", np.array([[0.1370, 0.1060], [0.0000, 0.0000], [-0.1370, -0.1060]]))
    put("ALPHA2_K", [0.0, 0.094])
    put("CY_P_V", np.array([[-0.0750, -0.1610], [-0.1450, -0.2310]]))
    put("CY_R_V", np.array([[0.2140, 0.1620], [0.2670, 0.2150]]))
    put("CL_R_V", np.array([[0.0798, 0.1246], [0.1869, 0.2317]]))
    put("CL_ALPHA_K", [-0.0900, 0.0000, 0.0900, 0.1000, 0.1200, 0.1400, 0.1600, 0.1700, 0.1900, 0.2100, 0.2400, 0.2600, 0.2800, 0.3000, 0.3200, 0.3400, 0.3600])
    cl_alpha = np.array([
        [-0.2200, 0.2500, 0.7300, 0.8300, 0.9200, 1.0200, 1.0800, 1.1300, 1.1900, 1.2500, 1.3500, 1.4400, 1.4700, 1.4300, 1.3800, 1.3000, 1.1500],
        [-0.2200, 0.2500, 0.7300, 0.7800, 0.7900, 0.8100, 0.8200, 0.8300, 0.8500, 0.8600, 0.8800, 0.9000, 0.9200, 0.9500, 0.9900, 1.0500, 1.1500],
    ]).T  # 17 x 2
    put("CL_ALPHA_V", cl_alpha)
    # scalar derivatives in the order of csrc/tables.h (AS_*)
    put("SCALARS", [0.027, 0.1870, 0.0, 0.4300, 3.900, 1.700, 0.229, 0.0147, -0.09226, -0.4840,
                    0.100, -1.1220, -1.8000, -12.400, -7.2700, -0.0430, -0.0053, 0.05874, -0.0278, -0.0937])
    return b


# ===================================== interpolation helpers ============================================
def _grid_interp(knots, vals, x, lo_flat, hi_flat):
    """Interpolations.jl Gridded(Linear()) 1-D with Flat / Line extrapolation per side."""
    k = np.asarray(knots, dtype=np.float64)
    v = np.asarray(vals, dtype=np.float64)
    if lo_flat:
        x = max(x, k[0])
    if hi_flat:
        x = min(x, k[-1])
    i = int(np.clip(np.searchsorted(k, x, side="right") - 1, 0, len(k) - 2))
    w = (x - k[i]) / (k[i + 1] - k[i])
    return (1 - w) * v[i] + w * v[i + 1]


def _range_interp2(a1, b1, n1, a2, b2, n2, data, x1, x2):
    """scale(interpolate(A, BSpline(Linear())), range, range) with Line() extrapolation. data[n1, n2]."""
    def locate(a, b, n, x):
        xi = (x - a) / ((b - a) / (n - 1))
        i = int(np.clip(np.floor(xi), 0, n - 2))
        return i, xi - i
    i, wi = locate(a1, b1, n1, x1)
    j, wj = locate(a2, b2, n2, x2)
    return (1 - wi) * ((1 - wj) * data[i, j] + wj * data[i, j + 1]) + wi * ((1 - wj) * data[i + 1, j] + wj * data[i + 1, j + 1])


def _grid_interp2(k1, k2, data, x1, x2, flat1=(True, True), flat2=(True, True)):
    k1 = np.asarray(k1, dtype=np.float64)
    k2 = np.asarray(k2, dtype=np.float64)
    def locate(k, x, fl):
        if fl[0]:
            x = max(x, k[0])
        if fl[1]:
            x = min(x, k[-1])
        i = int(np.clip(np.searchsorted(k, x, side="right") - 1, 0, len(k) - 2))
        return i, (x - k[i]) / (k[i + 1] - k[i])
    i, wi = locate(k1, x1, flat1)
    j, wj = locate(k2, x2, flat2)
    return (1 - wi) * ((1 - wj) * data[i, j] + wj * data[i, j + 1]) + wi * ((1 - wj) * data[i + 1, j] + wj * data[i + 1, j + 1])


# ===================================== piston engine ====================================================
def piston_blob(n_stall: float = 300.0 / 2700.0, n_max: float = 3100.0 / 2700.0) -> np.ndarray:
    """PistonEngineLookup(n_stall, n_max), piston.jl:70-195, packed per csrc/tables.h."""
    assert n_stall < 0.667 and n_max > 1.074  # piston.jl:72-73
    b = np.zeros(PT["SIZE"])

    def put(key, arr):
        a = np.asarray(arr, dtype=np.float64)
        b[PT[key]:PT[key] + a.size] = a.ravel(order="F")

    delta_wot = np.array([[0.455, 0.523, 0.587, 0.652, 0.718, 0.781, 0.844, 0.906, 0.965],
                          [0.464, 0.530, 0.596, 0.662, 0.727, 0.792, 0.855, 0.921, 0.981]])
    put("DELTA_WOT_V", delta_wot)
    # μ_wot: inverse interpolation of δ_wot row by row, resampled on δ ∈ range(0.441, 1, 9)
    n_range = np.linspace(0.667, 1.0, 2)
    d_range = np.linspace(0.441, 1.0, 9)
    mu_knots = np.linspace(0.401, 0.936, 9)
    mu_wot = np.zeros((2, 9))
    for i, n in enumerate(n_range):
        d_at_knots = [_range_interp2(0.667, 1.0, 2, 0.401, 0.936, 9, delta_wot, n, m) for m in mu_knots]
        mu_wot[i, :] = [_grid_interp(d_at_knots, mu_knots, d, False, False) for d in d_range]
    put("MU_WOT_V", mu_wot)
    # π_std
    n_data = [n_stall, 0.667, 0.704, 0.741, 0.778, 0.815, 0.852, 0.889, 0.926, 0.963, 1.000, 1.074, n_max]
    mu_data = [0, 0.568, 1.0]
    mu_kn = np.array([[0.0] * 13, [0.568] * 13,
                      [1.000, 0.836, 0.854, 0.874, 0.898, 0.912, 0.939, 0.961, 0.959, 0.958, 0.956, 0.953, 1.000]])
    pi_kn = np.array([[0.0] * 13,
                      [0, 0.270, 0.305, 0.335, 0.360, 0.380, 0.405, 0.428, 0.450, 0.476, 0.498, 0.498, 0],
                      [0, 0.489, 0.548, 0.609, 0.680, 0.729, 0.810, 0.880, 0.920, 0.965, 1.000, 0.950, 0]])
    pi_std = np.zeros((13, 3))
    for i in range(13):
        pi_std[i, :] = [_grid_interp(mu_kn[:, i], pi_kn[:, i], m, False, False) for m in mu_data]
    put("PISTD_N_K", n_data)
    put("PISTD_MU_K", mu_data)
    put("PISTD_V", pi_std)
    # π_wot
    n5 = [n_stall, 0.667, 1.000, 1.074, n_max]
    d3 = [0, 0.441, 1]
    pi_wot = np.zeros((5, 3))
    pi_wot[:, 1] = [0, 0.23, 0.409, 0.409, 0]
    pi_wot[:, 2] = [_grid_interp2(n_data, mu_data, pi_std, n, _range_interp2(0.667, 1.0, 2, 0.441, 1.0, 9, mu_wot, n, 1.0)) for n in n5]
    put("PIWOT_N_K", n5)
    put("PIWOT_D_K", d3)
    put("PIWOT_V", pi_wot)
    f_data = [0.0580] + list(np.linspace(0.0625, 0.0950, 10))
    put("F_K", f_data)
    put("PI_RATIO_V", [0.000, 0.8600, 0.9492, 0.9776, 0.9933, 1.000, 0.9983, 0.9910, 0.9798, 0.9657, 0.9500])
    put("SFC_RATIO_V", [5, 0.8700, 0.8524, 0.8818, 0.9261, 0.9839, 1.0510, 1.1279, 1.2135, 1.3163, 1.4280])
    put("SFC_N_K", np.array([2000, 2200, 2400, 2600, 2700]) / 2700)
    put("SFC_PI_K", 10.0 ** np.linspace(-1, 0, 8))
    put("SFC_POW_V", 1e-7 * np.array([
        [1.7671, 1.43728, 1.19992, 1.02909, 0.906153, 0.817674, 0.753997, 0.708169],
        [1.83791, 1.49664, 1.25103, 1.07427, 0.947056, 0.855503, 0.789613, 0.742193],
        [1.98614, 1.60588, 1.3322, 1.13524, 0.993496, 0.891482, 0.818064, 0.765226],
        [2.11663, 1.70062, 1.40123, 1.18576, 1.03069, 0.919083, 0.838765, 0.780961],
        [2.33484, 1.85418, 1.50825, 1.2593, 1.08012, 0.951177, 0.858376, 0.791588]]))
    return b


# ===================================== propeller (blade element) ========================================
_ALPHA0 = -2.1 * (np.pi / 180)  # DefaultAirfoil zero-lift angle, propellers.jl:48


def _cL(al, M):  # propellers.jl:50-58 (vectorised)
    def sub(al, M):
        return np.where(al < 0.25, 2 * np.pi * al, np.pi / 2 * np.cos(al) / np.cos(0.25)) / np.sqrt(1 - M ** 2)
    def sup(al, M):
        return np.where(al < 0.25, 4 * al, np.cos(al) / np.cos(0.25)) / np.sqrt(M ** 2 - 1)
    lo, hi = sub(al, 0.8), sup(al, 1.2)
    mid = lo + (hi - lo) / 0.4 * (M - 0.8)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(M <= 0.8, sub(al, np.minimum(M, 0.8)), np.where(M >= 1.2, sup(al, np.maximum(M, 1.2)), mid))


def _cL_alpha(al, M):  # propellers.jl:60-68
    def sub(al, M):
        return np.where(al < 0.25, 2 * np.pi, -np.pi / 2 * np.sin(al) / np.cos(0.25)) / np.sqrt(1 - M ** 2)
    def sup(al, M):
        return np.where(al < 0.25, 4.0, -np.sin(al) / np.cos(0.25)) / np.sqrt(M ** 2 - 1)
    lo, hi = sub(al, 0.8), sup(al, 1.2)
    mid = lo + (hi - lo) / 0.4 * (M - 0.8)
    return np.where(M <= 0.8, sub(al, np.minimum(M, 0.8)), np.where(M >= 1.2, sup(al, np.maximum(M, 1.2)), mid))


def _cD(al, M):  # propellers.jl:70-94
    cD_inc = np.where(al < 0.25, 0.006 + 0.224 * al ** 2, np.where(al < 0.3, -1.0234 + 16.6944 * al ** 2, np.pi / 2 * np.sin(al) / np.cos(0.25)))
    k_dd = np.where(M <= 0.8, 1.0, np.where(M <= 0.95, 1.0 + 160000 * (M - 0.8) ** 4 / 27, np.where(M <= 1.0, 6.0 - 800 * (1 - M) ** 2, 6 - 5 * (M - 1))))
    return k_dd * cD_inc


def _M_section(J, Mt, z, eps_i):  # propellers.jl:198-202
    return Mt * np.sqrt((np.pi ** 2 * z ** 2 + J ** 2) / (np.pi ** 2 + J ** 2)) * np.cos(eps_i)


def propeller_coefficients(n_blades: int, J, Mt, dbeta: float = 0.0, n_zeta: int = 101, zeta_h: float = 0.2,
                           chord_a: float = 0.075, pitch: float = 0.8):
    """Coefficients(n_blades, Blade(), J, Mt, Δβ, n_ζ) for arrays J, Mt (same shape) — propellers.jl:131-196.

    The induced-angle equation (:204-208) is solved per radial station, warm-started from the previous
    station's root as the reference does (`find_zero(f, ε_i)`), here by a vectorised secant iteration with
    a bisection fallback.
    """
    J = np.asarray(J, dtype=np.float64)
    Mt = np.asarray(Mt, dtype=np.float64)
    shape = J.shape
    J = J.ravel()
    Mt = Mt.ravel()
    zs = np.linspace(zeta_h, 1.0, n_zeta)
    beta_a_t = np.arctan(pitch / (np.pi * 1.0)) + dbeta - _ALPHA0
    d = {k: np.zeros((n_zeta, J.size)) for k in ("Fx", "Mx", "Fz", "Mz")}
    eps_i = np.ones_like(J)
    for iz, z in enumerate(zs):
        eps_inf = np.arctan(J / (np.pi * z))
        beta_a = np.arctan(pitch / (np.pi * z)) + dbeta - _ALPHA0
        c = chord_a * np.sqrt(1 - z ** 2)
        kprandtl = np.arccos(np.exp(-n_blades * (1 - z) / (2 * np.sin(beta_a_t))))

        def f(e):
            al = beta_a - eps_inf - e
            return n_blades * c / (8 * z) * _cL(al, _M_section(J, Mt, z, e)) - kprandtl * np.tan(e) * np.sin(eps_inf + e)

        x0 = eps_i.copy()
        f0 = f(x0)
        done = f0 == 0.0
        x1 = x0 + (np.abs(x0) * 1e-4 + 1e-6)
        f1 = f(x1)
        for _ in range(100):
            if done.all():
                break
            with np.errstate(invalid="ignore", divide="ignore"):
                x2 = x1 - f1 * (x1 - x0) / (f1 - f0)
            x2 = np.where(np.isfinite(x2), x2, x1)
            x2 = np.clip(x2, x1 - 0.5, x1 + 0.5)
            f2 = f(x2)
            newly = (~done) & ((f2 == 0.0) | (np.abs(x2 - x1) <= 4e-16 * np.maximum(np.abs(x2), 1e-300)))
            x0 = np.where(done, x0, x1); f0 = np.where(done, f0, f1)
            x1 = np.where(done, x1, x2); f1 = np.where(done, f1, f2)
            done = done | newly
        if not done.all():  # polish the stragglers by bisection on a local bracket
            from scipy.optimize import brentq
            for k in np.nonzero(~done)[0]:
                def fk(e, k=k):
                    al = beta_a - eps_inf[k] - e
                    return n_blades * c / (8 * z) * float(_cL(al, _M_section(J[k], Mt[k], z, e))) - kprandtl * np.tan(e) * np.sin(eps_inf[k] + e)
                h = 1e-3
                a_, b_ = x1[k] - h, x1[k] + h
                while fk(a_) * fk(b_) > 0 and h < 1.0:
                    h *= 2
                    a_, b_ = x1[k] - h, x1[k] + h
                x1[k] = brentq(fk, a_, b_, xtol=1e-16, rtol=4e-16)
        eps_i = x1
        eps = eps_inf + eps_i
        al = beta_a - eps
        M = _M_section(J, Mt, z, eps_i)
        kc = n_blades * c
        ce, se = np.cos(eps), np.sin(eps)
        c2i, c2inf = np.cos(eps_i) ** 2, np.cos(eps_inf) ** 2
        t, t2 = np.tan(eps_inf), np.tan(eps_inf) ** 2
        cl, cd, cla = _cL(al, M), _cD(al, M), _cL_alpha(al, M)
        p2 = np.pi ** 2
        d["Fx"][iz] = p2 / 4 * z ** 2 * kc * c2i / c2inf * (cl * ce - cd * se)
        d["Mx"][iz] = -p2 / 8 * z ** 3 * kc * c2i / c2inf * (cd * ce + cl * se)
        d["Fz"][iz] = -p2 / 8 * z ** 2 * kc * c2i * (2 * t * (cd * ce + cl * se) - t2 * (cl * ce - (cla + cd) * se))
        d["Mz"][iz] = -p2 / 16 * z ** 3 * kc * c2i * (2 * t * (cl * ce - cd * se) + t2 * ((cla + cd) * ce + cl * se))
    trap = lambda y: np.trapezoid(y, zs, axis=0) if hasattr(np, "trapezoid") else np.trapz(y, zs, axis=0)
    C_Fx, C_Mx, C_Fz, C_Mz = (trap(d[k]) for k in ("Fx", "Mx", "Fz", "Mz"))
    C_P = 2 * np.pi * C_Mx
    with np.errstate(invalid="ignore", divide="ignore"):
        eta = np.where(C_Fx > 0, -J * C_Fx / C_P, 0.0)
    return tuple(a.reshape(shape) for a in (C_Fx, C_Mx, C_Fz, C_Mz, C_P, eta))


def propeller_table(n_blades: int = 2) -> np.ndarray:
    """Lookup(2, Blade(); J ∈ range(0,1.5,21), Mt ∈ range(0,1.5,21), Δβ = 0) — propellers.jl:235-250.
    Returns [21, 21, 6] in Fortran order: (J, Mt, {C_Fx, C_Mx, C_Fz_α, C_Mz_α, C_P, η_p})."""
    Jg, Mg = np.meshgrid(np.linspace(0, 1.5, 21), np.linspace(0, 1.5, 21), indexing="ij")
    coeffs = propeller_coefficients(n_blades, Jg, Mg)
    out = np.zeros((21, 21, 6), order="F")
    for c in range(6):
        out[:, :, c] = coeffs[c]
    return out


_CACHE: dict = {}


def default_tables() -> dict:
    """All four tables for Cessna172Sv0 (c172s.jl:16-34 power plant defaults)."""
    if not _CACHE:
        _CACHE.update(egm96=load_egm96(), aero=aero_blob(), piston=piston_blob(), propeller=propeller_table(2))
    return _CACHE
