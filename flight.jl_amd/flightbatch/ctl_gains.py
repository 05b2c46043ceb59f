"""Packs the Cessna172X autopilot gain lookups (the ten HDF5 files of lib/FlightApps/src/c172/c172x/control/data, loaded in
the reference by build_lookup_lqr / build_lookup_pid, lib/FlightPhysics/src/control.jl:879-994; call sites
c172x_ctl.jl:208-213, 816-819) into the FB_TABLE_CTL_GAINS blob described in include/flightbatch.h."""
from __future__ import annotations

import os
import numpy as np

try:
    from . import hdf5_min
except ImportError:   # loaded as a plain module by the oracle tests
    import hdf5_min

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "c172x_ctl")
# blob order; file names are ASCII renderings of the reference's te2te, tv2te, vh2te, q2e, c2θ, v2t, ar2ar, φβ2ar, p2φ, χ2φ
LOOKUPS = (("te2te", "lqr"), ("tv2te", "lqr"), ("vh2te", "lqr"), ("q2e", "pid"), ("c2theta", "pid"), ("v2t", "pid"),
           ("ar2ar", "lqr"), ("phibeta2ar", "lqr"), ("p2phi", "pid"), ("chi2phi", "pid"))
INDEX = {name: k for k, (name, _) in enumerate(LOOKUPS)}


def _grid(d):
    b = d["bounds"]                       # 2 x D (Julia orientation): column k = (lo, hi) of dimension k
    assert b.shape == (2, 2), "the autopilot lookups are 2-D (EAS, h)"
    return b


def load_lqr(path: str) -> dict:
    """Arrays in Julia orientation: K_fbk [NU, NX, nE, nH], K_fwd/K_int [NU, NZ, nE, nH], x/u/z_trim [N, nE, nH]."""
    d = hdf5_min.read_all(path)
    out = {k: d["data/" + k] for k in ("K_fbk", "K_fwd", "K_int", "x_trim", "u_trim", "z_trim")}
    out["bounds"] = _grid(d)
    return out


def load_pid(path: str) -> dict:
    d = hdf5_min.read_all(path)
    out = {"k_p": d["data/k_p"], "k_i": d["data/k_i"], "k_d": d["data/k_d"], "tau_f": d["data/τ_f"], "bounds": _grid(d)}
    return out


def _pack(fields, bounds) -> np.ndarray:
    nE, nH = fields[0].shape[-2:]
    rec = np.concatenate([f.reshape(-1, nE, nH, order="F") for f in fields], axis=0)     # [rec, nE, nH]; matrices column-major
    body = np.transpose(rec, (2, 1, 0)).reshape(-1)                                      # h slowest, EAS, record fastest
    hdr = np.array([nE, nH, bounds[0, 0], bounds[1, 0], bounds[0, 1], bounds[1, 1]], dtype=np.float64)
    return np.concatenate([hdr, body])


def ctl_gains_blob(data_dir: str | None = None) -> np.ndarray:
    data_dir = data_dir or DATA_DIR
    parts = []
    for name, kind in LOOKUPS:
        path = os.path.join(data_dir, name + ".h5")
        if kind == "lqr":
            d = load_lqr(path)
            parts.append(_pack([d["K_fbk"], d["K_fwd"], d["K_int"], d["x_trim"], d["u_trim"], d["z_trim"]], d["bounds"]))
        else:
            d = load_pid(path)
            parts.append(_pack([d["k_p"][None], d["k_i"][None], d["k_d"][None], d["tau_f"][None]], d["bounds"]))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float64)
