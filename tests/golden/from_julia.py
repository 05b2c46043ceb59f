#!/usr/bin/env python3
"""Reads the raw fixtures tools/gen_golden.jl writes when run on a machine with Julia + Flight.jl (the TRUE reference):

    julia --project tools/gen_golden.jl tests/golden/julia

      x0.f64            27            Cessna172Sv0 state after init!(sim, C172.TrimParameters())      (config 1 of BASELINE.json)
      traj.f64          27 x 11       state every 100 steps of dt = 0.01 (t = 0 ... 10 s), column-major
      xdot0.f64         27            world.ẋ after f_ode!(world) at t = 10 s
      x2_traj.f64       34 x 11       Cessna172Xv2, README example 2, every 200 steps (reference ComponentVector order = C ABI order)
      robot2d_traj.f64  4 x 11        Robot2D mode_v, v_ref = 0.3: ω, v, θ, η every 100 steps

No such files exist in this repository (no Julia toolchain in the build image, and the reference holds no recorded trajectories):
tests/test_julia_fixtures.py skips until someone supplies them, and then compares BOTH the CPU oracle and the GPU path with them at
the north star's 1e-6. Running this file repacks them as tests/golden/julia/reference.npz."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
JULIA_DIR = os.path.join(HERE, "julia")
FILES = {"x0": ("x0.f64", (27,)), "traj": ("traj.f64", (27, 11)), "xdot_end": ("xdot0.f64", (27,)),
         "x2_traj": ("x2_traj.f64", (34, 11)), "robot2d_traj": ("robot2d_traj.f64", (4, 11))}


def available(directory: str = JULIA_DIR) -> list:
    return [k for k, (f, _) in FILES.items() if os.path.exists(os.path.join(directory, f))]


def load(directory: str = JULIA_DIR) -> dict:
    """{name: array} for the files present; trajectories come back as [sample, row] (11 x Nx), like the golden .npz's traj[:, :, 0]."""
    out = {}
    for key, (fname, shape) in FILES.items():
        path = os.path.join(directory, fname)
        if not os.path.exists(path):
            continue
        raw = np.fromfile(path, dtype="<f8")
        if raw.size != int(np.prod(shape)):
            raise ValueError(f"{fname}: {raw.size} doubles, expected {shape} — written by a different tools/gen_golden.jl?")
        a = raw.reshape(shape, order="F")                 # Julia writes column-major
        out[key] = a.T.copy() if a.ndim == 2 else a.copy()
    return out


def write_like_julia(directory: str, **arrays) -> None:
    """the inverse (used by the tests to exercise the consuming side with oracle-made stand-ins written to a temp dir)"""
    os.makedirs(directory, exist_ok=True)
    for key, a in arrays.items():
        fname, shape = FILES[key]
        a = np.asarray(a, dtype="<f8")
        col_major = a.T if a.ndim == 2 else a            # [sample, row] -> rows x samples
        assert col_major.shape == shape, (key, col_major.shape, shape)
        col_major.ravel(order="F").tofile(os.path.join(directory, fname))


if __name__ == "__main__":
    d = load(sys.argv[1] if len(sys.argv) > 1 else JULIA_DIR)
    if not d:
        raise SystemExit("no Julia fixtures found (see the module docstring)")
    np.savez_compressed(os.path.join(JULIA_DIR, "reference.npz"), **d)
    print("packed:", {k: v.shape for k, v in d.items()})
