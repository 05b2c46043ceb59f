#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU ORACLE (oracle/liboracle.so).

These are regression vectors of the oracle, NOT outputs of the Julia reference: Flight.jl cannot be run in
this environment (no Julia toolchain; see DESIGN.md "Oracle") and holds no recorded trajectories of its own.
They serve (a) to detect drift of the oracle between rounds and (b) as inputs/expected outputs for the
`-m gpu` tests on the GPU box. A maintainer with Julia can overwrite them with true reference output using
tools/gen_golden.jl, which writes the same arrays.

    python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_binding import Oracle  # noqa: E402


def default_tp(n):
    tp = np.zeros((18, n)); tp[0] = 1; tp[3] = 1050; tp[5] = 50; tp[10] = 0.5; tp[11] = 0.5
    tp[13:18] = np.array([75, 75, 0, 0, 50.0])[:, None]
    return tp


def main():
    o = Oracle()
    env = o.default_env()
    ts0 = np.repeat(np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])[:, None], 1, axis=1)
    # config 1: single Cessna172Sv0, C172.TrimParameters(), dt = 0.01, t_end = 10 s
    r = o.trim(default_tp(1), ts0, env)
    assert r["ok"].all()
    xd, y, st = o.f_ode(r["x"], r["u"], r["ui"], r["s"], env)
    xf, sf, stf, traj = o.step(r["x"], r["u"], r["ui"], r["s"], env, 0.01, 1000, save_every=100)
    np.savez_compressed(os.path.join(HERE, "c172s0_config1.npz"), trim_params=default_tp(1), trim_state=r["ts"], x0=r["x"], u=r["u"],
                        ui=r["ui"], s0=r["s"], env=env, xdot0=xd, y0=y, dt=0.01, save_every=100, traj=traj, s_final=sf)
    # a small randomised lattice with wind and non-standard sea level, perturbed off trim
    n = 64
    rng = np.random.default_rng(20260630)
    tp = default_tp(n)
    lat = rng.uniform(-1.2, 1.2, n); lon = rng.uniform(-np.pi, np.pi, n)
    tp[0:3] = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)])
    tp[3] = rng.uniform(200, 3000, n); tp[4] = rng.uniform(-np.pi, np.pi, n); tp[5] = rng.uniform(38, 52, n)
    tp[6] = rng.uniform(-0.02, 0.02, n); tp[7] = rng.uniform(-0.03, 0.03, n); tp[10] = rng.uniform(0.1, 1.0, n)
    env2 = o.default_env(T_sl=293.15, p_sl=100500.0, wind=(3.0, -2.0, 0.5))
    r2 = o.trim(tp, np.repeat(ts0, n, axis=1), env2)
    x = r2["x"].copy()
    x[21:24] += rng.normal(0, 0.02, (3, n)); x[24:27] += rng.normal(0, 1.0, (3, n))
    xd2, y2, st2 = o.f_ode(x, r2["u"], r2["ui"], r2["s"], env2)
    xf2, sf2, stf2, traj2 = o.step(x, r2["u"], r2["ui"], r2["s"], env2, 0.01, 500, save_every=100)
    np.savez_compressed(os.path.join(HERE, "c172s0_lattice64.npz"), trim_params=tp, trim_state=r2["ts"], trim_ok=r2["ok"], x0=x, u=r2["u"],
                        ui=r2["ui"], s0=r2["s"], env=env2, xdot0=xd2, y0=y2, dt=0.01, save_every=100, traj=traj2, s_final=sf2)
    print("written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
