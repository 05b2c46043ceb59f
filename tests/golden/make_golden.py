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
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "flight.jl_amd", "flightbatch"))
from oracle_binding import Oracle, OracleX, header_enums  # noqa: E402


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
    # config 4 scenario (README example 2) on Cessna172Xv2: default trim, wind N = 1, E = 0.5 m/s, EAS_clm with clm_ref = 2 m/s,
    # φ_β with φ_ref = 30°, dt = 0.01, Δt = 0.02, 20 s; plus 7 aircraft in other mode pairs off the design point
    import ctl_gains
    K = header_enums()
    X = OracleX(o, ctl_gains.ctl_gains_blob())
    envx = o.default_env(wind=(1.0, 0.5, 0.0))
    nx = 8
    tpx = default_tp(nx)
    tpx[3, 1:] = np.linspace(400, 2600, nx - 1); tpx[5, 1:] = np.linspace(43, 53, nx - 1); tpx[4, 1:] = np.linspace(-2.5, 2.5, nx - 1)
    st = X.trim_init(tpx, np.repeat(ts0, nx, axis=1), envx, 0.02)
    assert st["ok"].all()
    st["status"] = np.zeros(nx, np.int32); st["nstep"] = 0
    cu = st["cu"]
    lon = [K["FB_LON_EAS_CLM"], K["FB_LON_SAS"], K["FB_LON_THR_Q"], K["FB_LON_THR_THETA"], K["FB_LON_THR_EAS"], K["FB_LON_EAS_Q"], K["FB_LON_EAS_THETA"], K["FB_LON_EAS_ALT"]]
    lat = [K["FB_LAT_PHI_BETA"], K["FB_LAT_SAS"], K["FB_LAT_P_BETA"], K["FB_LAT_CHI_BETA"], K["FB_LAT_PHI_BETA"], K["FB_LAT_DIRECT"], K["FB_LAT_CHI_BETA"], K["FB_LAT_P_BETA"]]
    cu[K["FB_CU_LON_MODE_REQ"]] = lon; cu[K["FB_CU_LAT_MODE_REQ"]] = lat
    cu[K["FB_CU_CLM_REF"], 0] = 2.0; cu[K["FB_CU_PHI_REF"], 0] = np.deg2rad(30.0)
    cu[K["FB_CU_THETA_REF"], 3] += 0.03; cu[K["FB_CU_EAS_REF"], 4] -= 4.0; cu[K["FB_CU_CHI_REF"], 6] += 0.4; cu[K["FB_CU_H_REF"], 7] += 50.0
    init = {k: np.array(st[k]) for k in ("x", "u", "ui", "s", "cu", "cs")}
    trajx = X.step(st, envx, 0.01, 2, 2000, save_every=200)
    np.savez_compressed(os.path.join(HERE, "c172x2_modes8.npz"), trim_params=tpx, env=envx, dt=0.01, ratio=2, save_every=200, traj=trajx,
                        cs_final=st["cs"], cu_final=st["cu"], s_final=st["s"], **{k + "0": v for k, v in init.items()})
    # config 5's light model: Robot2D, InitParameters(), mode_v with v_ref = 0.3, dt = 0.01, Δt = 0.02, 10 s (+ 3 other commands)
    import ctypes as C
    D = C.POINTER(C.c_double)
    vp = np.array([0.15, 0.05, 1.0, 0.1, -1.0, -1.0, 0.32, 0.0189, 0.0014])
    import hdf5_min
    d = hdf5_min.read_all(os.path.join(os.path.dirname(os.path.dirname(HERE)), "flight.jl_amd", "data", "robot2d.h5"))
    gp = np.concatenate([d["K_fbk"].ravel(), d["K_fwd"].ravel(), d["K_int"].ravel(), d["x_trim"].ravel(), d["u_trim"].ravel(), d["z_trim"].ravel(),
                         [0.6, 0.0, 0.0, 0.01]]).astype(np.float64)
    nr = 4
    ip = np.zeros((3, nr)); ip[0] = [0.0, 0.05, 0.0, -0.05]
    r = np.zeros((10, nr))
    o.lib.fo_robot2d_init(C.c_int64(nr), vp.ctypes.data_as(D), ip.ctypes.data_as(D), r.ctypes.data_as(D))
    ur = np.zeros((4, nr)); ur[0] = [1, 1, 2, 0]; ur[1, 3] = 0.1; ur[2] = [0.3, -0.2, 0.0, 0.0]; ur[3, 2] = 1.0
    r0 = r.copy(); samples = [r0.copy()]
    strr = np.zeros(nr, np.int32)
    for k in range(10):
        o.lib.fo_robot2d_step(C.c_int64(nr), vp.ctypes.data_as(D), gp.ctypes.data_as(D), C.c_double(0.01), 2, 1, ur.ctypes.data_as(D), r.ctypes.data_as(D),
                              C.c_int64(100 * k), C.c_int64(100), strr.ctypes.data_as(C.POINTER(C.c_int32)))
        samples.append(r.copy())
    np.savez_compressed(os.path.join(HERE, "robot2d_modes4.npz"), vehicle=vp, gains=gp, init=ip, u=ur, dt=0.01, ratio=2, save_every=100,
                        traj=np.stack(samples), status=strr)
    print("written:", [f for f in os.listdir(HERE) if f.endswith(".npz")])


if __name__ == "__main__":
    main()
