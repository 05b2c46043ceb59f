"""The fp32 airborne stepper of Cessna172Sv0 (BASELINE.json configs[4] asks for an fp32 fleet): accuracy against the fp64 oracle,
and the hand-over of near-ground lanes to the fp64 kernel."""
import numpy as np
import pytest

import os
import sys

from test_gpu_parity import lattice_trim_params, state_scale

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import F32_TOLERANCE   # noqa: E402

pytestmark = pytest.mark.gpu


def test_f32_trajectory_error_against_fp64_oracle(fb, oracle):
    """10 s of perturbed flight with fp32 arithmetic (positions integrated in fp64) against the fp64 oracle, in physical units:
    body rates within 1e-5 rad/s, velocities within 2e-3 m/s, altitude within 5 cm, attitude quaternion within 2e-5 — three
    orders above the fp64 path's contract, and documented as such; discrete states and status agree."""
    n = 4096
    tp = lattice_trim_params(fb, n, seed=41)
    w = fb.BatchedWorld(n, dtype="f32")
    fb.f_init(w, tp)
    rng = np.random.default_rng(4)
    x = w.x
    x[21:24] += rng.normal(0, 0.02, (3, n)); x[24:27] += rng.normal(0, 1.0, (3, n))
    w.set_state(x, w.s)
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim, 10.0); w.sync()
    xo, so, sto = oracle.step(x0, u0, ui0, s0, oracle.default_env(), 0.01, 1000)
    ok = w.trim_success & (sto == 0)
    assert (w.status[ok] == 0).all() and np.array_equal(w.s[:, ok], so[:, ok])
    err = np.abs(w.x - xo) / state_scale(xo)
    worst = err[:, ok].max(axis=1)
    print("fp32 vs fp64 oracle after 1000 steps: max scaled error %.2e (row %d); position rows %.2e; median over aircraft %.2e"
          % (worst.max(), worst.argmax(), err[16:21][:, ok].max(), np.median(err[:, ok].max(axis=0))))
    d = np.abs(w.x - xo)[:, ok]
    print("absolute: rates %.1e rad/s, velocity %.1e m/s, altitude %.1e m, q_wb %.1e, q_ew %.1e, engine speed %.1e rad/s"
          % (d[21:24].max(), d[24:27].max(), d[20].max(), d[12:16].max(), d[16:20].max(), d[9].max()))
    tol = F32_TOLERANCE   # the stated fp32 bounds (bench.py: the same numbers go into extra.fleet.rel_err_vs_cpu.tolerance)
    assert tol["rates_rad_s"] == 1e-5 and tol["velocity_m_s"] == 2e-3 and tol["altitude_m"] == 0.05 and tol["q_wb"] == 2e-5 and tol["engine_speed_rad_s"] == 0.05
    assert d[21:24].max() < tol["rates_rad_s"] and d[24:27].max() < tol["velocity_m_s"] and d[20].max() < tol["altitude_m"]
    assert d[12:16].max() < tol["q_wb"] and d[9].max() < tol["engine_speed_rad_s"]
    # the aircraft did move over the Earth, and its position is as good as the velocity allows (the reason the position rows
    # are integrated in fp64: in fp32 the per-step increment of q_ew is below one ulp)
    moved = np.abs(w.x[16:20] - x0[16:20]).max(axis=0)
    assert (moved[ok] > 1e-6).all() and d[16:20].max() < tol["q_ew"]
    w.close()


def test_f32_hands_ground_contact_to_the_fp64_kernel(fb, oracle):
    """Approaches through the 10 m limit with an fp32 handle: from the hand-over on, the lane is stepped by the fp64 kernel, so
    touchdown happens (fp32 could not resolve a wheel height) and the status words match the oracle's."""
    n = 512
    rng = np.random.default_rng(18)
    h_trn = 300.0
    tp = fb.TrimParameters(EAS=rng.uniform(33, 40, n), h_e=h_trn + rng.uniform(14, 40, n), γ_wb_n=-np.deg2rad(rng.uniform(2, 5, n)),
                           flaps=1.0, ψ_nb=rng.uniform(-3, 3, n))
    w = fb.BatchedWorld(n, dtype="f32")
    w.set_params(h_terrain=h_trn)
    fb.f_init(w, tp)
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    ok = w.trim_success
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=25)
    fb.step(sim, 6.0); w.sync()
    env = oracle.default_env(h_trn=h_trn)
    xo, so, sto = oracle.step(x0, u0, ui0, s0, env, 0.01, 600)
    fb.f_ode(w)
    y = w.y
    wow = (y[fb.K["FB_Y_LDG"] + 1] + y[fb.K["FB_Y_LDG"] + 12] + y[fb.K["FB_Y_LDG"] + 23]) > 0
    _, yo, _ = oracle.f_ode(xo, u0, ui0, so, env)
    wow_o = (yo[fb.K["FB_Y_LDG"] + 1] + yo[fb.K["FB_Y_LDG"] + 12] + yo[fb.K["FB_Y_LDG"] + 23]) > 0
    print("on wheels: gpu %d, oracle %d; terminated gpu %d oracle %d" % (wow[ok].sum(), wow_o[ok].sum(), (w.status[ok] != 0).sum(), (sto[ok] != 0).sum()))
    assert wow[ok].sum() > 5 and abs(int(wow[ok].sum()) - int(wow_o[ok].sum())) <= max(3, wow_o[ok].sum() // 10)
    live = ok & (sto == 0) & (w.status == 0)
    assert live.sum() > 0.9 * ok.sum()
    h_err = np.abs(w.x[20] - xo[20])[live]
    assert np.median(h_err) < 0.05 and h_err.max() < 1.0     # metres, after a 6 s approach in fp32 + fp64 roll-out
    w.close()
