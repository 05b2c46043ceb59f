"""BASELINE.json's configurations at their full sizes on one MI355X (through the C ABI):
configs[2] — N = 1 048 576 Cessna172Sv0 on bench.py's randomised-trim lattice: a stratified sample against the CPU oracle at the
north star's 1e-6, and size-independent invariants on ALL aircraft;
configs[1] — N = 65 536 copies of C172.TrimParameters(): every lane must stay bit-identical to lane 0."""
import os
import sys
import numpy as np
import pytest

from test_gpu_parity import state_scale

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_config2_full_size_lattice(fb, oracle):
    import bench
    n = bench.N_TOTAL
    EAS, h, psi, cell = bench.lattice(0)
    assert n == 1 << 20 and np.unique(cell).size == 1024
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    assert w.trim_success.all(), f"{(~w.trim_success).sum()} aircraft of the bench lattice failed to trim"   # expected fraction: exactly 1
    assert w.trim_cost.max() < 1e-20
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    for _ in range(20):                       # 20 launches of 50 steps = 10 s of flight, like bench.py's timed region
        fb.step(sim, 0.5)
    w.sync()
    x1, s1, st = w.x, w.s, w.status
    # ---- invariants on ALL 1 048 576 aircraft
    assert (st == 0).all(), f"{(st != 0).sum()} aircraft terminated"
    assert np.isfinite(x1).all()
    assert (x1[8] < x0[8]).all(), "fuel must strictly decrease on every aircraft (nobody frozen, no redo flag left set)"
    for q in (x1[12:16], x1[16:20]):         # f_step! renormalises when | |q| - 1 | > 1e-8 (kinematics.jl:114-118, 226-229)
        assert np.abs(np.sqrt((q * q).sum(0)) - 1.0).max() <= 1e-8 * (1 + 1e-6)
    assert (x1[2:8] == 0).all() and (x1[10:12] == 0).all()      # contact regulators and saturated engine compensators stay exactly 0 airborne
    assert (s1[1] == 2).all()                                    # engines running
    assert np.abs(x1[20] - x0[20]).max() < 30.0                  # trimmed level flight: altitude holds within the phugoid
    # a second world stepping only the sample must reproduce the big batch bit for bit (results do not depend on batch size / lane position)
    sel = bench.stratified_sample(cell)
    assert sel.size == 4096 and np.unique(cell[sel]).size == 1024 and sel.max() > n - n // 64 and sel.min() < n // 64
    xs, ss, us, uis = (np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(s0[:, sel]), np.ascontiguousarray(u0[:, sel]), np.ascontiguousarray(ui0[sel]))
    # ---- the stratified sample against the CPU oracle
    xo, so, sto = oracle.step(xs, us, uis, ss, oracle.default_env(), 0.01, 1000, threads=min(oracle.max_threads(), bench.usable_cores()))
    assert (sto == 0).all() and np.array_equal(so, s1[:, sel])
    err = np.abs(x1[:, sel] - xo) / state_scale(xo)
    print("configs[2] at N = 1 048 576: max scaled error of 4096 stratified aircraft after 1000 steps: %.3e" % err.max())
    assert err.max() < 1e-6
    w2 = fb.BatchedWorld(sel.size)
    w2.set_state(xs, ss); w2.u = us; w2.ui = uis
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim2, 10.0); w2.sync()
    assert np.array_equal(w2.x, x1[:, sel])
    w2.close(); w.close()


def test_config1_identical_trim_stays_identical(fb, oracle):
    """N = 65 536 copies of Cessna172Sv0 at C172.TrimParameters() (BASELINE.json configs[1]): trim, 10 s of stepping; every lane
    equals lane 0 bit for bit at every check point, and lane 0 equals the oracle's config-1 run."""
    n = 65536
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters())
    assert w.trim_success.all()
    x0 = w.x
    assert (x0 == x0[:, :1]).all() and (w.trim_state == w.trim_state[:, :1]).all()
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    for _ in range(4):
        fb.step(sim, 2.5); w.sync()
        x = w.x
        assert (x == x[:, :1]).all() and (w.status == 0).all() and (w.s == w.s[:, :1]).all()
    xo, so, sto = oracle.step(np.ascontiguousarray(x0[:, :1]), np.ascontiguousarray(w.u[:, :1]), np.ascontiguousarray(w.ui[:1]),
                              np.array([[0], [2]], np.int32), oracle.default_env(), 0.01, 1000)
    err = np.abs(x[:, :1] - xo) / state_scale(xo)
    assert err.max() < 1e-6, err.max()
    w.close()
