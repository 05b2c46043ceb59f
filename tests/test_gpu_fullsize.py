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


def test_config3_per_gpu_share_xv2_autopilot(fb, oracle):
    """BASELINE.json configs[3] at its per-GPU size: N = 524 288 Cessna172Xv2, dt = 0.01, autopilot every 2 steps (Δt = 0.02), the
    README example 2 scenario (wind N 1 / E 0.5 m/s, EAS + climb-rate mode with clm_ref = 2 m/s, bank + sideslip mode with φ_ref = 30°)
    flown from bench.py's randomised-trim lattice for 10 s: size-independent invariants on ALL aircraft, a 512-aircraft stratified
    sample against the CPU oracle at 1e-6, and big batch == small batch bit for bit."""
    import bench
    from oracle_binding import OracleX
    from test_gpu_c172x import ref_to_dev_rows, x_scale
    K = fb.K
    n = bench.N_TOTAL // 2
    EAS, h, psi, cell = bench.lattice(3, n)
    gains = fb.ctl_gains.ctl_gains_blob()
    wind = (1.0, 0.5, 0.0)
    w = fb.Cessna172Xv2World(n, gains=gains)
    w.set_params(wind_ned=wind)
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    tp = fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi)
    fb.init(sim, tp)
    assert w.trim_success.all(), f"{(~w.trim_success).sum()} aircraft failed to trim"
    w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
    x0, s0, u0, ui0, cu0, cs0 = w.x, w.s, w.u, w.ui, w.cu, w.cs
    for _ in range(20):
        fb.step(sim, 0.5)
    w.sync()
    x1, s1, st, cs1 = w.x, w.s, w.status, w.cs
    perm = ref_to_dev_rows(K)                     # C ABI row -> oracle / device row
    row = {k: int(np.where(perm == k)[0][0]) for k in (8, 12, 16, 20)}
    # ---- invariants on ALL 524 288 aircraft
    assert (st == 0).all(), f"{(st != 0).sum()} aircraft terminated"
    assert np.isfinite(x1).all() and np.isfinite(cs1).all()
    assert (x1[row[8]] < x0[row[8]]).all(), "fuel must strictly decrease on every aircraft"
    for r0 in (row[12], row[16]):
        q = x1[r0:r0 + 4]
        assert np.abs(np.sqrt((q * q).sum(0)) - 1.0).max() <= 1e-8 * (1 + 1e-6)
    assert (s1[1] == 2).all()
    assert (cs1[K["FB_CS_LON_MODE"]] == float(fb.ModeControlLon.EAS_clm)).all() and (cs1[K["FB_CS_LAT_MODE"]] == float(fb.ModeControlLat.φ_β)).all()
    climbed = x1[row[20]] - x0[row[20]]
    assert climbed.min() > 8.0 and climbed.max() < 25.0, (climbed.min(), climbed.max())     # 2 m/s demanded for 10 s, minus the capture transient
    # ---- the stratified sample (one aircraft out of every second (EAS, h) cell, drawn over the whole permuted order) against the oracle
    sel = bench.stratified_sample(cell, per_cell=1)[::2]
    assert sel.size == 512 and sel.max() > n - n // 32 and sel.min() < n // 32
    X = OracleX(oracle, gains)
    env = oracle.default_env(wind=wind)
    o = dict(x=np.empty((34, sel.size)), u=np.ascontiguousarray(u0[:, sel]), ui=np.ascontiguousarray(ui0[sel]), s=np.ascontiguousarray(s0[:, sel]),
             cu=np.ascontiguousarray(cu0[:, sel]), cs=np.ascontiguousarray(cs0[:, sel]), status=np.zeros(sel.size, np.int32), nstep=0)
    o["x"][perm] = x0[:, sel]
    X.step(o, env, 0.01, 2, 1000, threads=min(oracle.max_threads(), bench.usable_cores()))
    assert (o["status"] == 0).all() and np.array_equal(o["s"], s1[:, sel])
    err = np.abs(x1[:, sel] - o["x"][perm]) / x_scale(o["x"])[perm]
    cerr = np.abs(cs1[:, sel] - o["cs"]) / np.maximum(np.abs(o["cs"]), 1.0)
    print("configs[3] at N = 524 288 per GPU: max scaled error of 512 stratified aircraft after 1000 closed-loop steps: %.3e (control-law record %.3e)" % (err.max(), cerr.max()))
    assert err.max() < 1e-6 and cerr.max() < 1e-6
    # ---- results do not depend on batch size / lane position
    w2 = fb.Cessna172Xv2World(sel.size, gains=gains)
    w2.set_params(wind_ned=wind)
    sim2 = fb.Simulation(w2, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    w2.set_state(np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(s0[:, sel])); w2.u = np.ascontiguousarray(u0[:, sel]); w2.ui = np.ascontiguousarray(ui0[sel])
    w2.cu = np.ascontiguousarray(cu0[:, sel]); w2.cs = np.ascontiguousarray(cs0[:, sel])
    fb.step(sim2, 10.0); w2.sync()
    assert np.array_equal(w2.x, x1[:, sel]) and np.array_equal(w2.cs, cs1[:, sel])
    w2.close(); w.close()


def test_config3_whole_batch_on_one_gpu_4m_aircraft(fb):
    """BASELINE.json's LARGEST N on one device: Cessna172Xv2World(4 194 304) — configs[3] whole, what an 8-GPU node splits — 2 launches
    of 50 steps with the autopilot every 2 steps. The control-law record is 66 rows x 4 194 304 x 8 B = 2.2 GB, its launch-start copy
    another 2.2 GB, cu 0.94 GB: every row offset of the record paths passes 2^31 BYTES from row 8 on and 2^32 from row 16 on, so a 32-bit
    offset anywhere in the 66-row / 28-row / 20-row paths shows here before an 8-GPU node finds it. The batch is lattice(3)'s 524 288
    aircraft eight times over: invariants on ALL aircraft, every copy equal to the first bit for bit (incl. the last, whose rows lie
    highest), and a 512-aircraft stratified sample of the LAST copy bitwise equal to a 512-aircraft world stepped from the same state
    (results independent of batch size)."""
    import bench
    K = fb.K
    n1 = bench.N_TOTAL // 2
    copies = 8
    n = n1 * copies
    EAS, h, psi, cell = bench.lattice(3, n1)
    gains = fb.ctl_gains.ctl_gains_blob()
    wind = (1.0, 0.5, 0.0)
    w = fb.Cessna172Xv2World(n, gains=gains)
    w.set_params(wind_ned=wind)
    sim = fb.Simulation(w, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.init(sim, fb.TrimParameters(EAS=np.tile(EAS, copies), h_e=np.tile(h, copies), ψ_nb=np.tile(psi, copies)))
    assert w.trim_success.all()
    w.ctl.lon.mode_req = float(fb.ModeControlLon.EAS_clm); w.ctl.lon.clm_ref = 2.0
    w.ctl.lat.mode_req = float(fb.ModeControlLat.φ_β); w.ctl.lat.φ_ref = float(np.deg2rad(30.0))
    sel = bench.stratified_sample(cell, per_cell=1)[::2]
    assert sel.size == 512
    last = (copies - 1) * n1 + sel
    x0 = w.x
    assert np.array_equal(x0.reshape(-1, copies, n1), np.broadcast_to(x0[:, None, :n1], (x0.shape[0], copies, n1)))
    fuel_row = K["FB_X_FUEL"]           # rows 0-11 of the C ABI's Xv2 state are the Sv0 rows
    fuel0 = x0[fuel_row].copy()
    small = dict(x=np.ascontiguousarray(x0[:, last]), s=np.ascontiguousarray(w.s[:, last]), u=np.ascontiguousarray(w.u[:, last]),
                 ui=np.ascontiguousarray(w.ui[last]))
    cu0 = w.cu; small["cu"] = np.ascontiguousarray(cu0[:, last]); del cu0
    cs0 = w.cs; small["cs"] = np.ascontiguousarray(cs0[:, last]); del cs0
    del x0
    for _ in range(2):
        fb.step(sim, 0.5)
    w.sync()
    st = w.status
    assert (st == 0).all(), f"{(st != 0).sum()} aircraft terminated"
    x1 = w.x
    assert np.isfinite(x1).all()
    assert (x1[fuel_row] < fuel0).all(), "fuel must strictly decrease on every aircraft"
    for r0 in (K["FB_X2_KIN"], K["FB_X2_KIN"] + 4):
        q = x1[r0:r0 + 4]
        assert np.abs(np.sqrt((q * q).sum(0)) - 1.0).max() <= 1e-8 * (1 + 1e-6)
    assert np.array_equal(x1.reshape(-1, copies, n1), np.broadcast_to(x1[:, None, :n1], (x1.shape[0], copies, n1))), "a copy of the batch differs from the first"
    cs1 = w.cs
    assert np.isfinite(cs1).all()
    assert (cs1[K["FB_CS_LON_MODE"]] == float(fb.ModeControlLon.EAS_clm)).all() and (cs1[K["FB_CS_LAT_MODE"]] == float(fb.ModeControlLat.φ_β)).all()
    assert np.array_equal(cs1.reshape(-1, copies, n1), np.broadcast_to(cs1[:, None, :n1], (cs1.shape[0], copies, n1)))
    x1s, cs1s = np.ascontiguousarray(x1[:, last]), np.ascontiguousarray(cs1[:, last])
    del x1, cs1
    w.close()
    w2 = fb.Cessna172Xv2World(sel.size, gains=gains)
    w2.set_params(wind_ned=wind)
    sim2 = fb.Simulation(w2, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    w2.set_state(small["x"], small["s"]); w2.u = small["u"]; w2.ui = small["ui"]; w2.cu = small["cu"]; w2.cs = small["cs"]
    fb.step(sim2, 1.0); w2.sync()
    assert np.array_equal(w2.x, x1s) and np.array_equal(w2.cs, cs1s)
    w2.close()


@pytest.mark.parametrize("kin", ["ECEF", "NED"])
def test_config2_full_size_other_mechanisations(fb, oracle, kin):
    """configs[2]'s batch (N = 1 048 576 on bench.py's lattice) in the ECEF and NED mechanisations, stepped by k_step_duo<KIN>: invariants on
    ALL aircraft after 4 launches of 50 steps, a stratified sample of 512 against the oracle in the same mechanisation, and big batch ==
    small batch bit for bit."""
    import bench
    K = fb.K
    n = bench.N_TOTAL
    nk = {"ECEF": 8, "NED": 6}[kin]
    EAS, h, psi, cell = bench.lattice(0)
    w = fb.BatchedWorld(n, kinematics=kin)
    fb.f_init(w, fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi))
    assert w.trim_success.all()
    x0, s0, u0, ui0 = w.x, w.s, w.u, w.ui
    assert x0.shape[0] == 18 + nk
    sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
    for _ in range(4):
        fb.step(sim, 0.5)
    w.sync()
    x1, s1, st = w.x, w.s, w.status
    assert (st == 0).all() and np.isfinite(x1).all()
    assert (x1[8] < x0[8]).all(), "fuel must strictly decrease on every aircraft"
    assert (x1[2:8] == 0).all() and (s1[1] == 2).all()
    if kin == "ECEF":
        for q in (x1[12:16], x1[16:19]):      # q_eb and n_e: renormalised by f_step! beyond 1e-8 (kinematics.jl:317-320)
            assert np.abs(np.sqrt((q * q).sum(0)) - 1.0).max() <= 1e-8 * (1 + 1e-6)
    sel = bench.stratified_sample(cell, per_cell=1)[::2]
    assert sel.size == 512
    xs = np.ascontiguousarray(x0[:, sel]); ss = np.ascontiguousarray(s0[:, sel]); us = np.ascontiguousarray(u0[:, sel]); uis = np.ascontiguousarray(ui0[sel])
    oracle.lib.fo_set_kinematics(K["FB_KIN_" + kin])
    try:
        xo27 = np.zeros((27, sel.size)); xo27[:12 + nk] = xs[:12 + nk]; xo27[21:] = xs[12 + nk:]
        xo, so, sto = oracle.step(xo27, us, uis, ss, oracle.default_env(), 0.01, 200, threads=min(oracle.max_threads(), bench.usable_cores()))
    finally:
        oracle.lib.fo_set_kinematics(K["FB_KIN_WA"])
    assert (sto == 0).all() and np.array_equal(so, s1[:, sel])
    xo_abi = np.vstack([xo[:12 + nk], xo[21:]])
    err = np.abs(x1[:, sel] - xo_abi) / np.maximum(np.abs(xo_abi), 1.0)
    print("configs[2] in %s at N = 1 048 576: max error of 512 stratified aircraft after 200 steps: %.3e" % (kin, err.max()))
    assert err.max() < 1e-6
    w2 = fb.BatchedWorld(sel.size, kinematics=kin)
    w2.set_state(xs, ss); w2.u = us; w2.ui = uis
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim2, 2.0); w2.sync()
    assert np.array_equal(w2.x, x1[:, sel])
    w2.close(); w.close()


def test_config4_mixed_fp32_fleet_as_stated(fb, oracle):
    """BASELINE.json configs[4] as stated, on one GPU: a MixedFleet of 524 288 fp32 Cessna172Sv0 (bench.py's lattice(1): the reference's
    Cessna172Sv0 at randomised C172.TrimParameters, FA/c172/c172s/c172s0.jl:14-18) and 524 288 fp32 Robot2D (FA/robot2d/robot2d.jl:526-570),
    interleaved in the caller's order, dt = 0.01, Δt = 0.02, 10 s. Invariants on ALL 1 048 576 vehicles; packed == homogeneous bit for
    bit; 1024 aircraft and 1024 robots against the fp64 oracle under the STATED fp32 tolerances (bench.F32_TOLERANCE — the numbers
    bench.py's extra.fleet leg reports as its tolerance)."""
    import ctypes as C
    import bench
    from test_oracle_robot2d import DEFAULT_VP, gains_from_h5, run
    from test_gpu_robot2d import oracle_init
    tol = bench.F32_TOLERANCE
    n = bench.N_TOTAL
    KC, KR = fb.K["FB_MODEL_C172S0"], fb.K["FB_MODEL_ROBOT2D"]
    types = np.where(np.arange(n) % 2 == 0, KC, KR)
    fleet = fb.MixedFleet(types, {KC: lambda m: fb.BatchedWorld(m, dtype="f32"), KR: lambda m: fb.Robot2DWorld(m, dtype="f32")})
    fleet.simulate(dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    nc, nr = fleet.index[KC].size, fleet.index[KR].size
    assert nc == n // 2 and nr == n // 2
    EAS, h, psi, cell = bench.lattice(1)
    EAS, h, psi, cell = EAS[:nc], h[:nc], psi[:nc], cell[:nc]
    rng = np.random.default_rng(404)
    ipar = fb.InitParameters(u_m=rng.uniform(-0.1, 0.1, nr), ω=rng.uniform(-0.05, 0.05, nr), η=rng.uniform(-1, 1, nr))
    v_ref = rng.uniform(-0.3, 0.3, nr)
    fleet.init({KC: fb.TrimParameters(EAS=EAS, h_e=h, ψ_nb=psi), KR: ipar})
    cw, rw = fleet.worlds[KC], fleet.worlds[KR]
    assert cw.trim_success.all()
    ur = np.zeros((4, nr)); ur[0] = 1; ur[2] = v_ref; rw.u = ur                      # mode_v, a velocity reference per robot
    x0, s0, u0, ui0 = cw.x, cw.s, cw.u, cw.ui
    r0 = rw.x
    for _ in range(20):
        fleet.step(0.5)
    fleet.sync()
    x = fleet.gather("x")
    st = fleet.gather("status", fill=-1)
    is_r = types == KR
    # ---- invariants on ALL vehicles
    assert x.shape == (27, n) and (st == 0).all(), f"{int((st != 0).sum())} vehicles terminated"
    xc, xr = x[:, ~is_r], x[:10, is_r]
    assert np.isfinite(xc).all() and np.isfinite(xr).all() and np.isnan(x[10:, is_r]).all()
    assert (xc[8] < x0[8]).all(), "fuel must strictly decrease on every aircraft"
    assert (xc[2:8] == 0).all() and (cw.s[1] == 2).all()
    assert np.abs(np.sqrt((xc[12:16] ** 2).sum(0)) - 1.0).max() < 1e-6          # fp32 attitude quaternion
    assert np.abs(np.sqrt((xc[16:20] ** 2).sum(0)) - 1.0).max() <= 1e-8 * (1 + 1e-6)   # the position quaternion is integrated in fp64
    assert np.abs(xc[20] - x0[20]).max() < 30.0
    assert np.abs(xr[1] - v_ref).max() < 0.05, "every robot within 0.05 m/s of its velocity reference after 10 s"
    assert np.array_equal(xc, cw.x) and np.array_equal(xr, rw.x)
    # ---- packed == homogeneous, bit for bit (a sample of each model stepped in worlds of its own: results do not depend on the
    # batch size, the lane position or what else runs on the GPU)
    sel = bench.stratified_sample(cell, per_cell=1)
    assert sel.size == 1024
    w2 = fb.BatchedWorld(sel.size, dtype="f32")
    xs, ss, us, uis = (np.ascontiguousarray(x0[:, sel]), np.ascontiguousarray(s0[:, sel]), np.ascontiguousarray(u0[:, sel]), np.ascontiguousarray(ui0[sel]))
    w2.set_state(xs, ss); w2.u = us; w2.ui = uis
    sim2 = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
    fb.step(sim2, 10.0); w2.sync()
    assert np.array_equal(w2.x, xc[:, sel]), "fp32 aircraft: packed fleet != homogeneous batch"
    selr = np.sort(rng.choice(nr, 1024, replace=False))
    r2 = fb.Robot2DWorld(selr.size, dtype="f32")
    r2.set_state(np.ascontiguousarray(r0[:, selr])); r2.u = np.ascontiguousarray(ur[:, selr])
    simr = fb.Simulation(r2, dt=0.01, Δt=0.02, save_on=False, steps_per_launch=50)
    fb.step(simr, 10.0); r2.sync()
    assert np.array_equal(r2.x, xr[:, selr]), "fp32 robots: packed fleet != homogeneous batch"
    w2.close(); r2.close()
    # ---- the samples against the fp64 oracle, stated fp32 tolerances
    xo, so, sto = oracle.step(xs, us, uis, ss, oracle.default_env(), 0.01, 1000, threads=min(oracle.max_threads(), bench.usable_cores()))
    assert (sto == 0).all() and np.array_equal(so, cw.s[:, sel])
    e = bench.f32_abs_errors(xc[:, sel], xo, np.ones(sel.size, bool))
    print("configs[4], 1024 fp32 aircraft of the fleet vs the fp64 oracle after 1000 steps:", {k: "%.2e" % v for k, v in e.items()})
    for k, v in e.items():
        assert v < tol[k], (k, v, tol[k])
    vp = DEFAULT_VP.copy(); gp = gains_from_h5()
    ro = oracle_init(oracle.lib, vp, np.ascontiguousarray(ipar.pack(nr)[:, selr]))
    assert np.abs(ro - r0[:, selr]).max() < 1e-6                                   # (the oracle's f_init! and the fp32 device's agree)
    sto_r = run(oracle.lib, vp, gp, ro, np.ascontiguousarray(ur[:, selr]), 0.01, 2, 1, 0, 1000)
    assert (sto_r == 0).all()
    er = np.max(np.abs(xr[:, selr] - ro) / np.maximum(np.abs(ro), 1.0))
    print("configs[4], 1024 fp32 robots of the fleet vs the fp64 oracle after 1000 steps: max scaled error %.2e" % er)
    assert er < tol["robot2d_scaled"]
    fleet.close()
