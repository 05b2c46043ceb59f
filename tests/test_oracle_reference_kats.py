"""Pins the CPU oracle against every known-answer / tolerance test the reference holds for the hot path.

The reference ships no golden vectors or recorded trajectories — its own tests are
algebraic identities, a few scalar known answers and closed-loop tolerances. Each test below restates one of
them (cited as reference file:line, relative to lib/) and asserts it on the oracle with the reference's own
tolerance (Julia's default `≈` is rtol = sqrt(eps) = 1.49e-8).
"""
import ctypes as C
import numpy as np
import pytest

RTOL = 1.4901161193847656e-08  # sqrt(eps(Float64)): Base.isapprox default
_D = C.POINTER(C.c_double)


def dp(a):
    return a.ctypes.data_as(_D)


def arr(*v):
    return np.array(v, dtype=np.float64)


def isapprox(a, b, rtol=RTOL, atol=0.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) <= max(atol, rtol * max(np.linalg.norm(a), np.linalg.norm(b)))


@pytest.fixture(scope="module")
def L(oracle):
    return oracle.lib


def quat_from_euler(L, psi, theta, phi):
    q = np.zeros(4); L.fo_quat_from_euler(C.c_double(psi), C.c_double(theta), C.c_double(phi), dp(q)); return q


def qmul(L, a, b):
    o = np.zeros(4); L.fo_quat_mul(dp(np.ascontiguousarray(a)), dp(np.ascontiguousarray(b)), dp(o)); return o


def qrot(L, q, v):
    o = np.zeros(3); L.fo_quat_rotate(dp(np.ascontiguousarray(q)), dp(np.ascontiguousarray(v, dtype=np.float64)), dp(o)); return o


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


# ---------------------------------------------------------------------------------------------------------
def test_quaternion_algebra(L):
    """FlightPhysics/test/test_quaternions.jl:60-105 — product associativity, inverse, norm."""
    rng = np.random.default_rng(0)
    a, b, c = (rng.normal(size=4) for _ in range(3))
    assert isapprox(qmul(L, qmul(L, a, b), c), qmul(L, a, qmul(L, b, c)))
    u = a / np.linalg.norm(a)
    assert isapprox(qmul(L, u, qconj(u)), [1, 0, 0, 0])
    assert isapprox(np.linalg.norm(qmul(L, a, b)), np.linalg.norm(a) * np.linalg.norm(b))


def test_attitude_roundtrips_and_euler_rate(L):
    """FlightPhysics/test/test_attitude.jl:54-78,182-213,217-233 — representation round trips;
    dt(REuler(φ = π/2), [0,1,0]) ≈ [1,0,0]."""
    rng = np.random.default_rng(1)
    for _ in range(50):
        psi, theta, phi = rng.uniform(-np.pi, np.pi), rng.uniform(-1.5, 1.5), rng.uniform(-np.pi, np.pi)
        q = quat_from_euler(L, psi, theta, phi)
        e = np.zeros(3); L.fo_euler_from_quat(dp(q), dp(e))
        assert isapprox(e, [psi, theta, phi])
        M = np.zeros(9); L.fo_rmatrix_from_quat(dp(q), dp(M))
        q2 = np.zeros(4); L.fo_quat_from_rmatrix(dp(M), dp(q2))
        assert isapprox(q, q2) or isapprox(q, -q2)
        v = rng.normal(size=3)
        assert isapprox(qrot(L, q, v), M.reshape(3, 3) @ v)
        assert isapprox(qrot(L, qconj(q), qrot(L, q, v)), v)
    ed = np.zeros(3); L.fo_euler_dot(dp(arr(0, 0, np.pi / 2)), dp(arr(0, 1, 0)), dp(ed))
    assert isapprox(ed, [1, 0, 0])


def test_geodesy_latlon_ltf_torture(L):
    """FlightPhysics/test/test_geodesy.jl:74-92 — get_ψ_nw(ltf(ll) ∘ Rz(ψ)) ≈ ψ; LatLon → NVector → ltf →
    ∘Rz(ψ_nw) → NVector → LatLon over a 10x10x10 grid."""
    n = np.zeros(3); q = np.zeros(4); n2 = np.zeros(3); ll = np.zeros(2)
    L.fo_nvector_from_latlon(C.c_double(np.pi / 3), C.c_double(-np.pi / 6), dp(n))
    L.fo_ltf(dp(n), C.c_double(0.0), dp(q))
    r_ew = qmul(L, q, arr(np.cos(np.pi / 6), 0, 0, np.sin(np.pi / 6)))
    assert abs(L.fo_psi_nw_from_qew(dp(r_ew)) - np.pi / 3) < RTOL
    for phi in np.linspace(-np.pi / 2, np.pi / 2, 10):
        for lam in np.linspace(-np.pi, np.pi, 10):
            for psi in np.linspace(-np.pi, np.pi, 10):
                L.fo_nvector_from_latlon(C.c_double(phi), C.c_double(lam), dp(n))
                L.fo_ltf(dp(n), C.c_double(0.0), dp(q))
                r = qmul(L, q, arr(np.cos(psi / 2), 0, 0, np.sin(psi / 2)))
                L.fo_nvector_from_qew(dp(r), dp(n2))
                assert isapprox(n, n2)  # LatLon ≈ is defined through NVector (geodesy.jl:109)


def test_geodesy_cartesian_torture_and_altitudes(L):
    """FlightPhysics/test/test_geodesy.jl:108,180-192 — Cartesian ⇄ Geographic across datums; altitude datum round trip."""
    r = np.zeros(3); g = np.zeros(4); r2 = np.zeros(3); n = np.zeros(3)
    for phi in np.linspace(-np.pi / 2, np.pi / 2, 10):
        for lam in np.linspace(-np.pi, np.pi, 10):
            for h_o in np.linspace(-800.0, 10000.0, 10):
                L.fo_nvector_from_latlon(C.c_double(phi), C.c_double(lam), dp(n))
                N = L.fo_geoid_height(dp(n))
                h_e = h_o + N
                L.fo_cartesian_from_geographic(dp(n), C.c_double(h_e), dp(r))
                L.fo_geographic_from_cartesian(dp(r), dp(g))
                # back through Orthometric and Geopotential
                h_o2 = g[3] - L.fo_geoid_height(dp(g[:3].copy()))
                h_g = L.fo_h_geop_from_orth(C.c_double(h_o2))
                h_e3 = L.fo_h_orth_from_geop(C.c_double(h_g)) + L.fo_geoid_height(dp(g[:3].copy()))
                L.fo_cartesian_from_geographic(dp(g[:3].copy()), C.c_double(h_e3), dp(r2))
                assert isapprox(r, r2)
    loc = arr(3, 1, -5) / np.linalg.norm(arr(3, 1, -5))
    N = L.fo_geoid_height(dp(loc))
    assert (1500.0 + N) - N == 1500.0 or abs((1500.0 + N) - N - 1500.0) < 1e-12   # test_geodesy.jl:108
    # EGM96 sanity: the geoid undulation is within [-107, 86] m everywhere
    assert -110 < N < 90


def test_kinematics_three_mechanizations_agree(L):
    """FlightPhysics/test/test_kinematics.jl:35-95 — from LatLon(π/3, -π/6), HOrth(12354), ω = [.1,.1,-.2],
    v = [100,10,-4]: WA ≈ ECEF ≈ NED for q_nb, v_eb_n, h_e at init and after 20 s of RK4 at the default
    dt = 0.02 (Simulation default, FlightCore/src/sim.jl:188)."""
    n = np.zeros(3); L.fo_nvector_from_latlon(C.c_double(np.pi / 3), C.c_double(-np.pi / 6), dp(n))
    h_e = 12354.0 + L.fo_geoid_height(dp(n))
    init = np.concatenate([[1, 0, 0, 0], n, [h_e], [0.1, 0.1, -0.2], [100, 10, -4]]).astype(np.float64)
    outs = []
    for mech in range(3):
        o0 = np.zeros(17); L.fo_kinematics_sim(mech, dp(init), C.c_double(0.02), C.c_int64(0), dp(o0))
        o1 = np.zeros(17); L.fo_kinematics_sim(mech, dp(init), C.c_double(0.02), C.c_int64(1000), dp(o1))
        outs.append((o0, o1))
    for k in (0, 1):
        wa, ecef, ned = outs[0][k], outs[1][k], outs[2][k]
        for other in (ecef, ned):
            assert isapprox(wa[0:4], other[0:4]) or isapprox(wa[0:4], -other[0:4])   # q_nb (double cover)
            assert isapprox(wa[4:7], other[4:7])    # v_eb_n
            assert isapprox(wa[7], other[7])        # h_e
            assert isapprox(wa[8:11], other[8:11])  # n_e
    # KinData(kin_init) equivalence (:64-70)
    kd = np.zeros(18); L.fo_kindata_from_init(dp(init), dp(kd))
    assert isapprox(kd[15:18], outs[0][0][11:14]) and isapprox(kd[12:15], outs[0][0][14:17])


def test_dynamics_unit_mass_cases(L):
    """FlightPhysics/test/test_dynamics.jl:24-63 — unit mass/inertia: ω̇ ≈ 0, v̇ ≈ F + q_nb'(g_n), a_eb_b ≈ v̇,
    a_ib_b ≈ F + q_nb'(G_n); CoM 1 m ahead with F = [0,0,1]: ω̇_y ≈ 1."""
    n = arr(1, 0, 0)
    h_e = 1000.0 + L.fo_geoid_height(dp(n))
    q_nb = quat_from_euler(L, 1.0, 0.5, 0.4)
    init = np.concatenate([q_nb, n, [h_e], [1.0, 1.0, 0.0], [100, 0, 0]]).astype(np.float64)
    kd = np.zeros(18); L.fo_kindata_from_init(dp(init), dp(kd))
    q_eb, r_eb_e = kd[0:4].copy(), kd[4:7].copy()
    mp = np.concatenate([[1.0], [0, 0, 0], np.eye(3).ravel()])
    x6 = np.zeros(6); out = np.zeros(12)
    L.fo_dynamics_f_ode(dp(x6), dp(mp), dp(arr(1, 2, 1, 0, 0, 0)), dp(np.zeros(3)), dp(q_eb), dp(r_eb_e), dp(out))
    g4 = np.zeros(4); L.fo_geographic_from_cartesian(dp(r_eb_e), dp(g4))
    g = L.fo_gravity(dp(g4[:3].copy()), C.c_double(g4[3]))
    g_b = qrot(L, qconj(q_nb), arr(0, 0, g))
    assert np.allclose(out[0:3], 0, atol=1e-12)
    assert isapprox(out[3:6], arr(1, 2, 1) + g_b)
    assert isapprox(out[6:9], out[3:6])
    Gn = np.zeros(3); L.fo_G_n(dp(g4[:3].copy()), C.c_double(g4[3]), dp(Gn))
    assert isapprox(out[9:12], arr(1, 2, 1) + qrot(L, qconj(q_nb), Gn))
    mp2 = np.zeros(13); L.fo_mp_translate(dp(arr(1, 0, 0)), C.c_double(1.0), dp(np.eye(3).ravel()), dp(mp2))
    L.fo_dynamics_f_ode(dp(x6), dp(mp2), dp(arr(0, 0, 1, 0, 0, 0)), dp(np.zeros(3)), dp(q_eb), dp(r_eb_e), dp(out))
    assert isapprox(out[1], 1.0)


def test_pivector_continuous(L):
    """FlightPhysics/test/test_control.jl:38-66 — PIVector{2}, k_p = k_i = 1, channel 1 bounded to ±1, input -1,
    2 s at dt = 0.02: ch1 saturated low, integrator halted, |y_i| < 0.1, output ≈ -1; ch2 y_i ≈ -2, output ≈ -3
    (atol 1e-2); external saturation of the same sign halts, of opposite sign does not."""
    inf = np.inf
    prm = arr(1, 1, 0, 1, -1, 1, 1, 1, 0, 1, -inf, inf)
    x = np.zeros(2); out = np.zeros(8)
    L.fo_pi_sim(dp(prm), dp(arr(-1, -1)), np.zeros(2, np.int32).ctypes.data_as(C.POINTER(C.c_int32)), dp(x), C.c_double(0.02), C.c_int64(100), dp(out))
    assert out[2] == -1 and out[3] == 1 and abs(out[1]) < 0.1 and isapprox(out[0], -1.0)
    assert out[6] == 0 and out[7] == 0 and abs(out[5] + 2.0) < 1e-2 and abs(out[4] + 3.0) < 1e-2
    se = np.array([0, 1], np.int32)   # opposite sign of u_i = -1  -> not halted
    L.fo_pi_sim(dp(prm), dp(arr(-1, -1)), se.ctypes.data_as(C.POINTER(C.c_int32)), dp(x), C.c_double(0.02), C.c_int64(50), dp(out))
    assert out[7] == 0
    se = np.array([0, -1], np.int32)  # same sign -> halted
    L.fo_pi_sim(dp(prm), dp(arr(-1, -1)), se.ctypes.data_as(C.POINTER(C.c_int32)), dp(x), C.c_double(0.02), C.c_int64(50), dp(out))
    assert out[7] == 1


def test_propeller_coefficients_and_lookup(L):
    """FlightPhysics/test/test_propellers.jl:52-75 (static vs moving coefficient signs/monotonicity),
    :88 (flat extrapolation), :127-145 (CCW fixed-pitch wrench signs)."""
    st = np.zeros(6); mv = np.zeros(6)
    L.fo_prop_coefficients(2, C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), dp(st))
    L.fo_prop_coefficients(2, C.c_double(0.5), C.c_double(0.0), C.c_double(0.0), dp(mv))
    C_Fx, C_Mx, C_Fz, C_Mz, C_P, eta = st
    assert eta == 0 and C_Fx > 0 and C_Mx < 0 and C_Fz == 0 and C_Mz == 0 and C_P < 0
    assert mv[5] > 0 and mv[0] < C_Fx and abs(mv[1]) < abs(C_Mx) and mv[2] < 0 and mv[3] < 0 and abs(mv[4]) < abs(C_P)
    a = np.zeros(6); b = np.zeros(6)
    L.fo_prop_lookup_eval(C.c_double(1.5), C.c_double(1.5), dp(a)); L.fo_prop_lookup_eval(C.c_double(3.0), C.c_double(2.0), dp(b))
    assert np.array_equal(a, b)
    wr = np.zeros(6)
    L.fo_propeller_f_ode(-1, dp(arr(1.0, 0, 0)), dp(arr(50, 0, 0)), C.c_double(-300.0), dp(wr))
    assert wr[0] > 0 and wr[3] > 0   # CCW: thrust forward, reaction torque positive along x
    # F_z < 0 and τ_z > 0 in the reference need an angle of attack at the disc; KinData default has none, so they vanish
    wr2 = np.zeros(6)
    L.fo_propeller_f_ode(1, dp(arr(1.0, 0, 0)), dp(arr(50, 0, 5)), C.c_double(300.0), dp(wr2))
    assert wr2[0] > 0 and wr2[2] < 0 and wr2[3] < 0


def test_piston_lookup_known_answers(L):
    """FlightPhysics/test/test_piston.jl:57-127 — IO-360 chart known answers on PistonEngineLookup(0.15, 1.4)."""
    ns, nm, w_r, P = 0.15, 1.4, 2700.0, 200.0
    inHg = lambda p: 3386.389 * p
    p_std = 101325.0
    ft = lambda h: 0.3048 * h
    h2d = lambda h: L.fo_h2delta(C.c_double(h))
    lk = lambda which, a, b, c=0.0: L.fo_piston_lookup(C.c_double(ns), C.c_double(nm), which, C.c_double(a), C.c_double(b), C.c_double(c))
    for rpm, map_, h in ((1800, 20, 9500), (2700, 22, 7000), (2100, 16, 15250), (2300, 12, 22000)):
        assert abs(lk(0, rpm / w_r, inHg(map_) / p_std) - h2d(ft(h))) < 0.1
    for rpm, map_, hp in ((1800, 20, 71), (2050, 24, 113), (2400, 17, 85), (2400, 28.8, 176)):
        assert abs(lk(1, rpm / w_r, inHg(map_) / p_std) * P - hp) < 1
    for rpm, h, hp in ((1800, 3e3, 108), (2300, 2.4e3, 153), (2500, 10e3, 129), (2000, 20e3, 65)):
        assert abs(lk(2, rpm / w_r, h2d(ft(h))) * P - hp) < 3
    assert abs(lk(3, ns, 0, 1)) < 1e-12
    assert abs(lk(3, ns, lk(4, ns, 1.0), 1)) < 1e-12
    assert abs(lk(3, 0.5 * ns, 0.5, 1)) < 1e-12
    assert lk(3, 1.5 * ns, 0.5, 1) > lk(3, 1.5 * ns, 0.3, 1)
    assert 71 < lk(3, 1800 / w_r, inHg(20) / p_std, h2d(ft(3e3))) * P < 84
    assert 131 < lk(3, 2310 / w_r, inHg(23.6) / p_std, h2d(ft(2.4e3))) * P < 139
    assert 102 < lk(3, 2500 / w_r, inHg(18) / p_std, h2d(ft(10e3))) * P < 119


def test_engine_state_machine_and_response(L):
    """FlightPhysics/test/test_piston.jl:129-209."""
    OFF, STARTING, RUNNING = 0, 1, 2
    w_idle, w_stall = 600 * np.pi / 30, 300 * np.pi / 30
    assert L.fo_engine_tau_shaft(C.c_double(0.0), OFF, C.c_double(0.0)) == 0
    assert L.fo_engine_f_step(OFF, C.c_double(0.0), 1, 1) == STARTING
    assert L.fo_engine_f_step(STARTING, C.c_double(0.9 * w_idle), 1, 1) == STARTING
    assert L.fo_engine_tau_shaft(C.c_double(0.9 * w_idle), STARTING, C.c_double(0.0)) > 0
    assert L.fo_engine_f_step(STARTING, C.c_double(1.1 * w_idle), 1, 1) == RUNNING
    assert L.fo_engine_tau_shaft(C.c_double(1.1 * w_idle), RUNNING, C.c_double(0.1)) > 0
    assert L.fo_engine_f_step(RUNNING, C.c_double(1.1 * w_idle), 2, 1) == OFF           # commanded stop
    assert L.fo_engine_f_step(RUNNING, C.c_double(0.95 * w_stall), 0, 1) == OFF         # stall stop
    assert L.fo_engine_f_step(RUNNING, C.c_double(1.1 * w_idle), 0, 0) == OFF           # no fuel
    assert L.fo_engine_f_step(OFF, C.c_double(1.1 * w_idle), 1, 0) == STARTING
    assert L.fo_engine_f_step(STARTING, C.c_double(1.1 * w_idle), 1, 0) != RUNNING
    assert L.fo_engine_f_step(STARTING, C.c_double(1.1 * w_idle), 1, 1) == RUNNING


def test_thruster_idle_rpm_and_shutdown(L):
    """FlightPhysics/test/test_piston.jl:211-260 — start command, two steps -> starting; 10 s -> running at
    ω ≈ ω_idle (atol 1); pushing, CW reaction torque negative; full throttle 5 s -> ω > 2 ω_idle; back to idle."""
    x = np.zeros(3); state = C.c_int32(0); out = np.zeros(3)
    w_idle = 600 * np.pi / 30
    L.fo_thruster_sim(dp(x), C.byref(state), 1, C.c_double(0.0), 1, C.c_double(0.02), C.c_int64(2), dp(out))
    assert state.value == 1
    L.fo_thruster_sim(dp(x), C.byref(state), 1, C.c_double(0.0), 1, C.c_double(0.02), C.c_int64(500), dp(out))
    assert state.value == 2 and abs(out[0] - w_idle) < 1
    assert out[1] > 0 and out[2] < 0
    L.fo_thruster_sim(dp(x), C.byref(state), 0, C.c_double(1.0), 1, C.c_double(0.02), C.c_int64(250), dp(out))
    assert out[0] > 2 * w_idle
    L.fo_thruster_sim(dp(x), C.byref(state), 0, C.c_double(0.0), 1, C.c_double(0.02), C.c_int64(250), dp(out))
    assert abs(out[0] - w_idle) < 1
    # starved engine shuts down and friction stops it (:300-308)
    L.fo_thruster_sim(dp(x), C.byref(state), 0, C.c_double(0.0), 0, C.c_double(0.02), C.c_int64(50), dp(out))
    assert state.value == 0
    L.fo_thruster_sim(dp(x), C.byref(state), 0, C.c_double(0.0), 0, C.c_double(0.02), C.c_int64(250), dp(out))
    assert abs(out[0]) < 1e-10


def test_landing_gear_unit_known_answers(L):
    """FlightPhysics/test/test_landing_gear.jl:93-214 — friction values, ξ ≈ -0.1, F_dmp ≈ 2500 / 3500, contact velocities."""
    assert isapprox(L.fo_get_mu(0, 0, C.c_double(0.0075)), 0.025)
    assert isapprox(L.fo_get_mu(1, 0, C.c_double(0.0075)), 0.5)
    assert isapprox(L.fo_get_mu(1, 1, C.c_double(1e-5)), 0.25)
    assert isapprox(L.fo_get_mu(1, 2, C.c_double(10.0)), 0.025)

    def unit(h, q=(1, 0, 0, 0), w=(0, 0, 0), v=(0, 0, 0), steer=0.0, brake=0.0, x=(0, 0)):
        o = np.zeros(23)
        L.fo_ldg_unit_f_ode(C.c_double(h), dp(np.array(list(q) + list(w) + list(v), dtype=np.float64)), C.c_double(steer), C.c_double(brake), dp(arr(*x)), dp(o))
        return dict(wow=o[0], xi=o[1], xi_dot=o[2], F=o[3], v_xy=o[4:6], mu_max=o[6:8], mu_eff=o[8:10], f_c=o[10:13], F_b=o[13:16],
                    xd=o[16:18], sat=o[18:20], mu_roll=o[20], dh=o[21])
    assert unit(1.1)["wow"] == 0
    r = unit(0.9)
    assert isapprox(r["xi"], -0.1) and isapprox(r["F"], 2500.0) and (r["mu_eff"] == 0).all() and (r["f_c"][:2] == 0).all() and r["f_c"][2] < 0
    r = unit(0.9, q=quat_from_euler(L, 0, 0, np.pi / 12))
    assert r["xi"] > -0.1 and r["F"] < 2500
    r = unit(0.9, v=(0, 0, 1))
    assert r["xi_dot"] < 0 and isapprox(r["F"], 3500.0)
    r = unit(0.9, v=(1e-4, 0, 0))
    assert np.allclose(r["v_xy"], [1e-4, 0], atol=1e-6) and r["mu_max"][0] <= r["mu_roll"] and r["mu_eff"][1] == 0
    assert r["mu_eff"][0] < 0 and abs(r["mu_eff"][0]) < r["mu_max"][0] and r["xd"][0] < 0
    r = unit(0.9, v=(0, -1e-4, 0))
    assert r["F_b"][1] > 0 and r["xd"][1] > 0
    r = unit(0.9, v=(0, -1, 0))
    assert r["sat"][1] == 1 and r["xd"][1] == 0
    r = unit(0.9, v=(10, 0, 1))
    assert np.allclose(r["v_xy"], [10, 0], atol=1e-5)
    r = unit(0.9, w=(1, 0, 0))
    assert np.allclose(r["v_xy"], [0, -0.9], atol=1e-5)
    r = unit(0.9, q=quat_from_euler(L, 0, 0, np.pi / 12), v=(1e-4, 0, 0))
    assert r["F_b"][1] < 0
    r = unit(0.9, v=(10, 0, 1), steer=0.5)
    assert r["v_xy"][0] < 10 and r["v_xy"][1] < 0
    r = unit(0.9, v=(1e-4, 0, 0), brake=1.0)
    assert r["mu_max"][0] > r["mu_roll"] and r["wow"] == 1


def test_c172s_trim_succeeds_and_holds(oracle):
    """FlightApps/test/c172/test_c172s.jl:22-38 — f_init!(vehicle, C172.TrimParameters()) succeeds
    (cost < 1e-16 from the default TrimState); FlightApps/test/c172/test_c172x1.jl:101-116 analogue — the
    trimmed aircraft holds ω_wb_b within 1e-5 and v_eb_b within 1e-2 for 10 s at dt = 0.01."""
    tp = np.zeros((18, 1)); tp[0] = 1; tp[3] = 1050; tp[5] = 50; tp[10] = 0.5; tp[11] = 0.5; tp[13:18, 0] = [75, 75, 0, 0, 50]
    ts0 = np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02]).reshape(7, 1)
    env = oracle.default_env()
    cost0 = oracle.lib.fo_c172_trim_cost(dp(np.ascontiguousarray(tp[:, 0])), dp(np.ascontiguousarray(ts0[:, 0])), dp(env))
    assert cost0 > 1e-3   # the default guess is not a trim point
    r = oracle.trim(tp, ts0, env)
    assert r["ok"].all() and r["cost"][0] < 1e-16
    xd, y, st = oracle.f_ode(r["x"], r["u"], r["ui"], r["s"], env)
    assert st[0] == 0 and y[78 + 1, 0] == 0 and y[89 + 1, 0] == 0 and y[100 + 1, 0] == 0   # no weight on wheels
    assert r["x"][9, 0] > 600 * np.pi / 30 and abs(xd[0, 0]) < 1e-10 and abs(xd[1, 0]) < 1e-10   # c172s.jl:256-262 asserts
    x1, s1, st1, traj = oracle.step(r["x"], r["u"], r["ui"], r["s"], env, 0.01, 1000, save_every=10)
    assert st1[0] == 0
    w0 = y[25:28, 0]
    for k in range(traj.shape[0]):
        xd_k, y_k, _ = oracle.f_ode(traj[k], r["u"], r["ui"], r["s"], env)
        assert np.all(np.abs(y_k[25:28, 0] - w0) < 1e-5)
        assert np.all(np.abs(traj[k][24:27, 0] - r["x"][24:27, 0]) < 1e-2)


def test_theta_constraint_gives_the_requested_flight_path_angle(oracle):
    """FPt/test_aircraft_base.jl:15-41: θ_constraint(v_wb_b, γ_wb_n, φ_nb) must return the pitch angle for which the wind-relative velocity has
    the requested inclination γ_wb_n at the bank angle φ_nb. The reference checks it on one hand-set (α, β, γ, φ); here on every trimmed state
    of a batch with random flight-path angles, sideslip and turn rates (so that the trim's bank angle is far from zero): in still air
    v_eb_n = v_wb_n, and KinData's γ_gnd = inclination(v_eb_n) (FP/kinematics.jl:77-91) must equal the TrimParameters' γ_wb_n."""
    n = 512
    rng = np.random.default_rng(12)
    tp = np.zeros((18, n)); tp[0] = 1; tp[3] = rng.uniform(500, 2500, n); tp[5] = rng.uniform(40, 52, n); tp[10] = 0.5; tp[11] = 0.5
    tp[13:18] = np.array([75, 75, 0, 0, 50.0])[:, None]
    tp[4] = rng.uniform(-3, 3, n)                  # ψ_nb
    tp[6] = rng.uniform(-0.07, 0.05, n)            # γ_wb_n
    tp[7] = rng.uniform(-0.08, 0.08, n)            # ψ_wb_dot: banked trims
    tp[9] = rng.uniform(-0.05, 0.05, n)            # β_a
    ts0 = np.tile(np.array([0.1, 0.0, 0.75, 0.47, 0.014, -0.0015, 0.02])[:, None], (1, n))
    env = oracle.default_env()
    r = oracle.trim(tp, ts0, env)
    ok = r["ok"]
    assert ok.mean() > 0.9 and np.abs(r["ts"][1][ok]).max() > 0.3          # bank angles up to ~0.4 rad
    xd, y, st = oracle.f_ode(r["x"], r["u"], r["ui"], r["s"], env)
    assert np.abs(y[39] - tp[6])[ok].max() < 1e-12                          # FB_Y_KIN + 39 = γ_gnd
