// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of SimpleWorld(Cessna172Sv0()) : one RHS evaluation (f_ode!), the
// discrete post-step update (f_step!), the RK4 stepper with Flight.jl's callback order, and trim.
// Follows lib/FlightApps/src/c172/c172.jl:18-942 ; c172s/c172s.jl:16-263 ; c172s/c172s0.jl:14-18
//         lib/FlightPhysics/src/aircraftbase.jl:49-252 ; world.jl:20-57
//         lib/FlightCore/src/sim.jl:183-231,301-381,390-414 (OrdinaryDiffEq RK4, fixed step)
// State layout: x[27] =
//   0-1 aero (α_filt, β_filt) | 2-7 ldg frc (left, right, nose; 2 each) | 8 fuel | 9-11 engine (ω, idle, frc)
//   12-15 q_wb | 16-19 q_ew | 20 h_e | 21-23 ω_eb_b | 24-26 v_eb_b
#pragma once
#include "fo_propulsion.hpp"
#include "fo_landinggear.hpp"
#include "fo_trim.hpp"

namespace fo {

constexpr int NX = 27;
enum XIdx { X_AFILT = 0, X_BFILT = 1, X_LDG = 2, X_FUEL = 8, X_ENG = 9, X_KIN = 12, X_DYN = 21 };

// ---------------------------------------------------------------------------------------------
// Aerodynamic coefficient tables, c172.jl:51-199 (JSBSim C172R data, as transcribed by the reference)
struct AeroTables {
    double C_D_zero = 0.027;
    Table1 C_D_de, C_D_beta, C_D_ge, C_D_df;
    Table2 C_D_alpha_df;
    double C_Y_dr = 0.1870, C_Y_da = 0.0;
    Table2 C_Y_beta_df, C_Y_p, C_Y_r;
    double C_L_de = 0.4300, C_L_q = 3.900, C_L_alpha_dot = 1.700;
    Table1 C_L_ge, C_L_df;
    Table2 C_L_alpha;
    double C_l_da = 0.229, C_l_dr = 0.0147, C_l_beta = -0.09226, C_l_p = -0.4840;
    Table2 C_l_r;
    double C_m_zero = 0.100, C_m_de = -1.1220, C_m_alpha = -1.8000, C_m_q = -12.400, C_m_alpha_dot = -7.2700;
    Table1 C_m_df;
    double C_n_dr = -0.0430, C_n_da = -0.0053, C_n_beta = 0.05874, C_n_p = -0.0278, C_n_r = -0.0937;

    void build() {
        auto deg = [](std::vector<double> v) { for (auto& x : v) x = deg2rad(x); return v; };
        C_D_de.k = {-1.0, 0.0, 1.0}; C_D_de.v = {0.06, 0.0, 0.06};
        C_D_beta.k = {-1.0, 0.0, 1.0}; C_D_beta.v = {0.17, 0.0, 0.17};
        C_D_ge.k = {0.0000, 0.1000, 0.1500, 0.2000, 0.3000, 0.4000, 0.5000, 0.6000, 0.7000, 0.8000, 0.9000, 1.0000, 1.1000};
        C_D_ge.v = {0.4800, 0.5150, 0.6290, 0.7090, 0.8150, 0.8820, 0.9280, 0.9620, 0.9880, 1.0000, 1.0000, 1.0000, 1.0000};
        C_D_df.k = deg({0, 10, 20, 30}); C_D_df.v = {0.0000, 0.0070, 0.0120, 0.0180};
        C_D_alpha_df.k1 = {-0.0873, -0.0698, -0.0524, -0.0349, -0.0175, 0.0000, 0.0175, 0.0349, 0.0524, 0.0698, 0.0873, 0.1047, 0.1222,
                           0.1396, 0.1571, 0.1745, 0.1920, 0.2094, 0.2269, 0.2443, 0.2618, 0.2793, 0.2967, 0.3142, 0.3316, 0.3491};
        C_D_alpha_df.k2 = deg({0, 10, 20, 30});
        const double cd[4][26] = {
            {0.0041, 0.0013, 0.0001, 0.0003, 0.0020, 0.0052, 0.0099, 0.0162, 0.0240, 0.0334, 0.0442, 0.0566, 0.0706, 0.0860, 0.0962, 0.1069, 0.1180, 0.1298, 0.1424, 0.1565, 0.1727, 0.1782, 0.1716, 0.1618, 0.1475, 0.1097},
            {0.0000, 0.0004, 0.0023, 0.0057, 0.0105, 0.0168, 0.0248, 0.0342, 0.0452, 0.0577, 0.0718, 0.0874, 0.1045, 0.1232, 0.1353, 0.1479, 0.1610, 0.1746, 0.1892, 0.2054, 0.2240, 0.2302, 0.2227, 0.2115, 0.1951, 0.1512},
            {0.0005, 0.0025, 0.0059, 0.0108, 0.0172, 0.0251, 0.0346, 0.0457, 0.0583, 0.0724, 0.0881, 0.1053, 0.1240, 0.1442, 0.1573, 0.1708, 0.1849, 0.1995, 0.2151, 0.2323, 0.2521, 0.2587, 0.2507, 0.2388, 0.2214, 0.1744},
            {0.0014, 0.0041, 0.0084, 0.0141, 0.0212, 0.0299, 0.0402, 0.0521, 0.0655, 0.0804, 0.0968, 0.1148, 0.1343, 0.1554, 0.1690, 0.1830, 0.1975, 0.2126, 0.2286, 0.2464, 0.2667, 0.2735, 0.2653, 0.2531, 0.2351, 0.1866}};
        C_D_alpha_df.v.assign(26 * 4, 0.0);
        for (int j = 0; j < 4; j++) for (int i = 0; i < 26; i++) C_D_alpha_df.v[i + 26 * j] = cd[j][i];

        C_Y_beta_df.k1 = {-0.3490, 0, 0.3490}; C_Y_beta_df.k2 = deg({0, 30});
        C_Y_beta_df.v = {0.1370, 0.0000, -0.1370, 0.1060, 0.0000, -0.1060};  // column-major 3x2
        C_Y_p.k1 = {0.0, 0.094}; C_Y_p.k2 = deg({0, 30});
        C_Y_p.v = {-0.0750, -0.1450, -0.1610, -0.2310};
        C_Y_r.k1 = {0.0, 0.094}; C_Y_r.k2 = deg({0, 30});
        C_Y_r.v = {0.2140, 0.2670, 0.1620, 0.2150};

        C_L_ge.k = C_D_ge.k;
        C_L_ge.v = {1.2030, 1.1270, 1.0900, 1.0730, 1.0460, 1.0550, 1.0190, 1.0130, 1.0080, 1.0060, 1.0030, 1.0020, 1.0000};
        C_L_alpha.k1 = {-0.0900, 0.0000, 0.0900, 0.1000, 0.1200, 0.1400, 0.1600, 0.1700, 0.1900, 0.2100, 0.2400, 0.2600, 0.2800, 0.3000, 0.3200, 0.3400, 0.3600};
        C_L_alpha.k2 = {0.0, 1.0};
        const double cl[2][17] = {
            {-0.2200, 0.2500, 0.7300, 0.8300, 0.9200, 1.0200, 1.0800, 1.1300, 1.1900, 1.2500, 1.3500, 1.4400, 1.4700, 1.4300, 1.3800, 1.3000, 1.1500},
            {-0.2200, 0.2500, 0.7300, 0.7800, 0.7900, 0.8100, 0.8200, 0.8300, 0.8500, 0.8600, 0.8800, 0.9000, 0.9200, 0.9500, 0.9900, 1.0500, 1.1500}};
        C_L_alpha.v.assign(34, 0.0);
        for (int j = 0; j < 2; j++) for (int i = 0; i < 17; i++) C_L_alpha.v[i + 17 * j] = cl[j][i];
        C_L_df.k = deg({0, 10, 20, 30}); C_L_df.v = {0.0000, 0.2, 0.3, 0.35};

        C_l_r.k1 = {0.0, 0.094}; C_l_r.k2 = deg({0, 30});
        C_l_r.v = {0.0798, 0.1869, 0.1246, 0.2317};

        C_m_df.k = deg({0, 10, 20, 30}); C_m_df.v = {0.0000, -0.0654, -0.0981, -0.1140};
    }
};

struct AeroCoeffs { double C_D = 0, C_Y = 0, C_L = 0, C_l = 0, C_m = 0, C_n = 0; };
// c172.jl:226-245
inline AeroCoeffs get_aero_coeffs(const AeroTables& t, double al, double be, double p_nd, double q_nd, double r_nd,
                                  double da, double dr, double de, double df, double ad_nd, double bd_nd,
                                  double dh_nd, bool stall) {
    al = std::clamp(al, -0.1, 0.36);
    be = std::clamp(be, -0.2, 0.2);
    ad_nd = std::clamp(ad_nd, -0.04, 0.04);
    bd_nd = std::clamp(bd_nd, -0.2, 0.2);
    (void)bd_nd;
    const double st = stall ? 1.0 : 0.0;
    AeroCoeffs c;
    c.C_D = t.C_D_zero + t.C_D_ge(dh_nd) * (t.C_D_alpha_df(al, df) + t.C_D_df(df)) + t.C_D_de(de) + t.C_D_beta(be);
    c.C_Y = t.C_Y_dr * dr + t.C_Y_da * da + t.C_Y_beta_df(be, df) + t.C_Y_p(al, df) * p_nd + t.C_Y_r(al, df) * r_nd;
    c.C_L = t.C_L_ge(dh_nd) * (t.C_L_alpha(al, st) + t.C_L_df(df)) + t.C_L_de * de + t.C_L_q * q_nd + t.C_L_alpha_dot * ad_nd;
    c.C_l = t.C_l_da * da + t.C_l_dr * dr + t.C_l_beta * be + t.C_l_p * p_nd + t.C_l_r(al, df) * r_nd;
    c.C_m = t.C_m_zero + t.C_m_de * de + t.C_m_df(df) + t.C_m_alpha * al + t.C_m_q * q_nd + t.C_m_alpha_dot * ad_nd;
    c.C_n = t.C_n_dr * dr + t.C_n_da * da + t.C_n_beta * be + t.C_n_p * p_nd + t.C_n_r * r_nd;
    return c;
}

struct AeroParams {  // c172.jl:247-258
    double S = 16.165, b = 10.912, c = 1.494;
    double de_range[2] = {deg2rad(-28), deg2rad(23)};
    double da_range[2] = {deg2rad(-20), deg2rad(20)};
    double dr_range[2] = {deg2rad(-16), deg2rad(16)};
    double df_range[2] = {deg2rad(0), deg2rad(30)};
    double alpha_stall[2] = {0.09, 0.36};
    double V_min = 1.0, tau = 0.02;
};
struct AeroY {
    double e = 0, a = 0, r = 0, f = 0, de = 0, da = 0, dr = 0, df = 0;
    double alpha = 0, beta = 0, alpha_filt = 0, beta_filt = 0, alpha_filt_dot = 0, beta_filt_dot = 0;
    bool stall = false;
    AeroCoeffs coeffs;
    Wrench wr_b;
};
// types.jl:66-73 : linear_scaling(u::Ranged{T,UMin,UMax}, range)
inline double linear_scaling(double u, double umin, double umax, const double* range) {
    return range[0] + (range[1] - range[0]) / (umax - umin) * (u - umin);
}
// c172.jl:307-373
inline void aero_f_ode(const AeroParams& p, const AeroTables& tb, const double* x_aero, const double* u_eaRf /*e,a,r,f*/,
                       bool stall, const Env& env, const AirData& air, const KinData& kin, double* xdot_aero, AeroY& y) {
    const double alpha_filt = x_aero[0], beta_filt = x_aero[1];
    const double e = u_eaRf[0], a = u_eaRf[1], r = u_eaRf[2], f = u_eaRf[3];
    const V3 v_wb_a = air.v_wb_b;  // f_ba is the identity transform (c172.jl:203)
    double alpha = 0, beta = 0;
    if (air.TAS > 0.1) airflow_angles(v_wb_a, alpha, beta);
    const double V = std::max(air.TAS, p.V_min);
    const double afd = 1 / p.tau * (alpha - alpha_filt);
    const double bfd = 1 / p.tau * (beta - beta_filt);
    const double p_nd = kin.w_wb_b.x * p.b / (2 * V);
    const double q_nd = kin.w_wb_b.y * p.c / (2 * V);
    const double r_nd = kin.w_wb_b.z * p.b / (2 * V);
    const double ad_nd = afd * p.c / (2 * V);
    const double bd_nd = bfd * p.b / (2 * V);
    const double de = linear_scaling(e, -1, 1, p.de_range);
    const double da = linear_scaling(a, -1, 1, p.da_range);
    const double dr = linear_scaling(r, -1, 1, p.dr_range);
    const double df = linear_scaling(f, 0, 1, p.df_range);
    const double dh_nd = (kin.h_o - env.h_trn) / p.b;
    const AeroCoeffs c = get_aero_coeffs(tb, alpha, beta, p_nd, q_nd, r_nd, da, dr, de, df, ad_nd, bd_nd, dh_nd, stall);
    const Quat q_as = Ry(-alpha);  // atmosphere.jl:353-356
    const V3 F_aero_s = (air.q * p.S) * V3{-c.C_D, c.C_Y, -c.C_L};
    const V3 F_aero_a = rotate(q_as, F_aero_s);
    const V3 tau_aero_a = (air.q * p.S) * V3{c.C_l * p.b, c.C_m * p.c, c.C_n * p.b};
    xdot_aero[0] = afd; xdot_aero[1] = bfd;
    y.e = e; y.a = a; y.r = r; y.f = f; y.de = de; y.da = da; y.dr = dr; y.df = df;
    y.alpha = alpha; y.beta = beta; y.alpha_filt = alpha_filt; y.beta_filt = beta_filt;
    y.alpha_filt_dot = afd; y.beta_filt_dot = bfd; y.stall = stall; y.coeffs = c;
    y.wr_b = {F_aero_a, tau_aero_a};
}

// ---------------------------------------------------------------------------------------------
// Inputs of the C172S (MechanicalActuation + engine + payload), c172s.jl:62-72, piston.jl:259-267, c172.jl:521-527
struct C172Inputs {
    double throttle = 0, mixture = 0.5;
    double aileron = 0, elevator = 0, rudder = 0;
    double aileron_offset = 0, elevator_offset = 0, rudder_offset = 0;
    double flaps = 0, brake_left = 0, brake_right = 0;
    double m_pilot = 75, m_copilot = 75, m_lpass = 0, m_rpass = 0, m_baggage = 50;
    bool eng_start = false, eng_stop = false;
    int mixture_ctl = MIX_AUTO;
    bool steering_engaged = true;
};
struct C172Disc {  // discrete states
    bool stall = false;
    int eng_state = ENG_OFF;
};

// All constant model data of Cessna172Sv0
enum KinKind : int { KIN_WA = 0, KIN_ECEF = 1, KIN_NED = 2 };  // kinematics.jl:148, 250, 329
struct C172Model {
    int kin = KIN_WA;  // kinematic mechanisation of the vehicle; rows X_KIN.. hold its 9 / 8 / 6 states, the rest stay zero
    AeroParams aero;
    AeroTables aero_tb;
    GearUnitParams ldg[3];  // left, right, nose
    EngineParams eng;
    PistonLookup eng_lookup;
    PropParams prop;
    PropLookup prop_lookup;
    double gear_ratio = 1.0;
    MassProperties mp_afm;
    double m_full = 114.4, m_res = 1.0;  // c172.jl:589-592
    V3 pld_slots[5] = {{0.183, -0.356, 0.899}, {0.183, 0.356, 0.899}, {-0.681, -0.356, 0.899}, {-0.681, 0.356, 0.899}, {-1.316, 0, 0.899}};
    V3 fuel_left = {0.325, -2.845, 0}, fuel_right = {0.325, 2.845, 0};

    void build() {
        aero_tb.build();
        // c172.jl:442-476
        Damper mlg{39404, 9340, 9340, 50000}, nlg{26269, 3503, 3503, 50000};
        ldg[0].strut.t_bs.r = {-0.381, -1.092, 1.902}; ldg[0].strut.damper = mlg; ldg[0].braking = DIRECT_BRAKING;
        ldg[1].strut.t_bs.r = {-0.381, 1.092, 1.902};  ldg[1].strut.damper = mlg; ldg[1].braking = DIRECT_BRAKING;
        ldg[2].strut.t_bs.r = {1.27, 0, 1.9};          ldg[2].strut.damper = nlg; ldg[2].steering = DIRECT_STEERING;
        eng_lookup.build(eng.w_stall / eng.w_rated, eng.w_max / eng.w_rated);
        prop.t_bp.r = {2.055, 0, 0.833};  // c172s.jl:28-30
        prop_lookup.build(2, Blade{});
        // c172.jl:26-35
        FrameTransform t_bc;
        t_bc.r = {0.056, 0, 0.582};
        mp_afm = mp_rigid_body(767.0, diag3(820.0, 1164.0, 1702.0), t_bc);
    }
};

struct C172Y {
    KinData kin;
    AirData air;
    AeroY aero;
    GearUnitY ldg[3];
    ThrusterY pwp;
    double fuel_x_avail = 0, fuel_m_total = 0, fuel_m_avail = 0;
    DynamicsData dyn;
};

// One full RHS: world.jl:26-32 -> aircraftbase.jl:221-230 -> :142-170 -> c172.jl:697-713.
// Returns status bits.
inline int32_t c172_f_ode(const C172Model& M, const Env& env, const C172Inputs& u, const C172Disc& s,
                          const double* x, double* xdot, C172Y& y) {
    int32_t st = 0;
    // kinematics.u .= dynamics.x ; f_ode!(kinematics)
    if (M.kin == KIN_WA) st |= wa_f_ode(x + X_KIN, x + X_DYN, xdot + X_KIN, y.kin);
    else {
        for (int k = 0; k < 9; k++) xdot[X_KIN + k] = 0;
        if (M.kin == KIN_ECEF) ecef_f_ode(x + X_KIN, x + X_DYN, xdot + X_KIN, y.kin);
        else ned_f_ode(x + X_KIN, x + X_DYN, xdot + X_KIN, y.kin);
        if (!(y.kin.h_e >= H_MIN) || !(y.kin.h_o >= H_MIN)) raise_status(st, ST_ALT_RANGE);
    }
    y.air = air_data(env, y.kin, st);

    // ---- systems (c172.jl:697-713) ----
    // act (c172s.jl:92-120): Ranged inputs saturate at assignment and at every + / - (types.jl:44-51)
    auto rng = [](double v, double lo, double hi) { return std::min(std::max(v, lo), hi); };
    const double ail = rng(u.aileron, -1, 1), elv = rng(u.elevator, -1, 1), rud = rng(u.rudder, -1, 1);
    const double ail_o = rng(u.aileron_offset, -1, 1), elv_o = rng(u.elevator_offset, -1, 1), rud_o = rng(u.rudder_offset, -1, 1);
    GearUnitU gu[3];
    gu[2].steering_engaged = u.steering_engaged;
    gu[2].steering_input = rng(rud_o + rud, -1, 1);
    gu[0].brake_input = rng(u.brake_left, 0, 1);
    gu[1].brake_input = rng(u.brake_right, 0, 1);
    const double aero_u[4] = {rng(-rng(elv_o + elv, -1, 1), -1, 1), rng(ail_o + ail, -1, 1),
                              rng(-rng(rud_o + rud, -1, 1), -1, 1), rng(u.flaps, 0, 1)};
    aero_f_ode(M.aero, M.aero_tb, x + X_AFILT, aero_u, s.stall, env, y.air, y.kin, xdot + X_AFILT, y.aero);
    for (int i = 0; i < 3; i++)
        st |= gear_unit_f_ode(M.ldg[i], gu[i], env, y.kin, x + X_LDG + 2 * i, xdot + X_LDG + 2 * i, y.ldg[i]);
    EngineU eu;
    eu.start = u.eng_start; eu.stop = u.eng_stop; eu.throttle = u.throttle; eu.mixture_ctl = u.mixture_ctl; eu.mixture = u.mixture;
    thruster_f_ode(M.eng, M.eng_lookup, M.prop, M.prop_lookup, M.gear_ratio, eu, s.eng_state, x + X_ENG, y.air, y.kin,
                   xdot + X_ENG, y.pwp);
    // fuel (c172.jl:607-616)
    const double x_avail = x[X_FUEL];
    y.fuel_x_avail = x_avail;
    y.fuel_m_total = M.m_res + x_avail * (M.m_full - M.m_res);
    y.fuel_m_avail = y.fuel_m_total - M.m_res;
    xdot[X_FUEL] = -y.pwp.engine.mdot / (M.m_full - M.m_res);

    // ---- aggregate (dynamics.jl:328-399; field order afm, aero, ldg, fuel, pld, pwp, act) ----
    DynamicsU du;
    {
        MassProperties mp;              // p = MassProperties()
        mp = mp + M.mp_afm;             // afm
        mp = mp + MassProperties{};     // aero
        mp = mp + MassProperties{};     // ldg
        {                               // fuel (c172.jl:618-636)
            const double m_fuel = std::max(0.0, y.fuel_m_total);
            MassProperties f;
            f = f + mp_point(0.5 * m_fuel, M.fuel_left);
            f = f + mp_point(0.5 * m_fuel, M.fuel_right);
            mp = mp + f;
        }
        {                               // pld (c172.jl:542-554)
            const double ms[5] = {rng(u.m_pilot, 0, 100), rng(u.m_copilot, 0, 100), rng(u.m_lpass, 0, 100),
                                  rng(u.m_rpass, 0, 100), rng(u.m_baggage, 0, 100)};
            MassProperties p;
            for (int i = 0; i < 5; i++) p = p + mp_point(ms[i], M.pld_slots[i]);
            mp = mp + p;
        }
        mp = mp + MassProperties{};     // pwp
        mp = mp + MassProperties{};     // act
        du.mp_S_b = mp;
    }
    {
        Wrench wr;                      // afm
        wr = wr + Wrench{};
        wr = wr + y.aero.wr_b;          // aero
        {                               // ldg: left, right, nose
            Wrench l;
            for (int i = 0; i < 3; i++) l = l + y.ldg[i].contact.wr_b;
            wr = wr + l;
        }
        wr = wr + Wrench{};             // fuel
        wr = wr + Wrench{};             // pld
        wr = wr + y.pwp.propeller.wr_b; // pwp
        wr = wr + Wrench{};             // act
        du.wr_S_b = wr;
    }
    du.ho_S_b = y.pwp.propeller.hr_b;
    du.q_eb = y.kin.q_eb;
    du.r_eb_e = y.kin.r_eb_e;
    st |= dynamics_f_ode(x + X_DYN, du, xdot + X_DYN, y.dyn);
    return st;
}

// f_step!(world): world.jl:34-39 -> aircraftbase.jl:244-252,172-181 -> c172.jl:715-724
// Reads y of the LAST f_ode! call. Returns status bits; sets *modified if x or s changed.
inline int32_t c172_f_step(const C172Model& M, const C172Inputs& u, C172Disc& s, double* x, const C172Y& y, bool* modified = nullptr) {
    int32_t st = 0;
    double x0[NX];
    for (int i = 0; i < NX; i++) x0[i] = x[i];
    const C172Disc s0 = s;
    if (M.kin == KIN_WA) wa_f_step(x + X_KIN);
    else if (M.kin == KIN_ECEF) ecef_f_step(x + X_KIN);   // NED: f_step! is a no-op (kinematics.jl:409)
    // aero stall hysteresis (c172.jl:375-384)
    if (y.aero.alpha > M.aero.alpha_stall[1]) s.stall = true;
    else if (y.aero.alpha < M.aero.alpha_stall[0]) s.stall = false;
    // landing gear (c172.jl:447 ; landinggear.jl:539-548)
    for (int i = 0; i < 3; i++) st |= gear_unit_f_step(y.ldg[i], x + X_LDG + 2 * i);
    // power plant (piston.jl:597-603, 428-453) ; is_fuel_available: c172.jl:641
    EngineU eu;
    eu.start = u.eng_start; eu.stop = u.eng_stop;
    s.eng_state = engine_f_step(M.eng, eu, s.eng_state, x[X_ENG], y.fuel_m_avail > 0);
    if (modified) {
        bool mod = (s.stall != s0.stall) || (s.eng_state != s0.eng_state);
        for (int i = 0; i < NX; i++) mod = mod || (x[i] != x0[i]);
        *modified = mod;
    }
    return st;
}

// Where a terminated simulation stopped (the FB_TERM_* codes of include/flightbatch.h, shared with the product).
struct Term {
    int32_t status = 0;   // the single status bit of the exception
    int where = 0;        // 0 none | 2-4 f_ode! at RK stage k2..k4 | 5 f_ode! at the new state | 6 f_step! | 7 f_ode! re-evaluation
    bool advanced = false;   // the RK update of this step was made before the throw (where >= 5 inside the step)
};
enum { TERM_NONE = 0, TERM_K2 = 2, TERM_K3 = 3, TERM_K4 = 4, TERM_NEW = 5, TERM_F_STEP = 6, TERM_REEVAL = 7 };

// One fixed-step RK4 step followed by the discrete callbacks (sim.jl:204-218, 318-328).
// OrdinaryDiffEqLowOrderRK RK4 perform_step!: k1 = f(x_n) [FSAL: re-evaluated after the u-modifying
// step callback], k2 = f(x + dt/2 k1), k3 = f(x + dt/2 k2), k4 = f(x + dt k3),
// x_{n+1} = x + dt/6 (2(k2 + k3) + (k1 + k4)), then f(x_{n+1}) (fills y), then callbacks.
// `n_rhs` counts RHS evaluations the reference would make (6 per step).
//
// TERMINATION (sim.jl:561-570): the first exception ends the simulation, and what the reference leaves behind is mdl.x / mdl.s as
// they stand at the throw. f_ode_wrapper! copies the integrator's argument into mdl.x before it calls f_ode! (sim.jl:306), so after an
// exception inside a stage evaluation mdl.x IS that stage's argument; an exception out of f_step! (cb_step_affect!, sim.jl:318-328)
// leaves x_{n+1} with the part of f_step! that ran before it. The restatement runs in throwing mode and catches where step!(sim) is
// left: x holds mdl.x, `term` says where. Returns the status bit (0: the step completed).
inline int32_t c172_step(const C172Model& M, const Env& env, const C172Inputs& u, C172Disc& s, double* x, double dt,
                         C172Y& y, long* n_rhs = nullptr, bool reference_like = true, Term* term = nullptr) {
    ThrowScope throwing;
    double k1[NX], k2[NX], k3[NX], k4[NX], xt[NX];
    C172Y yt;
    int where = TERM_REEVAL;   // k1 at x_n: the re-evaluation that follows the previous step's callbacks (or the first one after init)
    bool advanced = false;
    try {
        c172_f_ode(M, env, u, s, x, k1, yt);
        const double hdt = dt / 2;
        for (int i = 0; i < NX; i++) xt[i] = x[i] + hdt * k1[i];
        where = TERM_K2;
        c172_f_ode(M, env, u, s, xt, k2, yt);
        for (int i = 0; i < NX; i++) xt[i] = x[i] + hdt * k2[i];
        where = TERM_K3;
        c172_f_ode(M, env, u, s, xt, k3, yt);
        for (int i = 0; i < NX; i++) xt[i] = x[i] + dt * k3[i];
        where = TERM_K4;
        c172_f_ode(M, env, u, s, xt, k4, yt);
        for (int i = 0; i < NX; i++) x[i] = x[i] + (dt / 6) * (2 * (k2[i] + k3[i]) + (k1[i] + k4[i]));
        advanced = true;
        where = TERM_NEW;
        c172_f_ode(M, env, u, s, x, k1, y);  // evaluation at the new state: y seen by f_step!, and logged
        where = TERM_F_STEP;
        c172_f_step(M, u, s, x, y);
        if (n_rhs) *n_rhs += reference_like ? 6 : 5;
        if (reference_like) {
            // reeval_internals_due_to_modification!: fsalfirst = f(u_modified) — the 6th evaluation.
            // It has no effect on the state trajectory (f is a pure function of x,u,s) and is only
            // executed so that the CPU baseline does the same amount of work as the reference.
            where = TERM_REEVAL;
            C172Y y2;
            double kk[NX];
            c172_f_ode(M, env, u, s, x, kk, y2);
        }
    } catch (const Termination& t) {
        if (where >= TERM_K2 && where <= TERM_K4)
            for (int i = 0; i < NX; i++) x[i] = xt[i];   // mdl.x .= u of the stage that threw (sim.jl:306)
        if (term) { term->status = t.bit; term->where = where; term->advanced = advanced; }
        return t.bit;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Trim (c172.jl:796-942 ; aircraftbase.jl:76-86,110-118 ; c172s.jl:168-263)
struct TrimState {  // c172.jl:796-804 defaults
    double alpha_a = 0.1, phi_nb = 0.0, n_eng = 0.75, throttle = 0.47, aileron = 0.014, elevator = -0.0015, rudder = 0.02;
};
struct TrimParams {  // c172.jl:806-818 defaults
    V3 n_e = {1, 0, 0};
    double h_e = 1050;
    double psi_nb = 0.0, EAS = 50.0, gamma_wb_n = 0.0, psi_wb_dot = 0.0, theta_wb_dot = 0.0, beta_a = 0.0;
    double fuel_load = 0.5, mixture = 0.5, flaps = 0.0;
    double payload[5] = {75, 75, 0, 0, 50};
};
// aircraftbase.jl:110-118
inline double theta_constraint(V3 v_wb_b, double gamma_wb_n, double phi_nb) {
    const double TAS = norm(v_wb_b);
    const double a = v_wb_b.x / TAS;
    const double b = (v_wb_b.y * std::sin(phi_nb) + v_wb_b.z * std::cos(phi_nb)) / TAS;
    const double sg = std::sin(gamma_wb_n);
    return std::atan((a * b + sg * std::sqrt(a * a + b * b - sg * sg)) / (a * a - sg * sg));
}
// c172.jl:825-854
inline KinInit trim_kin_init(const TrimState& ts, const TrimParams& tp, const Env& env) {
    int32_t st = 0;
    // AtmosphericData(atmosphere, Ob): Ob is Geographic{NVector,Ellipsoidal} -> HGeop via HOrth (geodesy.jl:245)
    const double h_o = h_orth_from_ellip(tp.h_e, tp.n_e);
    const AtmData atm = atmospheric_data(env, h_o, st);
    const double TAS = tp.EAS * std::sqrt(isa::rho_std / atm.rho);
    const V3 v_wb_a = velocity_vector(TAS, ts.alpha_a, tp.beta_a);
    const V3 v_wb_b = v_wb_a;
    const double theta_nb = theta_constraint(v_wb_b, tp.gamma_wb_n, ts.phi_nb);
    const Euler e_nb = {tp.psi_nb, theta_nb, ts.phi_nb};
    const Quat q_nb = quat_from_euler(e_nb);
    const V3 ed_wb = {tp.psi_wb_dot, tp.theta_wb_dot, 0.0};
    KinInit ki;
    ki.q_nb = q_nb;
    ki.n_e = tp.n_e;
    ki.h_e = tp.h_e;
    ki.w_wb_b = omega_from_euler_dot(e_nb, ed_wb);
    const V3 v_wb_n = rotate(q_nb, v_wb_b);
    ki.v_eb_n = atm.v + v_wb_n;
    return ki;
}
// c172s.jl:227-263 + c172s.jl:168-220 + aircraftbase.jl:76-86 : map (params, state) to x, u, s
inline void trim_assign(const C172Model& M, const TrimParams& tp, const TrimState& ts, const Env& env,
                        double* x, C172Inputs& u, C172Disc& s) {
    const KinInit ki = trim_kin_init(ts, tp, env);
    for (int i = 0; i < NX; i++) x[i] = 0;
    if (M.kin == KIN_WA) wa_init(ki, x + X_KIN, x + X_DYN);  // dynamics.x .= kinematics.u
    else if (M.kin == KIN_ECEF) ecef_init(ki, x + X_KIN, x + X_DYN);
    else ned_init(ki, x + X_KIN, x + X_DYN);
    u = C172Inputs{};
    u.m_pilot = tp.payload[0]; u.m_copilot = tp.payload[1]; u.m_lpass = tp.payload[2]; u.m_rpass = tp.payload[3]; u.m_baggage = tp.payload[4];
    u.throttle = ts.throttle; u.mixture = tp.mixture; u.mixture_ctl = MIX_AUTO;
    u.elevator = ts.elevator; u.aileron = ts.aileron; u.rudder = ts.rudder; u.flaps = tp.flaps;
    s.eng_state = ENG_RUNNING;
    s.stall = false;
    x[X_ENG] = ts.n_eng * M.eng.w_rated;
    x[X_ENG + 1] = 0; x[X_ENG + 2] = 0;
    x[X_AFILT] = ts.alpha_a; x[X_BFILT] = tp.beta_a;
    x[X_FUEL] = std::clamp(tp.fuel_load, 0.0, 1.0);
}
// residuals whose squared sum is the reference cost (c172.jl:857-867)
inline void trim_residuals(const C172Model& M, const TrimParams& tp, const TrimState& ts, const Env& env, double* r) {
    double x[NX], xd[NX];
    C172Inputs u; C172Disc s; C172Y y;
    trim_assign(M, tp, ts, env, x, u, s);
    c172_f_ode(M, env, u, s, x, xd, y);
    const double nv = norm(y.kin.v_eb_b);
    r[0] = xd[X_DYN + 3] / nv; r[1] = xd[X_DYN + 4] / nv; r[2] = xd[X_DYN + 5] / nv;
    r[3] = xd[X_DYN + 0]; r[4] = xd[X_DYN + 1]; r[5] = xd[X_DYN + 2];
    r[6] = xd[X_ENG] / M.eng.w_rated;
}
inline double trim_cost(const C172Model& M, const TrimParams& tp, const TrimState& ts, const Env& env) {
    double r[7];
    trim_residuals(M, tp, ts, env, r);
    double c = 0;
    for (int i = 0; i < 7; i++) c += r[i] * r[i];
    return c;
}
// f_init!(vehicle, TrimParameters) (c172.jl:883-942): see fo_trim.hpp for what is and is not restated of the
// reference's NLopt BOBYQA run. `ts` in: the initial guess (the reference always starts from TrimState()); out: the trim.
inline bool trim_solve(const C172Model& M, const TrimParams& tp, const Env& env, TrimState& ts, double* cost_out = nullptr,
                       TrimSolveStats* stats = nullptr, bool allow_continuation = true) {
    const double lo[7] = {-PI / 12, -PI / 3, 0.4, 0, -1, -1, -1};                    // c172.jl:901-908
    const double hi[7] = {M.aero.alpha_stall[1], PI / 3, 1.1, 1, 1, 1, 1};          // c172.jl:910-917
    auto unpack = [](const double* v) { return TrimState{v[0], v[1], v[2], v[3], v[4], v[5], v[6]}; };
    const double z0[7] = {ts.alpha_a, ts.phi_nb, ts.n_eng, ts.throttle, ts.aileron, ts.elevator, ts.rudder};
    double z[7];
    for (int k = 0; k < 7; k++) z[k] = z0[k];
    double cost = trim_tr_minimize([&](const double* v, double* r) { trim_residuals(M, tp, unpack(v), env, r); }, lo, hi, z, 500, stats);
    if (cost > 1e-16 && allow_continuation) {
        // continuation in the parameters from TrimParameters() (its trim from TrimState() is pinned by test_c172s.jl:22-38)
        const TrimParams d0{};
        auto blend = [&](double t) {
            TrimParams p = tp;
            auto mix = [t](double a, double b) { return a + t * (b - a); };
            p.h_e = mix(d0.h_e, tp.h_e); p.EAS = mix(d0.EAS, tp.EAS); p.gamma_wb_n = mix(d0.gamma_wb_n, tp.gamma_wb_n);
            p.psi_wb_dot = mix(d0.psi_wb_dot, tp.psi_wb_dot); p.theta_wb_dot = mix(d0.theta_wb_dot, tp.theta_wb_dot);
            p.beta_a = mix(d0.beta_a, tp.beta_a); p.fuel_load = mix(d0.fuel_load, tp.fuel_load); p.mixture = mix(d0.mixture, tp.mixture);
            p.flaps = mix(d0.flaps, tp.flaps);
            for (int k = 0; k < 5; k++) p.payload[k] = mix(d0.payload[k], tp.payload[k]);
            return p;   // location and heading are taken as requested: the trim barely depends on them
        };
        double zc[7];
        for (int k = 0; k < 7; k++) zc[k] = z0[k];
        double t = 0, dt = 0.125;
        double c = trim_tr_minimize([&](const double* v, double* r) { trim_residuals(M, blend(0.0), unpack(v), env, r); }, lo, hi, zc, 200, stats);
        bool good = c <= 1e-16;
        while (good && t < 1.0) {
            const double tn = std::min(1.0, t + dt);
            double zt[7];
            for (int k = 0; k < 7; k++) zt[k] = zc[k];
            const TrimParams p = blend(tn);
            c = trim_tr_minimize([&](const double* v, double* r) { trim_residuals(M, p, unpack(v), env, r); }, lo, hi, zt, 60, stats);
            if (c <= 1e-16) { t = tn; for (int k = 0; k < 7; k++) zc[k] = zt[k]; dt = std::min(2 * dt, 0.25); }
            else { dt *= 0.5; if (dt < 1.0 / 1024) good = false; }
        }
        if (good && c < cost) { cost = c; for (int k = 0; k < 7; k++) z[k] = zc[k]; if (stats) stats->continued = true; }
    }
    ts = unpack(z);
    if (cost_out) *cost_out = cost;
    return cost <= 1e-16;  // reference success criterion: STOPVAL_REACHED with stopval = 1e-16 (c172.jl:926,934)
}

}  // namespace fo
