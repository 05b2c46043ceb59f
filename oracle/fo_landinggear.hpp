// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's landing gear unit (strut + contact + steering + braking).
// Follows lib/FlightPhysics/src/landinggear.jl:47-76,116-120,140-193,204-347,380-483,524-552
#pragma once
#include "fo_dynamics.hpp"

namespace fo {

struct Damper {  // landinggear.jl:140-145
    double k_s = 25000, k_d_ext = 1000, k_d_cmp = 1000, F_max = 50000;
};
// landinggear.jl:149-153
inline double damper_force(const Damper& c, double xi, double xi_dot) {
    const double k_d = (xi_dot > 0 ? c.k_d_ext : c.k_d_cmp);
    return -(c.k_s * xi + k_d * xi_dot);
}
struct FrictionCoefficients { double mu_s, mu_d, v_s, v_d; };
// landinggear.jl:171-175
inline double get_mu(const FrictionCoefficients& fr, double v) {
    const double k_sd = std::clamp((std::fabs(v) - fr.v_s) / (fr.v_d - fr.v_s), 0.0, 1.0);
    return k_sd * fr.mu_d + (1 - k_sd) * fr.mu_s;
}
inline FrictionCoefficients friction_rolling(int) { return {0.03, 0.02, 0.005, 0.01}; }  // :181-182
inline FrictionCoefficients friction_skidding(int srf) {                                 // :184-194
    if (srf == 0) return {0.75, 0.25, 0.005, 0.01};
    if (srf == 1) return {0.25, 0.15, 0.005, 0.01};
    return {0.075, 0.025, 0.005, 0.01};
}

struct StrutParams {
    FrameTransform t_bs;
    double l_0 = 0.0;
    Damper damper;
};
struct StrutY {  // landinggear.jl:210-222
    double dh = 0;
    bool wow = false;
    double xi = 0, xi_dot = 0, F_dmp_zs = 0, psi_sw = 0, alpha_ts = 0;
    FrameTransform t_sc, t_bc;
    double v_ec_xy[2] = {0, 0};
    int surface = 0;
};
enum SteeringKind : int { NO_STEERING = 0, DIRECT_STEERING = 1 };
enum BrakingKind : int { NO_BRAKING = 0, DIRECT_BRAKING = 1 };
struct GearUnitParams {
    StrutParams strut;
    int steering = NO_STEERING;
    double psi_max = PI / 6;   // DirectSteering default (landinggear.jl:47-49)
    int braking = NO_BRAKING;
    double eta_br = 1.0;       // DirectBraking default (:106-108)
    PIParams frc = {5.0, 400.0, 0.2, 1.0, -1.0, 1.0};  // Contact f_init! (:401-409)
};
struct GearUnitU {
    bool steering_engaged = true;
    double steering_input = 0;  // Ranged [-1,1]
    double brake_input = 0;     // Ranged [0,1]
};
struct ContactY {  // landinggear.jl:384-395
    double mu_roll = 0, mu_skid = 0, k_br = 0, psi_cv = 0;
    double mu_max[2] = {0, 0}, mu_eff[2] = {0, 0};
    V3 f_c, F_c;
    Wrench wr_b;
    PIOut frc[2];
};
struct GearUnitY { StrutY strut; ContactY contact; };

// landinggear.jl:228-328
inline int32_t strut_f_ode(const GearUnitParams& gp, const GearUnitU& gu, const Env& env, const KinData& kin, StrutY& y) {
    int32_t st = 0;
    const StrutParams& sp = gp.strut;
    const Quat q_bs = sp.t_bs.q;
    const V3 r_bs_b = sp.t_bs.r;
    const V3 e1 = {1, 0, 0}, e3 = {0, 0, 1};

    const Quat q_es = compose(kin.q_eb, q_bs);
    const V3 ks_e = rotate(q_es, e3);
    const V3 r_bs_e = rotate(kin.q_eb, r_bs_b);
    const V3 r_sw0_e = sp.l_0 * ks_e;
    const V3 r_ew0_e = kin.r_eb_e + r_bs_e + r_sw0_e;
    const GeoNE Ow0 = geographic_from_cartesian(r_ew0_e);
    if (!(Ow0.h_e >= H_MIN)) raise_status(st, ST_ALT_RANGE);
    const double he_Ow0 = Ow0.h_e;

    const V3 loc_Ot = Ow0.n_e;
    const double he_Ot = h_ellip_from_orth(env.h_trn, loc_Ot);

    y = StrutY{};
    y.dh = he_Ow0 - he_Ot;
    y.wow = y.dh <= 0;
    if (!y.wow) return st;

    const V3 r_et_e = cartesian_from_geographic(loc_Ot, he_Ot);
    const V3 r_es_e = kin.r_eb_e + r_bs_e;
    const V3 r_st_e = r_et_e - r_es_e;

    const V3 ut_n = {0, 0, 1};  // HorizontalTerrain normal (terrain.jl:46-48)
    const V3 ut_e = rotate(kin.q_en, ut_n);
    const double ut_ks = dot(ut_e, ks_e);
    const double l = dot(ut_e, r_st_e) / ut_ks;
    y.alpha_ts = std::acos(std::max(std::min(ut_ks, 1.0), -1.0));
    y.xi = std::min(0.0, l - sp.l_0);

    const V3 r_sc_s = e3 * (sp.l_0 + y.xi);
    const V3 r_sc_b = rotate(q_bs, r_sc_s);
    const V3 r_bc_b = r_sc_b + r_bs_b;

    const V3 v_ec_b_body = kin.v_eb_b + cross(kin.w_eb_b, r_bc_b);
    const V3 v_ec_s_body = rotate(inv(q_bs), v_ec_b_body);
    const double psi_v = std::atan2(v_ec_s_body.y, v_ec_s_body.x);

    // get_steering_angle (landinggear.jl:35, 72-76)
    double psi_sw = 0.0;
    if (gp.steering == DIRECT_STEERING)
        psi_sw = gu.steering_engaged ? std::clamp(gu.steering_input, -1.0, 1.0) * gp.psi_max : psi_v;
    y.psi_sw = psi_sw;
    const Quat q_sw = Rz(psi_sw);
    const Quat q_ns = compose(kin.q_nb, q_bs);
    const Quat q_nw = compose(q_ns, q_sw);

    const V3 kc_n = ut_n;
    const V3 iw_n = rotate(q_nw, e1);
    const V3 iw_n_trn = iw_n - dot(iw_n, kc_n) * kc_n;
    const V3 ic_n = normalize(iw_n_trn);
    const V3 jc_n = cross(kc_n, ic_n);
    M3 R_nc;
    R_nc.m[0][0] = ic_n.x; R_nc.m[0][1] = jc_n.x; R_nc.m[0][2] = kc_n.x;
    R_nc.m[1][0] = ic_n.y; R_nc.m[1][1] = jc_n.y; R_nc.m[1][2] = kc_n.y;
    R_nc.m[2][0] = ic_n.z; R_nc.m[2][1] = jc_n.z; R_nc.m[2][2] = kc_n.z;
    // q_ns' ∘ R_nc : mixed RQuat/RMatrix composition falls back to RQuat (attitude.jl:143)
    Quat q_sc = compose(inv(q_ns), quat_from_rmatrix(R_nc));
    const Quat q_bc = compose(q_bs, q_sc);

    y.t_sc = {r_sc_s, q_sc};
    y.t_bc = {r_bc_b, q_bc};

    const V3 v_ec_c_body = rotate(inv(q_bc), v_ec_b_body);
    q_sc = compose(inv(q_bs), q_bc);
    const V3 ks_c = rotate(inv(q_sc), e3);
    y.xi_dot = -v_ec_c_body.z / ks_c.z;
    y.F_dmp_zs = damper_force(sp.damper, y.xi, y.xi_dot);

    const V3 v_ec_dmp_c = ks_c * y.xi_dot;
    const V3 v_ec_c = v_ec_c_body + v_ec_dmp_c;
    if (!(std::fabs(v_ec_c.z) < 1e-8)) raise_status(st, ST_CONTACT_ASSERT);
    y.v_ec_xy[0] = v_ec_c.x; y.v_ec_xy[1] = v_ec_c.y;
    y.surface = env.surface;
    return st;
}
// landinggear.jl:331-347
inline int32_t strut_f_step(const StrutY& y) {
    int32_t st = 0;
    if (y.wow && rad2deg(y.alpha_ts) > 60) raise_status(st, ST_GROUND_CRASH);
    if (-y.xi_dot > 10) raise_status(st, ST_GROUND_CRASH);
    return st;
}
// landinggear.jl:411-476. x_frc[2]
inline void contact_f_ode(const GearUnitParams& gp, const GearUnitU& gu, const StrutY& s, const double* x_frc,
                          double* xdot_frc, ContactY& y) {
    y = ContactY{};
    for (int i = 0; i < 2; i++) xdot_frc[i] = pi_f_ode(gp.frc, -s.v_ec_xy[i], 0, x_frc[i], y.frc[i]);
    if (!s.wow) return;

    const double norm_v = std::sqrt(s.v_ec_xy[0] * s.v_ec_xy[0] + s.v_ec_xy[1] * s.v_ec_xy[1]);
    y.mu_roll = get_mu(friction_rolling(s.surface), norm_v);
    y.mu_skid = get_mu(friction_skidding(s.surface), norm_v);
    y.k_br = (gp.braking == DIRECT_BRAKING) ? std::clamp(gu.brake_input, 0.0, 1.0) * gp.eta_br : 0.0;  // :116-120
    const double mu_x = y.mu_roll + (y.mu_skid - y.mu_roll) * y.k_br;
    if (norm_v < 1e-3) y.psi_cv = PI / 2;
    else y.psi_cv = std::atan2(s.v_ec_xy[1], s.v_ec_xy[0]);
    const double psi_skid = deg2rad(10);
    const double psi_abs = std::fabs(y.psi_cv);
    double mu_y;
    if (psi_abs < psi_skid) mu_y = y.mu_skid * psi_abs / psi_skid;
    else if (psi_abs > PI - psi_skid) mu_y = y.mu_skid * (1 - (psi_skid + psi_abs - PI) / psi_skid);
    else mu_y = y.mu_skid;
    const double nrm = std::sqrt(mu_x * mu_x + mu_y * mu_y);
    const double sc = std::min(1.0, y.mu_skid / nrm);
    y.mu_max[0] = mu_x * sc; y.mu_max[1] = mu_y * sc;
    y.mu_eff[0] = y.frc[0].output * y.mu_max[0];
    y.mu_eff[1] = y.frc[1].output * y.mu_max[1];
    y.f_c = {y.mu_eff[0], y.mu_eff[1], -1};
    const V3 f_s = rotate(s.t_sc.q, y.f_c);
    double N = -s.F_dmp_zs / f_s.z;
    N = std::max(0.0, N);
    y.F_c = y.f_c * N;
    y.wr_b = translate(s.t_bc, Wrench{y.F_c, V3{}});
}
// landinggear.jl:524-537
inline int32_t gear_unit_f_ode(const GearUnitParams& gp, const GearUnitU& gu, const Env& env, const KinData& kin,
                               const double* x_frc, double* xdot_frc, GearUnitY& y) {
    const int32_t st = strut_f_ode(gp, gu, env, kin, y.strut);
    contact_f_ode(gp, gu, y.strut, x_frc, xdot_frc, y.contact);
    return st;
}
// landinggear.jl:539-548, 479-483
inline int32_t gear_unit_f_step(const GearUnitY& y, double* x_frc) {
    const int32_t st = strut_f_step(y.strut);
    if (!y.strut.wow) { x_frc[0] = 0; x_frc[1] = 0; }
    return st;
}

}  // namespace fo
