// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's kinematics (WA / ECEF / NED) and ISA atmosphere + air data.
// Follows lib/FlightPhysics/src/kinematics.jl:14,21-91,114-118,152-242,250-320,329-425
//         lib/FlightPhysics/src/atmosphere.jl:22-34,99-135,198-242,269-283,320-356
#pragma once
#include "fo_geodesy.hpp"

namespace fo {

constexpr double V_MIN_CHI_GAMMA = 0.1;  // kinematics.jl:14

// kinematics.jl:21-40
struct KinInit {
    Quat q_nb;
    V3 n_e = {1, 0, 0};
    double h_e = 0;
    V3 w_wb_b;
    V3 v_eb_n;
};
// kinematics.jl:46-63 (40 doubles)
struct KinData {
    Euler e_nb;
    Quat q_nb, q_eb, q_en;
    LatLon ll;
    V3 n_e = {1, 0, 0};
    double h_e = 0, h_o = 0;
    V3 r_eb_e;
    V3 w_wb_b, w_eb_b, v_eb_b, v_eb_n;
    double v_gnd = 0, chi_gnd = 0, gamma_gnd = 0;
};

// kinematics.jl:232-242
inline V3 get_w_ew_n(V3 v_eb_n, V3 n_e, double h_e) {
    const Radii R = radii(n_e);
    return {v_eb_n.y / (R.N + h_e), -v_eb_n.x / (R.M + h_e), 0.0};
}
// kinematics.jl:413-425
inline V3 get_w_en_n(V3 v_eb_n, V3 n_e, double h_e) {
    const Radii R = radii(n_e);
    const double phi = latlon_from_nvector(n_e).phi;
    return {v_eb_n.y / (R.N + h_e), -v_eb_n.x / (R.M + h_e), -v_eb_n.y * std::tan(phi) / (R.N + h_e)};
}

// kinematics.jl:65-91
inline KinData kindata_from_init(const KinInit& ic) {
    KinData k;
    k.q_nb = ic.q_nb; k.n_e = ic.n_e; k.h_e = ic.h_e; k.w_wb_b = ic.w_wb_b; k.v_eb_n = ic.v_eb_n;
    k.e_nb = euler_from_quat(ic.q_nb);
    k.q_en = ltf(ic.n_e);
    k.q_eb = compose(k.q_en, ic.q_nb);
    k.ll = latlon_from_nvector(ic.n_e);
    k.h_o = h_orth_from_ellip(ic.h_e, ic.n_e);
    k.r_eb_e = cartesian_from_geographic(ic.n_e, ic.h_e);
    const V3 w_ew_n = get_w_ew_n(ic.v_eb_n, ic.n_e, ic.h_e);
    const V3 w_ew_b = rotate(inv(ic.q_nb), w_ew_n);
    k.w_eb_b = w_ew_b + ic.w_wb_b;
    k.v_eb_b = rotate(inv(ic.q_nb), ic.v_eb_n);
    k.v_gnd = norm(ic.v_eb_n);
    k.chi_gnd = k.v_gnd > V_MIN_CHI_GAMMA ? azimuth(ic.v_eb_n) : 0.0;
    k.gamma_gnd = k.v_gnd > V_MIN_CHI_GAMMA ? inclination(ic.v_eb_n) : 0.0;
    return k;
}

// ---- WA mechanisation. x_kin = [q_wb(4), q_ew(4), h_e] ; u = [w_eb_b(3), v_eb_b(3)] -------------
// kinematics.jl:155-178
inline void wa_init(const KinInit& ic, double* x_kin, double* u_vel) {
    const V3 w_ew_n = get_w_ew_n(ic.v_eb_n, ic.n_e, ic.h_e);
    const V3 w_ew_b = rotate(inv(ic.q_nb), w_ew_n);
    const V3 w_eb_b = w_ew_b + ic.w_wb_b;
    const V3 v_eb_b = rotate(inv(ic.q_nb), ic.v_eb_n);
    const Quat q_wb = ic.q_nb;
    const Quat q_ew = ltf(ic.n_e);
    u_vel[0] = w_eb_b.x; u_vel[1] = w_eb_b.y; u_vel[2] = w_eb_b.z;
    u_vel[3] = v_eb_b.x; u_vel[4] = v_eb_b.y; u_vel[5] = v_eb_b.z;
    x_kin[0] = q_wb.w; x_kin[1] = q_wb.x; x_kin[2] = q_wb.y; x_kin[3] = q_wb.z;
    x_kin[4] = q_ew.w; x_kin[5] = q_ew.x; x_kin[6] = q_ew.y; x_kin[7] = q_ew.z;
    x_kin[8] = ic.h_e;
}
// kinematics.jl:181-223. Returns status bits (altitude range).
inline int32_t wa_f_ode(const double* x_kin, const double* u_vel, double* xdot_kin, KinData& y) {
    int32_t st = 0;
    const Quat q_wb = {x_kin[0], x_kin[1], x_kin[2], x_kin[3]};
    const Quat q_ew = {x_kin[4], x_kin[5], x_kin[6], x_kin[7]};
    const V3 w_eb_b = {u_vel[0], u_vel[1], u_vel[2]};
    const V3 v_eb_b = {u_vel[3], u_vel[4], u_vel[5]};
    const double h_e = x_kin[8];
    if (!(h_e >= H_MIN)) raise_status(st, ST_ALT_RANGE);

    const double psi_nw = psi_nw_from_qew(q_ew);
    const Quat q_nw = Rz(psi_nw);
    const Quat q_nb = compose(q_nw, q_wb);
    const Quat q_eb = compose(q_ew, q_wb);
    const Quat q_en = compose(q_eb, inv(q_nb));
    const Euler e_nb = euler_from_quat(q_nb);

    const V3 n_e = nvector_from_qew(q_ew);
    const LatLon ll = latlon_from_nvector(n_e);
    const double h_o = h_orth_from_ellip(h_e, n_e);
    if (!(h_o >= H_MIN)) raise_status(st, ST_ALT_RANGE);

    const V3 v_eb_n = rotate(q_nb, v_eb_b);
    const V3 r_eb_e = cartesian_from_geographic(n_e, h_e);
    const V3 w_ew_n = get_w_ew_n(v_eb_n, n_e, h_e);

    const V3 w_ew_w = rotate(inv(q_nw), w_ew_n);
    const V3 w_ew_b = rotate(inv(q_wb), w_ew_w);
    const V3 w_wb_b = w_eb_b - w_ew_b;

    const double v_gnd = norm(v_eb_n);
    const double chi = v_gnd > V_MIN_CHI_GAMMA ? azimuth(v_eb_n) : 0.0;
    const double gam = v_gnd > V_MIN_CHI_GAMMA ? inclination(v_eb_n) : 0.0;

    const Quat qd_wb = qdot(q_wb, w_wb_b);
    const Quat qd_ew = qdot(q_ew, w_ew_w);
    xdot_kin[0] = qd_wb.w; xdot_kin[1] = qd_wb.x; xdot_kin[2] = qd_wb.y; xdot_kin[3] = qd_wb.z;
    xdot_kin[4] = qd_ew.w; xdot_kin[5] = qd_ew.x; xdot_kin[6] = qd_ew.y; xdot_kin[7] = qd_ew.z;
    xdot_kin[8] = -v_eb_n.z;

    y.e_nb = e_nb; y.q_nb = q_nb; y.q_eb = q_eb; y.q_en = q_en; y.ll = ll; y.n_e = n_e;
    y.h_e = h_e; y.h_o = h_o; y.r_eb_e = r_eb_e; y.w_wb_b = w_wb_b; y.w_eb_b = w_eb_b;
    y.v_eb_b = v_eb_b; y.v_eb_n = v_eb_n; y.v_gnd = v_gnd; y.chi_gnd = chi; y.gamma_gnd = gam;
    return st;
}
// kinematics.jl:114-118
inline void normalize_block(double* x, int n, double eps) {
    double s = 0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    const double nrm = std::sqrt(s);
    if (std::fabs(nrm - 1.0) > eps)
        for (int i = 0; i < n; i++) x[i] /= nrm;
}
// kinematics.jl:226-229
inline void wa_f_step(double* x_kin, double eps = 1e-8) {
    normalize_block(x_kin, 4, eps);
    normalize_block(x_kin + 4, 4, eps);
}

// ---- ECEF mechanisation. x = [q_eb(4), n_e(3), h_e] -- kinematics.jl:250-320 --------------------
inline void ecef_init(const KinInit& ic, double* x_kin, double* u_vel) {
    const Quat q_en = ltf(ic.n_e);
    const Quat q_eb = compose(q_en, ic.q_nb);
    const V3 w_ew_n = get_w_ew_n(ic.v_eb_n, ic.n_e, ic.h_e);
    const V3 w_ew_b = rotate(inv(ic.q_nb), w_ew_n);
    const V3 w_eb_b = w_ew_b + ic.w_wb_b;
    const V3 v_eb_b = rotate(inv(ic.q_nb), ic.v_eb_n);
    u_vel[0] = w_eb_b.x; u_vel[1] = w_eb_b.y; u_vel[2] = w_eb_b.z;
    u_vel[3] = v_eb_b.x; u_vel[4] = v_eb_b.y; u_vel[5] = v_eb_b.z;
    x_kin[0] = q_eb.w; x_kin[1] = q_eb.x; x_kin[2] = q_eb.y; x_kin[3] = q_eb.z;
    x_kin[4] = ic.n_e.x; x_kin[5] = ic.n_e.y; x_kin[6] = ic.n_e.z;
    x_kin[7] = ic.h_e;
}
inline void ecef_f_ode(const double* x_kin, const double* u_vel, double* xdot_kin, KinData& y) {
    const Quat q_eb = {x_kin[0], x_kin[1], x_kin[2], x_kin[3]};
    const V3 n_e = {x_kin[4], x_kin[5], x_kin[6]};
    const V3 w_eb_b = {u_vel[0], u_vel[1], u_vel[2]};
    const V3 v_eb_b = {u_vel[3], u_vel[4], u_vel[5]};
    const double h_e = x_kin[7];
    const double h_o = h_orth_from_ellip(h_e, n_e);
    const LatLon ll = latlon_from_nvector(n_e);
    const Quat q_en = ltf(n_e);
    const Quat q_nb = compose(inv(q_en), q_eb);
    const Euler e_nb = euler_from_quat(q_nb);
    const V3 r_eb_e = cartesian_from_geographic(n_e, h_e);
    const V3 v_eb_n = rotate(q_nb, v_eb_b);
    const V3 w_ew_n = get_w_ew_n(v_eb_n, n_e, h_e);
    const V3 w_ew_b = rotate(inv(q_nb), w_ew_n);
    const V3 w_wb_b = w_eb_b - w_ew_b;
    const double v_gnd = norm(v_eb_n);
    const double chi = v_gnd > V_MIN_CHI_GAMMA ? azimuth(v_eb_n) : 0.0;
    const double gam = v_gnd > V_MIN_CHI_GAMMA ? inclination(v_eb_n) : 0.0;
    const Quat qd = qdot(q_eb, w_eb_b);
    const V3 nd = rotate(q_en, cross(w_ew_n, V3{0, 0, -1}));
    xdot_kin[0] = qd.w; xdot_kin[1] = qd.x; xdot_kin[2] = qd.y; xdot_kin[3] = qd.z;
    xdot_kin[4] = nd.x; xdot_kin[5] = nd.y; xdot_kin[6] = nd.z;
    xdot_kin[7] = -v_eb_n.z;
    y.e_nb = e_nb; y.q_nb = q_nb; y.q_eb = q_eb; y.q_en = q_en; y.ll = ll; y.n_e = n_e;
    y.h_e = h_e; y.h_o = h_o; y.r_eb_e = r_eb_e; y.w_wb_b = w_wb_b; y.w_eb_b = w_eb_b;
    y.v_eb_b = v_eb_b; y.v_eb_n = v_eb_n; y.v_gnd = v_gnd; y.chi_gnd = chi; y.gamma_gnd = gam;
}
inline void ecef_f_step(double* x_kin, double eps = 1e-8) {
    normalize_block(x_kin, 4, eps);
    normalize_block(x_kin + 4, 3, eps);
}

// ---- NED mechanisation. x = [psi, theta, phi, lat, lon, h_e] -- kinematics.jl:329-425 ----------
inline void ned_init(const KinInit& ic, double* x_kin, double* u_vel) {
    const V3 w_ew_n = get_w_ew_n(ic.v_eb_n, ic.n_e, ic.h_e);
    const V3 w_ew_b = rotate(inv(ic.q_nb), w_ew_n);
    const V3 w_eb_b = w_ew_b + ic.w_wb_b;
    const V3 v_eb_b = rotate(inv(ic.q_nb), ic.v_eb_n);
    const Euler e = euler_from_quat(ic.q_nb);
    const LatLon ll = latlon_from_nvector(ic.n_e);
    u_vel[0] = w_eb_b.x; u_vel[1] = w_eb_b.y; u_vel[2] = w_eb_b.z;
    u_vel[3] = v_eb_b.x; u_vel[4] = v_eb_b.y; u_vel[5] = v_eb_b.z;
    x_kin[0] = e.psi; x_kin[1] = e.theta; x_kin[2] = e.phi;
    x_kin[3] = ll.phi; x_kin[4] = ll.lam; x_kin[5] = ic.h_e;
}
inline void ned_f_ode(const double* x_kin, const double* u_vel, double* xdot_kin, KinData& y) {
    const Euler e_nb = {x_kin[0], x_kin[1], x_kin[2]};
    const LatLon ll = {x_kin[3], x_kin[4]};
    const double h_e = x_kin[5];
    const V3 w_eb_b = {u_vel[0], u_vel[1], u_vel[2]};
    const V3 v_eb_b = {u_vel[3], u_vel[4], u_vel[5]};
    const V3 n_e = nvector_from_latlon(ll);
    const double h_o = h_orth_from_ellip(h_e, n_e);
    const Quat q_nb = quat_from_euler(e_nb);
    const Quat q_en = ltf(n_e);
    const Quat q_eb = compose(q_en, q_nb);
    const V3 v_eb_n = rotate(q_nb, v_eb_b);
    const V3 r_eb_e = cartesian_from_geographic(n_e, h_e);
    const V3 w_en_n = get_w_en_n(v_eb_n, n_e, h_e);
    const V3 w_en_b = rotate(inv(q_nb), w_en_n);
    const V3 w_nb_b = w_eb_b - w_en_b;
    const V3 w_ew_n = get_w_ew_n(v_eb_n, n_e, h_e);
    const V3 w_ew_b = rotate(inv(q_nb), w_ew_n);
    const V3 w_wb_b = w_eb_b - w_ew_b;
    const double v_gnd = norm(v_eb_n);
    const V3 ed = euler_dot(e_nb, w_nb_b);
    xdot_kin[0] = ed.x; xdot_kin[1] = ed.y; xdot_kin[2] = ed.z;
    xdot_kin[3] = -w_en_n.y;                      // geodesy.jl:112-118
    xdot_kin[4] = w_en_n.x / std::cos(ll.phi);
    xdot_kin[5] = -v_eb_n.z;
    y.e_nb = e_nb; y.q_nb = q_nb; y.q_eb = q_eb; y.q_en = q_en; y.ll = ll; y.n_e = n_e;
    y.h_e = h_e; y.h_o = h_o; y.r_eb_e = r_eb_e; y.w_wb_b = w_wb_b; y.w_eb_b = w_eb_b;
    y.v_eb_b = v_eb_b; y.v_eb_n = v_eb_n; y.v_gnd = v_gnd;
    y.chi_gnd = azimuth(v_eb_n); y.gamma_gnd = inclination(v_eb_n);
}

// =============================================================================================
// Atmosphere (atmosphere.jl)
namespace isa {
constexpr double R = 287.05287;
constexpr double gamma = 1.40;
constexpr double beta_s = 1.458e-6;
constexpr double S = 110.4;
constexpr double T_std = 288.15;
constexpr double p_std = 101325.0;
constexpr double rho_std = p_std / (R * T_std);
constexpr double g_std = 9.80665;
constexpr double layer_beta[7] = {-6.5e-3, 0, 1e-3, 2.8e-3, 0, -2.8e-3, -2e-3};
constexpr double layer_hceil[7] = {11000, 20000, 32000, 47000, 51000, 71000, 84852};
}  // namespace isa

inline double isa_T_law(double h, double T_b, double h_b, double beta) { return T_b + beta * (h - h_b); }  // :103
inline double isa_p_law(double h, double g0, double p_b, double T_b, double h_b, double beta) {           // :105-111
    if (beta != 0.0) return p_b * std::pow(1 + beta / T_b * (h - h_b), -g0 / (beta * isa::R));
    return p_b * std::exp(-g0 / (isa::R * T_b) * (h - h_b));
}
struct ISAData { double T = isa::T_std, p = isa::p_std; };
// atmosphere.jl:116-135
inline ISAData isa_data(double h_geop, ISAData sl, int32_t& st) {
    double h_base = 0, T_base = sl.T, p_base = sl.p;
    const double g0 = isa::g_std;
    for (int i = 0; i < 7; i++) {
        const double beta = isa::layer_beta[i], h_ceil = isa::layer_hceil[i];
        if (h_geop < h_ceil) {
            return {isa_T_law(h_geop, T_base, h_base, beta), isa_p_law(h_geop, g0, p_base, T_base, h_base, beta)};
        }
        const double T_ceil = isa_T_law(h_ceil, T_base, h_base, beta);
        const double p_ceil = isa_p_law(h_ceil, g0, p_base, T_base, h_base, beta);
        h_base = h_ceil; T_base = T_ceil; p_base = p_ceil;
    }
    raise_status(st, ST_ISA_RANGE);
    return {T_base, p_base};
}

struct AtmData { double T, p, rho, a, mu; V3 v; };
struct Env {              // SimpleWorld defaults: atmosphere.jl:75-78,165 ; terrain.jl:34-38
    double T_sl = isa::T_std;
    double p_sl = isa::p_std;
    V3 wind;              // NED
    double h_trn = 0.0;   // terrain orthometric elevation
    int surface = 0;      // 0 DryTarmac, 1 WetTarmac, 2 IcyTarmac
};
// atmosphere.jl:269-278 ; position given as (n_e, orthometric altitude)
inline AtmData atmospheric_data(const Env& env, double h_o, int32_t& st) {
    if (!(h_o >= H_MIN)) raise_status(st, ST_ALT_RANGE);
    const double h_g = h_geop_from_orth(h_o);
    const ISAData d = isa_data(h_g, {env.T_sl, env.p_sl}, st);
    AtmData a;
    a.T = d.T; a.p = d.p;
    a.rho = d.p / (isa::R * d.T);
    a.a = std::sqrt(isa::gamma * isa::R * d.T);
    a.mu = (isa::beta_s * std::pow(d.T, 1.5)) / (d.T + isa::S);
    a.v = env.wind;
    return a;
}
// atmosphere.jl:198-215 (22 doubles)
struct AirData {
    V3 v_ew_n, v_ew_b, v_wb_b;
    double T = 0, p = 0, rho = 0, a = 0, mu = 0, M = 0, Tt = 0, pt = 0, dp = 0, q = 0, TAS = 0, EAS = 0, CAS = 0;
};
// atmosphere.jl:220-242
inline AirData air_data(const AtmData& atm, const KinData& kin) {
    using namespace isa;
    AirData d;
    d.v_ew_n = atm.v;
    d.v_ew_b = rotate(inv(kin.q_nb), d.v_ew_n);
    d.v_wb_b = kin.v_eb_b - d.v_ew_b;
    d.T = atm.T; d.p = atm.p; d.rho = atm.rho; d.a = atm.a; d.mu = atm.mu;
    d.TAS = norm(d.v_wb_b);
    d.M = d.TAS / d.a;
    d.Tt = d.T * (1 + (gamma - 1) / 2 * (d.M * d.M));
    d.pt = d.p * std::pow(d.Tt / d.T, gamma / (gamma - 1));
    d.dp = d.pt - d.p;
    d.q = 1.0 / 2 * d.rho * (d.TAS * d.TAS);
    d.EAS = d.TAS * std::sqrt(d.rho / rho_std);
    d.CAS = std::sqrt(2 * gamma / (gamma - 1) * p_std / rho_std * (std::pow(1 + d.dp / p_std, (gamma - 1) / gamma) - 1));
    return d;
}
// atmosphere.jl:280-283
inline AirData air_data(const Env& env, const KinData& kin, int32_t& st) {
    return air_data(atmospheric_data(env, kin.h_o, st), kin);
}
constexpr double TAS_MIN_AB = 0.1;  // atmosphere.jl:320
// atmosphere.jl:329-337
inline void airflow_angles(V3 v, double& alpha, double& beta) {
    if (norm(v) < TAS_MIN_AB) { alpha = 0; beta = 0; return; }
    alpha = std::atan2(v.z, v.x);
    beta = std::atan2(v.y, std::sqrt(v.x * v.x + v.z * v.z));
}
// atmosphere.jl:323-326
inline V3 velocity_vector(double TAS, double alpha, double beta) {
    const double cb = std::cos(beta);
    return TAS * V3{std::cos(alpha) * cb, std::sin(beta), std::sin(alpha) * cb};
}

}  // namespace fo
