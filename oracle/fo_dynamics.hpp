// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's rigid-body dynamics and the continuous PI compensator.
// Follows lib/FlightPhysics/src/dynamics.jl:37-156,200-317,406-525
//         lib/FlightPhysics/src/control.jl:11-88
#pragma once
#include "fo_kinatm.hpp"

namespace fo {

// dynamics.jl:37-40 ; every C172 frame transform is a pure translation unless q is given
struct FrameTransform {
    V3 r;
    Quat q;
};
struct Wrench { V3 F, tau; };
inline Wrench operator+(Wrench a, Wrench b) { return {a.F + b.F, a.tau + b.tau}; }  // :131
// dynamics.jl:141-156
inline Wrench translate(const FrameTransform& t_bc, Wrench wr_c) {
    const V3 F_c_b = rotate(t_bc.q, wr_c.F);
    const V3 tau_c_b = rotate(t_bc.q, wr_c.tau);
    return {F_c_b, tau_c_b + cross(t_bc.r, F_c_b)};
}

// dynamics.jl:200-204
struct MassProperties {
    double m = 0;
    M3 J;
    V3 r_OG;
};
inline M3 skew_sq(V3 r) { const M3 S = v2skew(r); return S * S; }
// dynamics.jl:211-214 : point mass at r
inline MassProperties mp_point(double m, V3 r) {
    MassProperties p;
    p.m = m;
    p.J = (-m) * skew_sq(r);
    p.r_OG = r;
    return p;
}
// dynamics.jl:235-256 : rigid body (m, J_G_c) translated (pure translation) to frame b
inline MassProperties mp_rigid_body(double m, const M3& J_G_c, const FrameTransform& t_bc) {
    M3 J_G_b;
    if (!rq_equal(t_bc.q, Quat{})) {
        const M3 R = rmatrix_from_quat(t_bc.q);
        J_G_b = R * J_G_c * transpose(R);
    } else {
        J_G_b = J_G_c;
    }
    MassProperties p;
    p.m = m;
    p.J = J_G_b - m * skew_sq(t_bc.r);
    p.r_OG = t_bc.r;
    return p;
}
// dynamics.jl:262-272
inline MassProperties operator+(const MassProperties& p1, const MassProperties& p2) {
    const double m = p1.m + p2.m;
    if (m > 0) {
        MassProperties r;
        r.m = m;
        r.J = p1.J + p2.J;
        r.r_OG = (1 / m) * (p1.m * p1.r_OG + p2.m * p2.r_OG);
        return r;
    }
    return MassProperties{};
}
// dynamics.jl:284-317
inline MassProperties translate(const FrameTransform& t_bc, const MassProperties& mp_c) {
    const double m = mp_c.m;
    const M3 J_G_c = mp_c.J + m * skew_sq(mp_c.r_OG);
    M3 J_G_b;
    if (!rq_equal(t_bc.q, Quat{})) {
        const M3 R = rmatrix_from_quat(t_bc.q);
        J_G_b = R * J_G_c * transpose(R);
    } else {
        J_G_b = J_G_c;
    }
    const V3 r_cG_b = rotate(t_bc.q, mp_c.r_OG);
    const V3 r_bG_b = t_bc.r + r_cG_b;
    MassProperties p;
    p.m = m;
    p.J = J_G_b - m * skew_sq(r_bG_b);
    p.r_OG = r_bG_b;
    return p;
}

// dynamics.jl:416-434 (77 doubles)
struct DynamicsData {
    Wrench wr_S_c, wr_S_b;
    MassProperties mp_S_c, mp_S_b;
    V3 ho_S_b;
    V3 wd_ec_c, vd_ec_c, a_ec_c, a_ic_c, g_c_c, gam_c_c, f_c_c;
    V3 wd_eb_b, vd_eb_b, alpha_ib_b, a_eb_b, a_ib_b;
};
struct DynamicsU {
    MassProperties mp_S_b;
    Wrench wr_S_b;
    V3 ho_S_b;
    Quat q_eb;
    V3 r_eb_e;
};
// dynamics.jl:443-525. x_dyn = [w_eb_b(3), v_eb_b(3)]
// Returns status bits: ltf(Oc) / gravity(Oc) convert the CoM position to Geographic, whose Altitude constructor throws below h_min (geodesy.jl:218-221).
inline int32_t dynamics_f_ode(const double* x_dyn, const DynamicsU& u, double* xdot_dyn, DynamicsData& y) {
    int32_t st = 0;
    const V3 w_eb_b = {x_dyn[0], x_dyn[1], x_dyn[2]};
    const V3 v_eb_b = {x_dyn[3], x_dyn[4], x_dyn[5]};
    const Quat q_eb = u.q_eb;

    const V3 w_ie_e = {0, 0, wgs::w_ie};
    const V3 w_ie_b = rotate(inv(q_eb), w_ie_e);

    const V3 r_bc_b = u.mp_S_b.r_OG;
    FrameTransform t_cb;
    t_cb.r = -r_bc_b;

    const MassProperties mp_S_c = translate(t_cb, u.mp_S_b);
    const Wrench wr_S_c = translate(t_cb, u.wr_S_b);
    const V3 ho_S_c = u.ho_S_b;

    const V3 F_S_c = wr_S_c.F, tau_S_c = wr_S_c.tau;
    const double m_S = mp_S_c.m;
    const M3 J_S_c = mp_S_c.J;

    const V3 w_ec_c = w_eb_b;
    const V3 v_ec_c = v_eb_b + cross(w_ec_c, r_bc_b);

    const V3 w_ie_c = w_ie_b;
    const V3 w_ic_c = w_ie_c + w_ec_c;

    const V3 r_bc_e = rotate(q_eb, r_bc_b);
    const V3 r_ec_e = u.r_eb_e + r_bc_e;
    const GeoNE Oc = geographic_from_cartesian(r_ec_e);
    if (!(Oc.h_e >= H_MIN)) raise_status(st, ST_ALT_RANGE);

    const Quat q_el = ltf(Oc.n_e);
    const Quat q_be = inv(q_eb);
    const Quat q_ce = q_be;
    const Quat q_cl = compose(q_ce, q_el);

    // gravity(Oc) performs its own ECEF->geodetic conversion (geodesy.jl:453); same inputs, same result
    const V3 g_c_l = {0, 0, gravity(Oc.n_e, Oc.h_e)};
    const V3 g_c_c = rotate(q_cl, g_c_l);

    const V3 hc_S_c = J_S_c * w_ic_c + ho_S_c;
    const V3 wd_ec_c = solve3(J_S_c, tau_S_c - J_S_c * cross(w_ie_c, w_ec_c) - cross(w_ic_c, hc_S_c));
    const V3 vd_ec_c = (1 / m_S) * F_S_c + g_c_c - cross(w_ec_c + 2.0 * w_ie_c, v_ec_c);

    const V3 wd_eb_b = wd_ec_c;
    const V3 vd_eb_b = vd_ec_c - cross(wd_ec_c, r_bc_b);

    const V3 r_ec_c = rotate(q_ce, r_ec_e);
    const V3 r_eb_b = rotate(q_be, u.r_eb_e);

    y.a_ec_c = vd_ec_c + cross(w_ec_c, v_ec_c);
    y.a_ic_c = vd_ec_c + cross(w_ec_c + 2.0 * w_ie_c, v_ec_c) + cross(w_ie_c, cross(w_ie_c, r_ec_c));
    y.gam_c_c = g_c_c + cross(w_ie_c, cross(w_ie_c, r_ec_c));
    y.f_c_c = y.a_ic_c - y.gam_c_c;
    y.alpha_ib_b = wd_eb_b - cross(w_eb_b, w_ie_b);
    y.a_eb_b = vd_eb_b + cross(w_eb_b, v_eb_b);
    y.a_ib_b = vd_eb_b + cross(w_eb_b + 2.0 * w_ie_b, v_eb_b) + cross(w_ie_b, cross(w_ie_b, r_eb_b));

    xdot_dyn[0] = wd_eb_b.x; xdot_dyn[1] = wd_eb_b.y; xdot_dyn[2] = wd_eb_b.z;
    xdot_dyn[3] = vd_eb_b.x; xdot_dyn[4] = vd_eb_b.y; xdot_dyn[5] = vd_eb_b.z;

    y.wr_S_c = wr_S_c; y.wr_S_b = u.wr_S_b; y.mp_S_c = mp_S_c; y.mp_S_b = u.mp_S_b; y.ho_S_b = u.ho_S_b;
    y.wd_ec_c = wd_ec_c; y.vd_ec_c = vd_ec_c; y.g_c_c = g_c_c; y.wd_eb_b = wd_eb_b; y.vd_eb_b = vd_eb_b;
    return st;
}

// =============================================================================================
// control.jl:11-88 : continuous PI compensator with anti-windup (per channel)
struct PIParams {
    double k_p = 1, k_i = 0, k_l = 0, beta_p = 1;
    double bound_lo = -INFINITY, bound_hi = INFINITY;
};
struct PIOut {
    double u_p = 0, u_i = 0, y_p = 0, y_i = 0, out_free = 0, output = 0;
    int sat_out = 0;
    bool int_halted = false;
};
inline double sign_d(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }
// control.jl:52-81
inline double pi_f_ode(const PIParams& p, double input, int sat_ext, double x_i, PIOut& y) {
    y.u_p = p.beta_p * input;
    y.u_i = input;
    y.y_p = p.k_p * y.u_p;
    y.y_i = x_i;
    y.out_free = y.y_p + y.y_i;
    y.output = std::clamp(y.out_free, p.bound_lo, p.bound_hi);
    const int sat_hi = y.out_free >= p.bound_hi;
    const int sat_lo = y.out_free <= p.bound_lo;
    y.sat_out = sat_hi - sat_lo;
    y.int_halted = (sign_d(y.u_i * y.sat_out) > 0) || (sign_d(y.u_i * sat_ext) > 0);
    return p.k_i * y.u_i * (y.int_halted ? 0.0 : 1.0) - p.k_l * x_i;
}

}  // namespace fo
