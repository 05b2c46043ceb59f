// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's Cessna172Xv2: the C172 airframe with seven first-order fly-by-wire actuators and
// the gain-scheduled longitudinal / lateral control laws, run as a discrete system every Δt.
// Follows lib/FlightApps/src/c172/c172x/c172x.jl:19-52 (Actuator1), :112-143 (FlyByWireActuation, assign!), :222-281 (init),
//         lib/FlightApps/src/c172/c172x/c172x2.jl:18-60 (Avionics), control/c172x_ctl.jl:19-25,29-77 (modes),
//         :84-199 (LQR vectors), :203-447 (ControlLawsLon), :449-519 (assign!, f_init!), :727-790 (lateral modes/vectors),
//         :814-995 (ControlLawsLat), :996-1032 (f_init!), lib/FlightPhysics/src/control.jl:123-185 (Integrator),
//         :370-471 (PID), :620-743 (LQR), :950-994 (lookups), lib/FlightPhysics/src/aircraftbase.jl:221-265 (call order).
// State: x[34] = the 27 states of Cessna172Sv0 in fo_c172.hpp order followed by the actuator positions
// (throttle, aileron, elevator, rudder, flaps, brake_left, brake_right); the C ABI permutes rows to the reference order.
// Controller inputs cu[FB_NCU], controller record cs[FB_NCS]: include/flightbatch.h.
#pragma once
#include "fo_c172.hpp"
#include "../include/flightbatch.h"
#include <limits>

namespace fo {

constexpr int NXX = 34, X_ACT = 27;
constexpr double ACT_TAU = 1.0 / 20;  // c172x.jl:21
constexpr double ACT_LO[7] = {0, -1, -1, -1, 0, 0, 0}, ACT_HI[7] = {1, 1, 1, 1, 1, 1, 1};  // c172x.jl:113-119
inline double rngd(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }
inline double sgn(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }

// ---- gain lookups (control.jl:950-994): linear in (EAS, h), Flat extrapolation ------------------------------
struct CtlGains {
    const double* lk[10] = {};  // te2te tv2te vh2te q2e c2θ v2t ar2ar φβ2ar p2φ χ2φ
    void bind(const double* blob) {
        const int rec[10] = {FB_CTL_LQR8_REC, FB_CTL_LQR8_REC, FB_CTL_LQR9_REC, FB_CTL_PID_REC, FB_CTL_PID_REC, FB_CTL_PID_REC,
                             FB_CTL_LQR8_REC, FB_CTL_LQR8_REC, FB_CTL_PID_REC, FB_CTL_PID_REC};
        const double* p = blob;
        for (int k = 0; k < 10; k++) { lk[k] = p; p += FB_CTL_GRID_HDR + (int)p[0] * (int)p[1] * rec[k]; }
    }
};
inline void ctl_lookup(const double* lk, int rec, double EAS, double h, double* out) {
    const int nE = (int)lk[0], nH = (int)lk[1];
    const GridLoc l1 = nE > 1 ? range_locate(lk[2], lk[3], nE, EAS, FLAT, FLAT) : GridLoc{0, 0.0};   // NoInterp on singleton dims (control.jl:957-961)
    const GridLoc l2 = nH > 1 ? range_locate(lk[4], lk[5], nH, h, FLAT, FLAT) : GridLoc{0, 0.0};
    const int i1 = nE > 1 ? l1.i + 1 : 0, j1 = nH > 1 ? l2.i + 1 : 0;
    const double* d = lk + FB_CTL_GRID_HDR;
    const double* a00 = d + (size_t)(l1.i + nE * l2.i) * rec;
    const double* a10 = d + (size_t)(i1 + nE * l2.i) * rec;
    const double* a01 = d + (size_t)(l1.i + nE * j1) * rec;
    const double* a11 = d + (size_t)(i1 + nE * j1) * rec;
    for (int c = 0; c < rec; c++)
        out[c] = (1 - l1.w) * ((1 - l2.w) * a00[c] + l2.w * a01[c]) + l1.w * ((1 - l2.w) * a10[c] + l2.w * a11[c]);
}

// ---- compensators --------------------------------------------------------------------------------------
constexpr double INF = std::numeric_limits<double>::infinity();
struct PidP { double k_p = 1, k_i = 0, k_d = 0, tau_f = 0.01, lo = -INF, hi = INF; };   // β_p = β_d = 1 (control.jl:373-375)
// PID f_periodic! (control.jl:431-471); s = {x_i0, x_d0, sat_out_0}
inline double pid_run(const PidP& P, double dT, double input, double sat_ext, double* s) {
    const double a = 1 / (P.tau_f + dT);
    const double u_p = 1.0 * input, u_d = 1.0 * input, u_i = input;
    const bool halted = (sgn(u_i * s[2]) > 0) || (sgn(u_i * sat_ext) > 0);
    const double x_i = s[0] + dT * P.k_i * u_i * (halted ? 0.0 : 1.0);
    const double x_d = a * P.tau_f * s[1] + dT * a * P.k_d * u_d;
    const double y_p = P.k_p * u_p, y_i = x_i, y_d = a * (-s[1] + P.k_d * u_d);
    const double out_free = y_p + y_i + y_d;
    const double sat = (out_free >= P.hi ? 1.0 : 0.0) - (out_free <= P.lo ? 1.0 : 0.0);
    s[0] = x_i; s[1] = x_d; s[2] = sat;
    return rngd(out_free, P.lo, P.hi);
}
inline void pid_init(const PidP& P, double dT, double* s) { s[0] = s[1] = s[2] = 0; pid_run(P, dT, 0, 0, s); }  // control.jl:420-426
// Integrator f_periodic! (control.jl:161-183), unbounded; s = {x0, sat_out_0}
inline double integ_run(double dT, double input, double sat_ext, double* s) {
    const bool halted = (sgn(input * s[1]) > 0) || (sgn(input * sat_ext) > 0);
    const double x1 = s[0] + dT * input * (halted ? 0.0 : 1.0);
    s[0] = x1;
    s[1] = (x1 >= INF ? 1.0 : 0.0) - (x1 <= -INF ? 1.0 : 0.0);
    return rngd(x1, -INF, INF);
}
inline void integ_init(double dT, double* s) { s[0] = s[1] = 0; integ_run(dT, 0, 0, s); }
// LQR{NX,2,2} f_periodic! (control.jl:708-743). g = interpolated record [K_fbk 2xNX | K_fwd 2x2 | K_int 2x2 | x_trim | u_trim | z_trim];
// s = {int_out_0[2], out_sat_0[2]}; sat_ext is never set by the control laws (stays 0).
template <int NX>
inline void lqr_run(const double* g, const double* lo, const double* hi, double dT, const double* x, const double* z, const double* z_ref,
                    double* s, double* out) {
    const double* K_fbk = g; const double* K_fwd = g + 2 * NX; const double* K_int = K_fwd + 4;
    const double* x_trim = K_int + 4; const double* u_trim = x_trim + NX; const double* z_trim = u_trim + 2;
    const double dz[2] = {z_ref[0] - z[0], z_ref[1] - z[1]}, dzt[2] = {z_ref[0] - z_trim[0], z_ref[1] - z_trim[1]};
    double dx[NX];
    for (int k = 0; k < NX; k++) dx[k] = x[k] - x_trim[k];
    for (int i = 0; i < 2; i++) {
        const double int_in = K_int[i] * dz[0] + K_int[i + 2] * dz[1];
        const bool halted = (sgn(int_in * s[2 + i]) > 0) || (sgn(int_in * 0.0) > 0);
        const double int_out = s[i] + dT * int_in * (halted ? 0.0 : 1.0);
        const double fwd = K_fwd[i] * dzt[0] + K_fwd[i + 2] * dzt[1];
        double fbk = K_fbk[i] * dx[0];
        for (int k = 1; k < NX; k++) fbk += K_fbk[i + 2 * k] * dx[k];
        const double out_free = u_trim[i] + int_out + fwd - fbk;
        s[i] = int_out;
        s[2 + i] = (out_free >= hi[i] ? 1.0 : 0.0) - (out_free <= lo[i] ? 1.0 : 0.0);
        out[i] = rngd(out_free, lo[i], hi[i]);
    }
}
template <int NX>
inline void lqr_init(const double* g, const double* lo, const double* hi, double dT, double* s) {  // control.jl:693-701
    double x[NX] = {}, z[2] = {0, 0}, out[2];
    s[0] = s[1] = s[2] = s[3] = 0;
    lqr_run<NX>(g, lo, hi, dT, x, z, z, s, out);
}

// ---- what the control laws read from vehicle.y ---------------------------------------------------------------
struct CtlIn {
    double EAS = 0, h_e = 0, theta = 0, phi = 0, clm = 0, chi = 0;
    LatLon ll;  // kinematics.y.ϕ_λ (guidance)
    V3 w_wb_b, w_eb_b;
    double alpha = 0, beta = 0, alpha_filt = 0, beta_filt = 0, n_eng = 0;
    double pos[4] = {0, 0, 0, 0};  // throttle, aileron, elevator, rudder actuator positions (Ranged)
    double cmd[4] = {0, 0, 0, 0};  // idem, commands seen by the last f_ode!
    bool on_gnd = false;
};
inline CtlIn ctl_in_from(const C172Model& M, const C172Y& y, const double* x, const double* cmd4) {
    CtlIn c;
    c.EAS = y.air.EAS; c.h_e = y.kin.h_e; c.theta = y.kin.e_nb.theta; c.phi = y.kin.e_nb.phi; c.clm = -y.kin.v_eb_n.z;
    c.chi = y.kin.chi_gnd; c.w_wb_b = y.kin.w_wb_b; c.w_eb_b = y.kin.w_eb_b; c.ll = y.kin.ll;
    c.alpha = y.aero.alpha; c.beta = y.aero.beta; c.alpha_filt = x[X_AFILT]; c.beta_filt = x[X_AFILT + 1];
    c.n_eng = x[X_ENG] / M.eng.w_rated;
    for (int k = 0; k < 4; k++) { c.pos[k] = rngd(x[X_ACT + k], ACT_LO[k], ACT_HI[k]); c.cmd[k] = rngd(cmd4[k], ACT_LO[k], ACT_HI[k]); }
    c.on_gnd = y.ldg[0].strut.wow || y.ldg[1].strut.wow || y.ldg[2].strut.wow;  // c172.jl:998-1001
    return c;
}

// ---- ControlLawsLon f_periodic! (c172x_ctl.jl:293-447) ----------------------------------------------------------
constexpr double LON_LO[2] = {0, -1}, LON_HI[2] = {1, 1}, LAT_LO[2] = {-1, -1}, LAT_HI[2] = {1, 1};
inline void pid_gains(const CtlGains& G, int which, double EAS, double h, PidP& P) {
    double g[4];
    ctl_lookup(G.lk[which], FB_CTL_PID_REC, EAS, h, g);
    P.k_p = g[0]; P.k_i = g[1]; P.k_d = g[2]; P.tau_f = g[3];
}
inline void ctl_lon_periodic(const CtlGains& G, double dT, const CtlIn& v, const double* cu, double* cs) {
    const int mode_req = (int)cu[FB_CU_LON_MODE_REQ];
    double q_ref = cu[FB_CU_Q_REF], theta_ref = cu[FB_CU_THETA_REF];
    const double EAS_ref = cu[FB_CU_EAS_REF], clm_ref = cu[FB_CU_CLM_REF], h_ref = cu[FB_CU_H_REF];
    const double EAS = v.EAS, h_e = v.h_e, q = v.w_wb_b.y, r = v.w_wb_b.z, theta = v.theta, phi = v.phi, clm = v.clm;
    const double h_err = h_ref - h_e;
    const int h_state = (int)cs[FB_CS_H_STATE], mode_prev = (int)cs[FB_CS_LON_MODE];
    double throttle_ref = rngd(rngd(cu[FB_CU_THROTTLE_AXIS], 0, 1) + rngd(cu[FB_CU_THROTTLE_OFFSET], 0, 1), 0, 1);
    double elevator_ref = rngd(rngd(cu[FB_CU_ELEVATOR_AXIS], -1, 1) + rngd(cu[FB_CU_ELEVATOR_OFFSET], -1, 1), -1, 1);
    double throttle_cmd = throttle_ref, elevator_cmd = elevator_ref;
    const double h_thr = 10.0, h_hys = 1.0, k_p_theta = 1.0;
    int mode;
    if (v.on_gnd) mode = FB_LON_DIRECT;
    else if (mode_req == FB_LON_EAS_ALT) {
        if (h_state == FB_ALT_ACQUIRE) {
            mode = FB_LON_THR_EAS;
            throttle_ref = h_err > 0 ? 1.0 : 0.0;
            if (std::fabs(h_err) < h_thr - h_hys) cs[FB_CS_H_STATE] = FB_ALT_HOLD;
        } else {
            mode = FB_LON_EAS_ALT;
            if (std::fabs(h_err) > h_thr + h_hys) cs[FB_CS_H_STATE] = FB_ALT_ACQUIRE;
        }
    } else mode = mode_req;
    const bool te2te = mode == FB_LON_SAS || mode == FB_LON_THR_Q || mode == FB_LON_THR_THETA || mode == FB_LON_EAS_Q ||
                       mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    const bool q2e = te2te && mode != FB_LON_SAS;
    const bool th2q = mode == FB_LON_THR_THETA || mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    const bool v2t = mode == FB_LON_EAS_Q || mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    const bool c2th = mode == FB_LON_EAS_CLM;
    const double x_red[8] = {v.w_eb_b.y, theta, EAS, v.alpha, v.alpha_filt, v.n_eng, v.pos[0], v.pos[2]};
    double g[FB_CTL_LQR9_REC], out[2];
    PidP P;
    if (te2te) {
        double* te = cs + FB_CS_TE2TE;
        const double sat_thr = te[2], sat_ele = te[3];  // te2te_lqr.y.out_sat of the previous update
        if (v2t) {
            pid_gains(G, 5, EAS, h_e, P);
            double* s = cs + FB_CS_V2T_PID;
            if (mode != mode_prev) { pid_init(P, dT, s); if (P.k_i != 0) s[0] = cs[FB_CS_THROTTLE_CMD]; }
            throttle_ref = pid_run(P, dT, EAS_ref - EAS, sat_thr, s);
        }
        if (q2e) {
            pid_gains(G, 3, EAS, h_e, P);
            double* si = cs + FB_CS_Q2E_INT; double* sp = cs + FB_CS_Q2E_PID;
            if (mode != mode_prev) { integ_init(dT, si); pid_init(P, dT, sp); if (P.k_i != 0) sp[0] = te[5]; }
            if (th2q) {
                if (c2th) {
                    PidP Pc;
                    pid_gains(G, 4, EAS, h_e, Pc);
                    double* sc = cs + FB_CS_C2THETA_PID;
                    if (mode != mode_prev) { pid_init(Pc, dT, sc); if (Pc.k_i != 0) sc[0] = theta; }
                    theta_ref = pid_run(Pc, dT, clm_ref - clm, sat_ele, sc);
                }
                const double theta_dot_ref = k_p_theta * (theta_ref - theta);
                const double phi_bnd = rngd(phi, -PI / 3, PI / 3);
                q_ref = 1 / std::cos(phi_bnd) * theta_dot_ref + r * std::tan(phi_bnd);
            }
            const double io = integ_run(dT, q_ref - q, sat_ele, si);
            elevator_ref = pid_run(P, dT, io, sat_ele, sp);
        }
        ctl_lookup(G.lk[0], FB_CTL_LQR8_REC, EAS, h_e, g);
        const double z[2] = {v.cmd[0], v.cmd[2]}, z_ref[2] = {throttle_ref, elevator_ref};
        te[4] = z_ref[0]; te[5] = z_ref[1];
        lqr_run<8>(g, LON_LO, LON_HI, dT, x_red, z, z_ref, te, out);
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    if (mode == FB_LON_THR_EAS) {
        ctl_lookup(G.lk[1], FB_CTL_LQR8_REC, EAS, h_e, g);
        double* s = cs + FB_CS_TV2TE;
        if (mode != mode_prev) lqr_init<8>(g, LON_LO, LON_HI, dT, s);
        const double z[2] = {v.cmd[0], EAS}, z_ref[2] = {throttle_ref, EAS_ref};
        lqr_run<8>(g, LON_LO, LON_HI, dT, x_red, z, z_ref, s, out);
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    if (mode == FB_LON_EAS_ALT) {
        ctl_lookup(G.lk[2], FB_CTL_LQR9_REC, EAS, h_e, g);
        double* s = cs + FB_CS_VH2TE;
        if (mode != mode_prev) lqr_init<9>(g, LON_LO, LON_HI, dT, s);
        const double x_full[9] = {v.w_eb_b.y, theta, EAS, v.alpha, h_e, v.alpha_filt, v.n_eng, v.pos[0], v.pos[2]};
        const double z[2] = {EAS, h_e}, z_ref[2] = {EAS_ref, h_ref};
        lqr_run<9>(g, LON_LO, LON_HI, dT, x_full, z, z_ref, s, out);
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    cs[FB_CS_LON_MODE] = mode;
    cs[FB_CS_THROTTLE_REF] = rngd(throttle_ref, 0, 1); cs[FB_CS_ELEVATOR_REF] = rngd(elevator_ref, -1, 1);
    cs[FB_CS_Q_REF] = q_ref; cs[FB_CS_THETA_REF] = theta_ref;
    cs[FB_CS_THROTTLE_CMD] = rngd(throttle_cmd, 0, 1); cs[FB_CS_ELEVATOR_CMD] = rngd(elevator_cmd, -1, 1);
}

// ---- ControlLawsLat f_periodic! (c172x_ctl.jl:883-979) ----------------------------------------------------------
inline void ctl_lat_periodic(const CtlGains& G, double dT, const CtlIn& v, const double* cu, double* cs) {
    const int mode_req = (int)cu[FB_CU_LAT_MODE_REQ];
    const double p_ref = cu[FB_CU_P_REF], beta_ref = cu[FB_CU_BETA_REF], chi_ref = cu[FB_CU_CHI_REF];
    double phi_ref = cu[FB_CU_PHI_REF];
    const double EAS = v.EAS, h_e = v.h_e, p = v.w_wb_b.x;
    const int mode_prev = (int)cs[FB_CS_LAT_MODE];
    const int mode = v.on_gnd ? (int)FB_LAT_DIRECT : mode_req;
    const double aileron_ref = rngd(rngd(cu[FB_CU_AILERON_AXIS], -1, 1) + rngd(cu[FB_CU_AILERON_OFFSET], -1, 1), -1, 1);
    const double rudder_ref = rngd(rngd(cu[FB_CU_RUDDER_AXIS], -1, 1) + rngd(cu[FB_CU_RUDDER_OFFSET], -1, 1), -1, 1);
    double aileron_cmd = aileron_ref, rudder_cmd = rudder_ref;
    const double x_lat[8] = {v.w_eb_b.x, v.w_eb_b.z, v.phi, EAS, v.beta, v.beta_filt, v.pos[1], v.pos[3]};
    double g[FB_CTL_LQR8_REC], out[2];
    if (mode == FB_LAT_SAS) {
        ctl_lookup(G.lk[6], FB_CTL_LQR8_REC, EAS, h_e, g);
        const double z[2] = {v.cmd[1], v.cmd[3]}, z_ref[2] = {aileron_ref, rudder_ref};
        lqr_run<8>(g, LAT_LO, LAT_HI, dT, x_lat, z, z_ref, cs + FB_CS_AR2AR, out);
        aileron_cmd = out[0]; rudder_cmd = out[1];
    }
    if (mode == FB_LAT_P_BETA || mode == FB_LAT_PHI_BETA || mode == FB_LAT_CHI_BETA) {
        double* pb = cs + FB_CS_PHIBETA2AR;
        const double sat_ail = pb[2];
        PidP P;
        if (mode == FB_LAT_P_BETA) {
            pid_gains(G, 8, EAS, h_e, P);
            double* si = cs + FB_CS_P2PHI_INT; double* sp = cs + FB_CS_P2PHI_PID;
            if (mode != mode_prev) { integ_init(dT, si); pid_init(P, dT, sp); if (P.k_i != 0) sp[0] = pb[4]; }
            const double io = integ_run(dT, p_ref - p, sat_ail, si);
            phi_ref = pid_run(P, dT, io, sat_ail, sp);
        } else if (mode == FB_LAT_CHI_BETA) {
            pid_gains(G, 9, EAS, h_e, P);
            P.lo = -PI / 4; P.hi = PI / 4;  // c172x_ctl.jl:875-876
            double* sp = cs + FB_CS_CHI2PHI_PID;
            if (mode != mode_prev) { pid_init(P, dT, sp); if (P.k_i != 0) sp[0] = pb[4]; }
            phi_ref = pid_run(P, dT, wrap_to_pi(chi_ref - v.chi), sat_ail, sp);
        }
        ctl_lookup(G.lk[7], FB_CTL_LQR8_REC, EAS, h_e, g);
        if (mode != mode_prev) lqr_init<8>(g, LAT_LO, LAT_HI, dT, pb);
        const double z[2] = {v.phi, v.beta}, z_ref[2] = {phi_ref, beta_ref};
        pb[4] = z_ref[0]; pb[5] = z_ref[1];
        lqr_run<8>(g, LAT_LO, LAT_HI, dT, x_lat, z, z_ref, pb, out);
        aileron_cmd = out[0]; rudder_cmd = out[1];
    }
    cs[FB_CS_LAT_MODE] = mode;
    cs[FB_CS_AILERON_REF] = aileron_ref; cs[FB_CS_RUDDER_REF] = rudder_ref; cs[FB_CS_PHI_REF] = phi_ref;
    cs[FB_CS_AILERON_CMD] = rngd(aileron_cmd, -1, 1); cs[FB_CS_RUDDER_CMD] = rngd(rudder_cmd, -1, 1);
}
// ---- guidance (c172x/guidance/c172x_gdc.jl) -------------------------------------------------------------------------
struct GeoPoint { LatLon ll; double h = 0; };   // Geographic{LatLon, Ellipsoidal}
inline V3 cartesian_of(const GeoPoint& p) { return cartesian_from_geographic(nvector_from_latlon(p.ll), p.h); }
// Segment(p1; s, χ, Δh) (c172x_gdc.jl:56-83): the end point s metres along azimuth χ in the local-level frame of p1
inline GeoPoint segment_end(const GeoPoint& p1, double s, double chi, double dh) {
    const Quat q_en1 = ltf(nvector_from_latlon(p1.ll));
    const V3 r_12_e = rotate(q_en1, V3{s * std::cos(chi), s * std::sin(chi), 0.0});
    const V3 r_e2_e = cartesian_of(p1) + r_12_e;
    const GeoNE g2 = geographic_from_cartesian(r_e2_e);   // LatLon(r_e2_e): Cartesian -> NVector -> LatLon
    return {latlon_from_nvector(g2.n_e), p1.h + dh};
}
struct SegmentData { double chi_12 = 0, gamma_12 = 0, s_12 = 0, s_1b = 0, s_2b = 0, e_sb = 0, v_sb = 0, h_s = 0; };
// SegmentGuidanceData(seg, Ob) (c172x_gdc.jl:113-149)
inline SegmentData segment_data(const GeoPoint& p1, const GeoPoint& p2, const GeoPoint& Ob) {
    const V3 r_e1_e = cartesian_of(p1), r_e2_e = cartesian_of(p2), r_eb_e = cartesian_of(Ob);
    const Quat q_en = ltf(nvector_from_latlon(Ob.ll));
    const V3 r_1b_n = rotate(inv(q_en), r_eb_e - r_e1_e);
    const V3 r_1b_h = {r_1b_n.x, r_1b_n.y, 0.0};
    const V3 r_12_n = rotate(inv(q_en), r_e2_e - r_e1_e);
    const V3 r_12_h = {r_12_n.x, r_12_n.y, 0.0};
    SegmentData d;
    d.s_12 = norm(r_12_h);
    const V3 u_12 = r_12_h / d.s_12;
    d.s_1b = dot(u_12, r_1b_h);
    d.s_2b = d.s_1b - d.s_12;
    d.e_sb = cross(u_12, r_1b_h).z;
    d.h_s = p1.h + (p2.h - p1.h) * d.s_1b / d.s_12;
    d.v_sb = Ob.h - d.h_s;
    d.chi_12 = azimuth(u_12);
    d.gamma_12 = std::atan2(p2.h - p1.h, d.s_12);
    return d;
}
// GuidanceLaws f_periodic! (c172x_gdc.jl:297-329) with SegmentGuidance (:232-252). Writes the control laws' inputs cu.
inline void gdc_periodic(const CtlIn& v, double* cu, double* cs) {
    const int mode_req = (int)cu[FB_CU_GDC_MODE_REQ];
    const int mode = v.on_gnd ? (int)FB_GDC_DIRECT : mode_req;
    if (mode == FB_GDC_SEGMENT) {
        const GeoPoint p1 = {{cu[FB_CU_SEG_P1], cu[FB_CU_SEG_P1 + 1]}, cu[FB_CU_SEG_P1 + 2]};
        const GeoPoint p2 = {{cu[FB_CU_SEG_P2], cu[FB_CU_SEG_P2 + 1]}, cu[FB_CU_SEG_P2 + 2]};
        const SegmentData d = segment_data(p1, p2, GeoPoint{v.ll, v.h_e});
        const double dchi_inf = PI / 2, e_sf = 250.0, e_thr = 1000.0;   // c172x_gdc.jl:200-204
        const double dchi = -dchi_inf / (PI / 2) * std::atan(d.e_sb / e_sf);
        const double chi_ref = wrap_to_pi(d.chi_12 + dchi);
        const bool hor = cu[FB_CU_SEG_HOR_REQ] != 0;
        const bool vrt = std::fabs(d.e_sb) < e_thr ? (cu[FB_CU_SEG_VRT_REQ] != 0) : false;
        cs[FB_CS_SEG_DCHI] = dchi; cs[FB_CS_SEG_CHI_REF] = chi_ref; cs[FB_CS_SEG_H_REF] = d.h_s;
        cs[FB_CS_SEG_HOR_GDC] = hor; cs[FB_CS_SEG_VRT_GDC] = vrt; cs[FB_CS_SEG_E_SB] = d.e_sb; cs[FB_CS_SEG_S_1B] = d.s_1b; cs[FB_CS_SEG_S_2B] = d.s_2b;
        if (hor) { cu[FB_CU_CHI_REF] = chi_ref; cu[FB_CU_LAT_MODE_REQ] = FB_LAT_CHI_BETA; }
        if (vrt) { cu[FB_CU_H_REF] = d.h_s; cu[FB_CU_LON_MODE_REQ] = FB_LON_EAS_ALT; }
    }
    cs[FB_CS_GDC_MODE] = mode;
}
// Avionics f_periodic! (c172x2.jl:27-37): guidance (it may rewrite the control laws' inputs), then the control laws
inline void ctl_periodic(const CtlGains& G, double dT, const CtlIn& v, double* cu, double* cs) {
    gdc_periodic(v, cu, cs);
    ctl_lon_periodic(G, dT, v, cu, cs);
    ctl_lat_periodic(G, dT, v, cu, cs);
}

// f_init!(avionics, vehicle) (c172x2.jl:46-50; c172x_ctl.jl:463-519, 1000-1032) for a freshly built model:
// resets every compensator, aligns the inputs cu with the vehicle's current outputs, then runs the control laws once in
// each SAS-based mode so that the LQR trackers hold the trim point, and leaves both channels in `direct`.
inline void ctl_init(const CtlGains& G, double dT, const CtlIn& v, double beta, double* cu, double* cs) {
    for (int k = 0; k < FB_NCS; k++) cs[k] = 0;
    cs[FB_CS_H_STATE] = FB_ALT_HOLD;   // ControlLawsLonS default (c172x_ctl.jl:238-240)
    cu[FB_CU_THROTTLE_AXIS] = v.pos[0]; cu[FB_CU_ELEVATOR_AXIS] = v.pos[2]; cu[FB_CU_THROTTLE_OFFSET] = 0; cu[FB_CU_ELEVATOR_OFFSET] = 0;
    cu[FB_CU_Q_REF] = v.w_wb_b.y; cu[FB_CU_THETA_REF] = v.theta; cu[FB_CU_EAS_REF] = v.EAS; cu[FB_CU_CLM_REF] = v.clm; cu[FB_CU_H_REF] = v.h_e;
    const int lon_seq[4] = {FB_LON_SAS, FB_LON_THR_EAS, FB_LON_EAS_ALT, FB_LON_DIRECT};
    for (int m : lon_seq) { cu[FB_CU_LON_MODE_REQ] = m; ctl_lon_periodic(G, dT, v, cu, cs); }
    cu[FB_CU_AILERON_AXIS] = v.pos[1]; cu[FB_CU_RUDDER_AXIS] = v.pos[3]; cu[FB_CU_AILERON_OFFSET] = 0; cu[FB_CU_RUDDER_OFFSET] = 0;
    cu[FB_CU_P_REF] = v.w_wb_b.x; cu[FB_CU_PHI_REF] = v.phi; cu[FB_CU_BETA_REF] = beta; cu[FB_CU_CHI_REF] = v.chi;
    const int lat_seq[3] = {FB_LAT_SAS, FB_LAT_PHI_BETA, FB_LAT_DIRECT};
    for (int m : lat_seq) { cu[FB_CU_LAT_MODE_REQ] = m; ctl_lat_periodic(G, dT, v, cu, cs); }
    // guidance: GuidanceLawsU / SegmentGuidanceU defaults (c172x_gdc.jl:206-210, 281-283; Segment() :85)
    cu[FB_CU_GDC_MODE_REQ] = FB_GDC_DIRECT; cu[FB_CU_SEG_HOR_REQ] = 0; cu[FB_CU_SEG_VRT_REQ] = 0;
    cu[FB_CU_SEG_P1] = 0; cu[FB_CU_SEG_P1 + 1] = 0; cu[FB_CU_SEG_P1 + 2] = 0;
    cu[FB_CU_SEG_P2] = 1e-3; cu[FB_CU_SEG_P2 + 1] = 0; cu[FB_CU_SEG_P2 + 2] = 0;
}

// ---- vehicle with fly-by-wire actuation ------------------------------------------------------------------------
// actuator commands in force: the four control-law outputs (assign!, c172x_ctl.jl:449-458, 986-995) + flaps / brakes from u
inline void x2_commands(const C172Inputs& u, const double* cs, double* cmd7) {
    cmd7[0] = cs[FB_CS_THROTTLE_CMD]; cmd7[1] = cs[FB_CS_AILERON_CMD]; cmd7[2] = cs[FB_CS_ELEVATOR_CMD]; cmd7[3] = cs[FB_CS_RUDDER_CMD];
    cmd7[4] = u.flaps; cmd7[5] = u.brake_left; cmd7[6] = u.brake_right;
    for (int k = 0; k < 7; k++) cmd7[k] = rngd(cmd7[k], ACT_LO[k], ACT_HI[k]);
}
// f_ode!(aircraft) (aircraftbase.jl:221-230): Actuator1.f_ode! (c172x.jl:39-52), assign! (c172x.jl:127-143), then the C172 RHS
inline int32_t c172x_f_ode(const C172Model& M, const Env& env, const C172Inputs& u, const double* cmd7, const C172Disc& s,
                           const double* x, double* xdot, C172Y& y) {
    C172Inputs v = u;
    double pos[7];
    for (int k = 0; k < 7; k++) {
        pos[k] = rngd(x[X_ACT + k], ACT_LO[k], ACT_HI[k]);
        xdot[X_ACT + k] = 1 / ACT_TAU * (cmd7[k] - x[X_ACT + k]);
    }
    v.throttle = pos[0]; v.aileron = pos[1]; v.elevator = pos[2]; v.rudder = pos[3]; v.flaps = pos[4];
    v.brake_left = pos[5]; v.brake_right = pos[6];
    v.aileron_offset = v.elevator_offset = v.rudder_offset = 0;
    return c172_f_ode(M, env, v, s, x, xdot, y);
}
// One step!(sim): RK4 over the 34 states, evaluation at the new state, f_step!, then — when `periodic` — the control laws
// (they read the y of that last evaluation, aircraftbase.jl:232-242; sim.jl:204-218).
// Termination as in c172_step (fo_c172.hpp): the first exception ends the simulation with x = mdl.x at the throw; an exception out
// of f_step! (cb_step) precedes cb_periodic, so the control laws are not updated in that step.
inline int32_t c172x_step(const C172Model& M, const CtlGains& G, const Env& env, const C172Inputs& u, double* cu, double* cs,
                          C172Disc& s, double* x, double dt, double dT, bool periodic, C172Y& y, Term* term = nullptr) {
    ThrowScope throwing;
    double k1[NXX], k2[NXX], k3[NXX], k4[NXX], xt[NXX], cmd7[7];
    C172Y yt;
    x2_commands(u, cs, cmd7);
    int where = TERM_REEVAL;
    bool advanced = false;
    try {
        c172x_f_ode(M, env, u, cmd7, s, x, k1, yt);
        for (int i = 0; i < NXX; i++) xt[i] = x[i] + dt / 2 * k1[i];
        where = TERM_K2;
        c172x_f_ode(M, env, u, cmd7, s, xt, k2, yt);
        for (int i = 0; i < NXX; i++) xt[i] = x[i] + dt / 2 * k2[i];
        where = TERM_K3;
        c172x_f_ode(M, env, u, cmd7, s, xt, k3, yt);
        for (int i = 0; i < NXX; i++) xt[i] = x[i] + dt * k3[i];
        where = TERM_K4;
        c172x_f_ode(M, env, u, cmd7, s, xt, k4, yt);
        for (int i = 0; i < NXX; i++) x[i] = x[i] + (dt / 6) * (2 * (k2[i] + k3[i]) + (k1[i] + k4[i]));
        advanced = true;
        where = TERM_NEW;
        c172x_f_ode(M, env, u, cmd7, s, x, k1, y);
        const CtlIn v = ctl_in_from(M, y, x, cmd7);  // vehicle.y as the periodic update will see it (before f_step! touches x)
        C172Inputs uf = u;
        where = TERM_F_STEP;
        c172_f_step(M, uf, s, x, y);
        if (periodic) ctl_periodic(G, dT, v, cu, cs);
    } catch (const Termination& t) {
        if (where >= TERM_K2 && where <= TERM_K4)
            for (int i = 0; i < NXX; i++) x[i] = xt[i];
        if (term) { term->status = t.bit; term->where = where; term->advanced = advanced; }
        return t.bit;
    }
    return 0;
}
// f_init!(aircraft, trim) (aircraftbase.jl:255-265; c172x.jl:285-326): actuator states = commands = trim values, then ctl_init
inline bool c172x_trim_init(const C172Model& M, const CtlGains& G, const Env& env, const TrimParams& tp, TrimState& ts, double dT,
                            double* x, C172Inputs& u, C172Disc& s, double* cu, double* cs, double* cost = nullptr) {
    const bool ok = trim_solve(M, tp, env, ts, cost);
    trim_assign(M, tp, ts, env, x, u, s);
    const double cmd7[7] = {u.throttle, u.aileron, u.elevator, u.rudder, u.flaps, 0.0, 0.0};
    for (int k = 0; k < 7; k++) x[X_ACT + k] = cmd7[k];
    u.brake_left = u.brake_right = 0;
    double xd[NXX];
    C172Y y;
    c172x_f_ode(M, env, u, cmd7, s, x, xd, y);
    const CtlIn v = ctl_in_from(M, y, x, cmd7);
    ctl_init(G, dT, v, y.aero.beta, cu, cs);
    return ok;
}

}  // namespace fo
