// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's Robot2D (self-balancing two-wheel robot) with its discrete controller.
// Follows lib/FlightApps/src/robot2d/robot2d.jl:20-92 (vehicle), :208-228 (init), :349-449 (controller),
//         :526-570 (root model) ; lib/FlightPhysics/src/control.jl:431-471 (PID), :708-743 (LQR)
//         lib/FlightCore/src/sim.jl:204-218,318-381 (callback order: step, periodic at k Δt)
// State record r[10] per robot: [ω, v, θ, η | u_m, lqr_int_out, lqr_out_sat, pid_x_i, pid_x_d, pid_sat_out]
// Inputs u[4]: [mode (0 motor, 1 velocity, 2 position), m_ref, v_ref, η_ref]
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>

namespace fo {

struct R2Vehicle {  // robot2d.jl:20-30
    double L = 0.15, R = 0.05, m_b = 1.0, m_r = 0.1, J_b = -1, J_r = -1, k_m = 0.32, b_m = 0.0189, J_m = 0.0014;
    void finish() {
        if (J_b < 0) J_b = 1.0 / 12 * m_b * ((2 * L) * (2 * L));
        if (J_r < 0) J_r = 1.0 / 2 * m_r * (R * R);
    }
};
struct R2Gains {  // LQRDataPoint from robot2d.h5 (robot2d.jl:419-422) + PID gains (:430-436)
    double K_fbk[3] = {0, 0, 0}, K_fwd = 0, K_int = 0, x_trim[3] = {0, 0, 0}, u_trim = 0, z_trim = 0;
    double pid_kp = 0.6, pid_ki = 0.0, pid_kd = 0.0, pid_tau_f = 0.01;
};
constexpr double R2_G = 9.80665;
enum { R2_ST_LOST_BALANCE = 32 };

// robot2d.jl:50-92
inline void r2_f_ode(const R2Vehicle& p, const double* x, double u_m, double* xd, double* tau_m = nullptr) {
    const double w = x[0], v = x[1], th = x[2];
    const double w_m = v / p.R - w;
    const double tau_ss = p.k_m * u_m - p.b_m * w_m;
    const double s = std::sin(th), c = std::cos(th);
    const double M11 = p.m_b * (p.L * p.L) + p.J_b + p.J_m;
    const double M22 = p.m_b + p.m_r + (p.J_r + p.J_m) / (p.R * p.R);
    const double M12 = p.m_b * p.L * c - p.J_m / p.R;
    const double b1 = -tau_ss + p.m_b * p.L * R2_G * s;
    const double b2 = tau_ss / p.R + p.m_b * p.L * (w * w) * s;
    // 2x2 solve (StaticArrays closed form)
    const double det = M11 * M22 - M12 * M12;
    const double wd = (M22 * b1 - M12 * b2) / det;
    const double vd = (M11 * b2 - M12 * b1) / det;
    xd[0] = wd; xd[1] = vd; xd[2] = w; xd[3] = v;
    if (tau_m) *tau_m = tau_ss - p.J_m * (vd / p.R - wd);
}
inline double r2_sign(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }
// Controller.f_periodic! (robot2d.jl:379-407) with PID (control.jl:431-471) and LQR{3,1,1} (control.jl:708-743)
inline void r2_f_periodic(const R2Vehicle& p, const R2Gains& g, double dT, const double* u, double* r) {
    const double v_lim = 0.4 * (p.k_m * p.R / p.b_m);  // robot2d.jl:415-416
    const int mode = (int)u[0];
    double m_cmd = std::clamp(u[1], -1.0, 1.0);
    double v_ref = u[2];
    const double w = r[0], v = r[1], th = r[2], eta = r[3];
    if (mode == 2) {
        const double input = u[3] - eta;
        const double alpha = 1 / (g.pid_tau_f + dT);
        const double sat0 = r[9];
        const bool halted = r2_sign(input * sat0) > 0;  // sat_ext = 0
        const double x_i = r[7] + dT * g.pid_ki * input * (halted ? 0.0 : 1.0);
        const double x_d = alpha * g.pid_tau_f * r[8] + dT * alpha * g.pid_kd * input;
        const double out_free = g.pid_kp * input + x_i + alpha * (-r[8] + g.pid_kd * input);
        const double sat = (out_free >= v_lim ? 1.0 : 0.0) - (out_free <= -v_lim ? 1.0 : 0.0);
        v_ref = std::clamp(out_free, -v_lim, v_lim);
        r[7] = x_i; r[8] = x_d; r[9] = sat;
    }
    if (mode == 1 || mode == 2) {
        const double z_ref = std::clamp(v_ref, -v_lim, v_lim);
        const double int_in = g.K_int * (z_ref - v);
        const bool halted = r2_sign(int_in * r[6]) > 0;
        const double int_out = r[5] + dT * int_in * (halted ? 0.0 : 1.0);
        const double fbk = g.K_fbk[0] * (w - g.x_trim[0]) + g.K_fbk[1] * (v - g.x_trim[1]) + g.K_fbk[2] * (th - g.x_trim[2]);
        const double out_free = g.u_trim + int_out + g.K_fwd * (z_ref - g.z_trim) - fbk;
        r[6] = (out_free >= 1.0 ? 1.0 : 0.0) - (out_free <= -1.0 ? 1.0 : 0.0);
        r[5] = int_out;
        m_cmd = std::clamp(out_free, -1.0, 1.0);
    }
    r[4] = std::clamp(m_cmd, -1.0, 1.0);  // vehicle.u[] = controller.y.m_cmd (Ranged [-1,1]) (robot2d.jl:547-549)
}
// f_init!(robot, InitParameters(u_m, ω, η)) (robot2d.jl:214-228, 563-570): θ = 0, v = (ω + k_m u_m / b_m) R; controller reset
inline void r2_init(const R2Vehicle& p, double u_m, double w, double eta, double* r) {
    r[0] = w; r[1] = (w + (p.k_m * u_m) / p.b_m) * p.R; r[2] = 0; r[3] = eta;
    r[4] = std::clamp(u_m, -1.0, 1.0);
    for (int k = 5; k < 10; k++) r[k] = 0;
}
// nsteps x step!(sim): RK4, then f_step! (LostBalance), then f_periodic! when the step count hits a multiple of ratio.
// step0 = number of steps already taken since init (the periodic counter must persist across calls).
// LostBalance is thrown by f_step! (cb_step) right after the RK update of step k: the simulation ends with x = x_k, before that step's
// f_periodic! (sim.jl:204-218, 561-570); *term_k = the number of RK updates completed (step0 + k).
inline int32_t r2_step(const R2Vehicle& p, const R2Gains& g, double dt, int ratio, bool with_controller, const double* u, double* r,
                       int64_t step0, int64_t nsteps, int64_t* term_k = nullptr) {
    int32_t st = 0;
    for (int64_t k = 1; k <= nsteps && st == 0; k++) {
        double k1[4], k2[4], k3[4], k4[4], xt[4];
        r2_f_ode(p, r, r[4], k1);
        for (int i = 0; i < 4; i++) xt[i] = r[i] + dt / 2 * k1[i];
        r2_f_ode(p, xt, r[4], k2);
        for (int i = 0; i < 4; i++) xt[i] = r[i] + dt / 2 * k2[i];
        r2_f_ode(p, xt, r[4], k3);
        for (int i = 0; i < 4; i++) xt[i] = r[i] + dt * k3[i];
        r2_f_ode(p, xt, r[4], k4);
        for (int i = 0; i < 4; i++) r[i] = r[i] + (dt / 6) * (2 * (k2[i] + k3[i]) + (k1[i] + k4[i]));
        if (with_controller) {
            if (std::fabs(r[2]) > 45 * (3.14159265358979323846 / 180)) { st |= R2_ST_LOST_BALANCE; if (term_k) *term_k = step0 + k; }  // robot2d.jl:553-561
            if (st == 0 && ((step0 + k) % ratio) == 0) r2_f_periodic(p, g, dt * ratio, u, r);
        }
    }
    return st;
}

}  // namespace fo
