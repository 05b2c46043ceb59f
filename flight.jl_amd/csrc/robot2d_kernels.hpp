// robot2d_kernels.hpp — Robot2D (self-balancing two-wheel robot + discrete LQR/PID controller) on gfx950.
//
// Reference: lib/FlightApps/src/robot2d/robot2d.jl:20-92 (Vehicle.f_ode!), :208-228 (f_init!), :349-449 (Controller),
// :526-570 (Robot f_ode!/f_periodic!/f_step!); lib/FlightPhysics/src/control.jl:431-471 (PID), :708-743 (LQR{3,1,1}).
// One lane = one robot. The model is ~500x lighter than the C172 RHS (BASELINE.json config 5 uses it as the
// divergent-model stressor): 10 values per robot live in registers, occupancy is not register-limited, so the
// kernel is templated on the real type (fp64 like the reference, or fp32 for config 5) and launched on its own
// stream next to the C172 batch rather than branching per lane inside one kernel.
//
// Record r[10] (SoA, [10 x n]): [ω, v, θ, η | u_m, lqr_int_out, lqr_out_sat, pid_x_i, pid_x_d, pid_sat_out]
// Inputs u[4]: [mode (0 motor / 1 velocity / 2 position), m_ref, v_ref, η_ref]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/flightbatch.h"

namespace fbr {

#define FBR __device__ __forceinline__

template <class T>
struct R2Params {   // host-prepared, wave-uniform
    T L, R, m_b, m_r, J_b, J_r, k_m, b_m, J_m;
    T K_fbk[3], K_fwd, K_int, x_trim[3], u_trim, z_trim;
    T pid_kp, pid_ki, pid_kd, pid_tau_f;
    T v_lim;        // 0.4 k_m R / b_m (robot2d.jl:415-416)
};

template <class T> FBR T clampT(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <class T> FBR T signT(T v) { return v > T(0) ? T(1) : (v < T(0) ? T(-1) : T(0)); }

// Vehicle.f_ode! (robot2d.jl:50-92)
template <class T>
FBR void r2_f_ode(const R2Params<T>& p, const T (&x)[4], T u_m, T (&xd)[4], T& tau_m) {
    constexpr T g = T(9.80665);
    const T w = x[0], v = x[1], th = x[2];
    const T w_m = v / p.R - w;
    const T tau_ss = p.k_m * u_m - p.b_m * w_m;
    T s, c;
    if constexpr (sizeof(T) == 8) sincos(th, &s, &c); else sincosf(th, &s, &c);
    const T M11 = p.m_b * (p.L * p.L) + p.J_b + p.J_m;
    const T M22 = p.m_b + p.m_r + (p.J_r + p.J_m) / (p.R * p.R);
    const T M12 = p.m_b * p.L * c - p.J_m / p.R;
    const T b1 = -tau_ss + p.m_b * p.L * g * s;
    const T b2 = tau_ss / p.R + p.m_b * p.L * (w * w) * s;
    const T det = M11 * M22 - M12 * M12;
    const T wd = (M22 * b1 - M12 * b2) / det;
    const T vd = (M11 * b2 - M12 * b1) / det;
    xd[0] = wd; xd[1] = vd; xd[2] = w; xd[3] = v;
    tau_m = tau_ss - p.J_m * (vd / p.R - wd);
}

// Controller.f_periodic! + vehicle.u[] = m_cmd (robot2d.jl:379-407, 544-550)
template <class T>
FBR void r2_f_periodic(const R2Params<T>& p, T dT, const T (&u)[4], T (&r)[10]) {
    const int mode = (int)u[0];
    T m_cmd = clampT(u[1], T(-1), T(1));
    T v_ref = u[2];
    const T w = r[0], v = r[1], th = r[2], eta = r[3];
    if (mode == 2) {  // PID position loop (control.jl:431-471), bounds ±v_lim
        const T input = u[3] - eta;
        const T alpha = T(1) / (p.pid_tau_f + dT);
        const bool halted = signT(input * r[9]) > T(0);
        const T x_i = r[7] + (halted ? T(0) : dT * p.pid_ki * input);
        const T x_d = alpha * p.pid_tau_f * r[8] + dT * alpha * p.pid_kd * input;
        const T out_free = p.pid_kp * input + x_i + alpha * (-r[8] + p.pid_kd * input);
        r[9] = (out_free >= p.v_lim ? T(1) : T(0)) - (out_free <= -p.v_lim ? T(1) : T(0));
        v_ref = clampT(out_free, -p.v_lim, p.v_lim);
        r[7] = x_i; r[8] = x_d;
    }
    if (mode == 1 || mode == 2) {  // LQR{3,1,1} velocity loop (control.jl:708-743), bounds ±1
        const T z_ref = clampT(v_ref, -p.v_lim, p.v_lim);
        const T int_in = p.K_int * (z_ref - v);
        const bool halted = signT(int_in * r[6]) > T(0);
        const T int_out = r[5] + (halted ? T(0) : dT * int_in);
        const T fbk = p.K_fbk[0] * (w - p.x_trim[0]) + p.K_fbk[1] * (v - p.x_trim[1]) + p.K_fbk[2] * (th - p.x_trim[2]);
        const T out_free = p.u_trim + int_out + p.K_fwd * (z_ref - p.z_trim) - fbk;
        r[6] = (out_free >= T(1) ? T(1) : T(0)) - (out_free <= T(-1) ? T(1) : T(0));
        r[5] = int_out;
        m_cmd = clampT(out_free, T(-1), T(1));
    }
    r[4] = clampT(m_cmd, T(-1), T(1));
}

template <class T>
struct R2Args {
    T* r;              // [10 x n]
    const T* u;        // [4 x n]
    int32_t* status;   // [n]
    long long* term_step;   // [n] termination record (fb_get_termination)
    int32_t* term_where;    // [n]
    int64_t n;
    R2Params<T> p;
    T dt;
    int ratio;         // Δt / dt
    int with_controller;
};

// nsteps x step!(sim): RK4 (OrdinaryDiffEq stage order) -> f_step! (LostBalance, robot2d.jl:553-561) -> f_periodic! at k Δt.
// LostBalance is thrown out of cb_step right after the RK update of step k: the robot's simulation ends with x = x_k, before that
// step's f_periodic! (FC/sim.jl:204-218, 561-570); the termination record says so (FB_TERM_F_STEP, k updates complete).
template <class T>
__global__ __launch_bounds__(256) void k_r2_step(R2Args<T> a, long long step0, int nsteps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != 0) return;
    T r[10], u[4];
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = a.r[(int64_t)k * a.n + i];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = a.u[(int64_t)k * a.n + i];
    const T dt = a.dt, hdt = a.dt / 2, dt6 = a.dt / 6;
    int32_t st = 0;
#pragma unroll 1
    for (int k = 1; k <= nsteps; k++) {
        T x[4] = {r[0], r[1], r[2], r[3]}, xt[4], k1[4], k2[4], k3[4], k4[4], tm;
        r2_f_ode(a.p, x, r[4], k1, tm);
#pragma unroll
        for (int j = 0; j < 4; j++) xt[j] = x[j] + hdt * k1[j];
        r2_f_ode(a.p, xt, r[4], k2, tm);
#pragma unroll
        for (int j = 0; j < 4; j++) xt[j] = x[j] + hdt * k2[j];
        r2_f_ode(a.p, xt, r[4], k3, tm);
#pragma unroll
        for (int j = 0; j < 4; j++) xt[j] = x[j] + dt * k3[j];
        r2_f_ode(a.p, xt, r[4], k4, tm);
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = x[j] + dt6 * (2 * (k2[j] + k3[j]) + (k1[j] + k4[j]));
        if (a.with_controller) {
            if (fabs((double)r[2]) > 45 * (3.14159265358979323846 / 180)) {
                st |= FB_ST_LOST_BALANCE;
                a.term_where[i] = FB_TERM_F_STEP; a.term_step[i] = step0 + k;
                break;
            }
            if (((step0 + k) % a.ratio) == 0) r2_f_periodic(a.p, (T)(a.dt * a.ratio), u, r);
        }
    }
#pragma unroll
    for (int k = 0; k < 10; k++) a.r[(int64_t)k * a.n + i] = r[k];
    if (st) a.status[i] = st;
}

// f_ode!(robot): xdot[4 x n] and VehicleY y[8 x n] = ω v θ η u_m τ_m ω_dot v_dot (robot2d.jl:32-41)
template <class T>
__global__ __launch_bounds__(256) void k_r2_f_ode(R2Args<T> a, T* xdot, T* y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    T x[4], xd[4], tm;
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = a.r[(int64_t)k * a.n + i];
    const T u_m = a.r[(int64_t)4 * a.n + i];
    r2_f_ode(a.p, x, u_m, xd, tm);
    if (xdot) {
#pragma unroll
        for (int k = 0; k < 4; k++) xdot[(int64_t)k * a.n + i] = xd[k];
    }
    if (y) {
        const T yy[8] = {x[0], x[1], x[2], x[3], u_m, tm, xd[0], xd[1]};
#pragma unroll
        for (int k = 0; k < 8; k++) y[(int64_t)k * a.n + i] = yy[k];
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_r2_f_periodic(R2Args<T> a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    T r[10], u[4];
#pragma unroll
    for (int k = 0; k < 10; k++) r[k] = a.r[(int64_t)k * a.n + i];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = a.u[(int64_t)k * a.n + i];
    r2_f_periodic(a.p, (T)(a.dt * a.ratio), u, r);
#pragma unroll
    for (int k = 4; k < 10; k++) a.r[(int64_t)k * a.n + i] = r[k];
}
template <class T>
__global__ __launch_bounds__(256) void k_r2_f_step(R2Args<T> a, long long step0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (fabs((double)a.r[(int64_t)2 * a.n + i]) > 45 * (3.14159265358979323846 / 180)) {
        if ((a.status[i] & ~FB_ST_NAN) == 0) { a.term_where[i] = FB_TERM_OUTSIDE_STEP; a.term_step[i] = step0; }   // (the verb, not fb_step: see fb_get_termination)
        a.status[i] |= FB_ST_LOST_BALANCE;
    }
}
// f_init!(robot, InitParameters(u_m, ω, η)) (robot2d.jl:214-228, 563-570)
template <class T>
__global__ __launch_bounds__(256) void k_r2_init(R2Args<T> a, const T* ip /*[3 x n]*/) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const T u_m = ip[i], w = ip[a.n + i], eta = ip[2 * a.n + i];
    const T r[10] = {w, (w + (a.p.k_m * u_m) / a.p.b_m) * a.p.R, T(0), eta, clampT(u_m, T(-1), T(1)), T(0), T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int k = 0; k < 10; k++) a.r[(int64_t)k * a.n + i] = r[k];
    a.status[i] = 0; a.term_where[i] = FB_TERM_NONE; a.term_step[i] = 0;   // (init! clears terminations, record included)
}

}  // namespace fbr
