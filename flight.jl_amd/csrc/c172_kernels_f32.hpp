// c172_kernels_f32.hpp — the fp32 airborne stepper of Cessna172Sv0 (FB_F32 handles; BASELINE.json configs[4] asks for an fp32 fleet).
//
// Same algorithm and call order as k_step<false, WA, false> (c172_kernels.hpp), instantiated from the same source over
// `float` (namespace fbf of c172_device.hpp). What fp32 buys on gfx950 is not a faster FMA — scalar fp32 and fp64 VALU
// instructions issue at the same rate — but half the registers and half the LDS per aircraft: the kernel fits TWO
// workgroups per CU (two waves per SIMD), so one wave's LDS / memory waits are covered by the other's arithmetic, and
// every 64-bit move, select and transcendental expansion of the fp64 path shrinks.
//
// Precision design (errors measured against the fp64 oracle in tests/test_gpu_f32.py):
//   * the state lives in HBM in fp64 for every instance, so handles, ABI and the other kernels are unchanged;
//   * the position states q_ew[4], h_e are INTEGRATED in fp64 (their per-step increments are ~1e-8 of their magnitude,
//     below fp32 resolution: in fp32 the aircraft would not move over the Earth); the RHS reads them rounded to fp32;
//   * everything else — attitude, rates, velocities, aerodynamics, engine, mass, dynamics — is fp32;
//   * fp32 cannot resolve wheel heights (ECEF coordinates are ~6.4e6 m: 0.5 m per ulp), so this instance is airborne-only by
//     construction: a lane that comes within 10 m of the terrain stops uncommitted and is re-run, like in the fp64 path, by
//     the fp64 ground-capable kernel k_step<false, WA, true>.
//   * q_wb is renormalised every step (its fp32 drift per step is of the order of the 1e-8 trigger of the reference).
#pragma once
#include "c172_kernels.hpp"

namespace fbf {

constexpr int F32_ROWS_INPUT = 6;            // de da dr df throttle mixture; payload masses are read from global memory
constexpr int XP0 = FB_X_Q_EW, XPN = 5;      // rows integrated in fp64: q_ew[4], h_e
FBD constexpr int xsrow(int j) { return j < XP0 ? j : j - XPN; }   // row of state j in the fp32 x_n panel
template <int STRIDE>
struct InputsLdsF {
    lds_cptr p;
    const double* u_glob;
    int64_t n;
    int ui;
    FBD real get_de() const { return p[0 * STRIDE]; }
    FBD real get_da() const { return p[1 * STRIDE]; }
    FBD real get_dr() const { return p[2 * STRIDE]; }
    FBD real get_df() const { return p[3 * STRIDE]; }
    FBD real get_throttle() const { return p[4 * STRIDE]; }
    FBD real get_mixture() const { return p[5 * STRIDE]; }
    FBD real get_m_pld(int k) const { return clampd((real)u_glob[(FB_U_M_PILOT + k) * n], 0, 100); }
    FBD real get_steering() const { return 0; }   // ground-only inputs: never read by the airborne instance
    FBD real get_brake(int) const { return 0; }
};

__global__ __launch_bounds__(fbd::STEP_BLOCK, 2) void k_step_f32(fbd::KArgs a, int nsteps) {
    constexpr int B = fbd::STEP_BLOCK;
    __shared__ float lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ float rk[LDS_RK_DOUBLES];
    __shared__ float xs_l[(FB_NX - XPN) * B];   // x_n of the fp32 rows (the five fp64-integrated rows live in xp_l)
    __shared__ float acc_l[FB_NX * B];     // k1 + 2 k2 + 2 k3
    __shared__ double xp_l[XPN * B];       // x_n of the fp64-integrated rows
    __shared__ float in_l[F32_ROWS_INPUT * B];
    // tables: fp64 blob in global memory -> fp32 in LDS (propeller compacted to four coefficients like the fp64 stepper)
    for (int k = threadIdx.x; k < AT_SIZE + PT_SIZE; k += blockDim.x) lds[k] = (float)a.tables[k];
    for (int k = threadIdx.x; k < PR_NJ * PR_NM * PR_NC_STEP; k += blockDim.x)
        lds[LDS_PROP + k] = (float)a.tables[LDS_PROP + (k / PR_NC_STEP) * PR_NC + (k % PR_NC_STEP)];
    for (int k = threadIdx.x; k < LDS_RK_DOUBLES; k += blockDim.x) rk[k] = (float)(1.0 / (a.tables[k + 1] - a.tables[k]));
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != 0) return;
    bool dead = false;
    const int t = threadIdx.x;
    float xt[FB_NX];
#pragma unroll
    for (int k = 0; k < FB_NX; k++) {
        const double v = a.x[(int64_t)k * a.n + i];
        xt[k] = (float)v;
        if (k >= XP0 && k < XP0 + XPN) xp_l[(k - XP0) * B + t] = v;
        else xs_l[xsrow(k) * B + t] = xt[k];
    }
    {
        fbd::Inputs in_r;
        fbd::load_inputs(a, i, in_r);
        lds_ptr q = (lds_ptr)in_l + t;
        q[0 * B] = (float)in_r.de; q[1 * B] = (float)in_r.da; q[2 * B] = (float)in_r.dr; q[3 * B] = (float)in_r.df;
        q[4 * B] = (float)in_r.throttle; q[5 * B] = (float)in_r.mixture;
    }
    const int ui = a.ui[i];
    int stall = a.s[i], eng = a.s[a.n + i];
    const Env env = {(float)a.env.T_sl, (float)a.env.p_sl, (float)a.env.wind_n, (float)a.env.wind_e, (float)a.env.wind_d, (float)a.env.h_trn, a.env.surface};
    const float dt = (float)a.dt, hdt = (float)(a.dt / 2), dt6 = (float)(a.dt / 6);
    const double dtd = a.dt, hdtd = a.dt / 2, dt6d = a.dt / 6;
    int stage = 0, step = 0;
    bool pending_cb = false;
#pragma unroll 1
    while (true) {
        float xn[FB_NX];
        StepAux aux;
        int lds_off = 0;
        asm volatile("" : "+s"(lds_off));   // keeps the loop-invariant table / input loads inside the loop (see k_step)
        const Tables T = {(lds_cptr)lds + lds_off, a.egm96, (lds_cptr)rk + lds_off};
        const InputsLdsF<B> inl = {(lds_cptr)in_l + t + lds_off, a.u + i + lds_off, a.n, ui};
        const float cdt = (stage == 2) ? dt : hdt;
        const double cdtd = (stage == 2) ? dtd : hdtd;
        auto emit = [&](int j, float kj) {
            const int idx = j * B + t;
            if (j >= XP0 && j < XP0 + XPN) {   // fp64 integration of the position rows (j is a compile-time constant at every call)
                const int ip = (j - XP0) * B + t;
                const double xs = xp_l[ip];
                if (stage == 0) { acc_l[idx] = kj; xn[j] = (float)(xs + cdtd * (double)kj); }
                else if (stage < 3) { acc_l[idx] = acc_l[idx] + 2 * kj; xn[j] = (float)(xs + cdtd * (double)kj); }
                else { const double v = xs + dt6d * ((double)acc_l[idx] + (double)kj); xp_l[ip] = v; xn[j] = (float)v; }
            } else {
                const int ix = xsrow(j) * B + t;
                const float xs = xs_l[ix];
                if (stage == 0) { acc_l[idx] = kj; xn[j] = xs + cdt * kj; }
                else if (stage < 3) { acc_l[idx] = acc_l[idx] + 2 * kj; xn[j] = xs + cdt * kj; }
                else { const float v = xs + dt6 * (acc_l[idx] + kj); xs_l[ix] = v; xn[j] = v; }
            }
        };
        int32_t bits = rhs<FB_KIN_WA, false>(xt, stall, eng, inl, env, T, emit, aux, NoSink{});
        if (bits & FB_ST_INTERNAL_REDO) { a.redo[i] = 1; return; }   // within reach of the ground: the fp64 kernel takes this lane over
        bool mod = false;
        if (stage == 0 && pending_cb) {   // f_step! on x_{n+1} (aircraftbase.jl:172-181; kinematics.jl:226-229; c172.jl:375-384,715-724; piston.jl:428-453)
            pending_cb = false;
            step++;
            {   // q_wb: renormalised every step, silently (fp32 drift); q_ew: the reference's rule, on the fp64 copy
                const float inr = rsqrtf(xt[FB_X_Q_WB] * xt[FB_X_Q_WB] + xt[FB_X_Q_WB + 1] * xt[FB_X_Q_WB + 1] +
                                         xt[FB_X_Q_WB + 2] * xt[FB_X_Q_WB + 2] + xt[FB_X_Q_WB + 3] * xt[FB_X_Q_WB + 3]);
#pragma unroll
                for (int k = 0; k < 4; k++) { xt[FB_X_Q_WB + k] *= inr; xs_l[xsrow(FB_X_Q_WB + k) * B + t] = xt[FB_X_Q_WB + k]; }
                double n2 = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) n2 += xp_l[k * B + t] * xp_l[k * B + t];
                const double nr = sqrt(n2);
                if (fabs(nr - 1.0) > 1e-8) {
#pragma unroll
                    for (int k = 0; k < 4; k++) { const double v = xp_l[k * B + t] / nr; xp_l[k * B + t] = v; xt[XP0 + k] = (float)v; }
                    mod = true;
                }
            }
            const int stall0 = stall, eng0 = eng;
            if (aux.alpha > c172::alpha_stall_hi) stall = 1;
            else if (aux.alpha < c172::alpha_stall_lo) stall = 0;
            if (aux.crash) bits |= FB_ST_GROUND_CRASH;
#pragma unroll
            for (int k = 0; k < 6; k++) {   // airborne: the contact regulators are reset (landinggear.jl:479-483)
                if (xt[FB_X_LDG_FRC + k] != 0.0f) { xt[FB_X_LDG_FRC + k] = 0.0f; xs_l[xsrow(FB_X_LDG_FRC + k) * B + t] = 0.0f; mod = true; }
            }
            const float w = xt[FB_X_ENG_OMEGA];
            const bool fuel = aux.m_avail > 0;
            const bool start = ui & FB_UI_ENG_START, stop = ui & FB_UI_ENG_STOP;
            if (eng == 0) { if (start) eng = 1; }
            else if (eng == 1) { if (!start) eng = 0; if (w > c172::w_idle && fuel) eng = 2; }
            else if (stop || w < c172::w_stall || !fuel) eng = 0;
            mod = mod || stall != stall0 || eng != eng0;
            if (bits != 0) { a.status[i] |= bits; dead = true; bits = 0; }
            if (dead || step == nsteps) break;
            if (mod) continue;   // k1 must be re-evaluated on the modified state
        }
        if (bits != 0) { a.status[i] |= bits; dead = true; }
#pragma unroll
        for (int j = 0; j < FB_NX; j++) xt[j] = xn[j];
        stage = (stage + 1) & 3;
        pending_cb = (stage == 0);
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < FB_NX; j++) bad = bad || !isfinite(xt[j]);
    if (bad) a.status[i] |= FB_ST_NAN;
#pragma unroll
    for (int j = 0; j < FB_NX; j++)
        a.x[(int64_t)j * a.n + i] = (j >= XP0 && j < XP0 + XPN) ? xp_l[(j - XP0) * B + t] : (double)xt[j];
    a.s[i] = stall;
    a.s[a.n + i] = eng;
}

}  // namespace fbf
