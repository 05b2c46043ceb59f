// c172_kernels.hpp — HIP kernels for gfx950: f_ode, fused RK4 stepper, f_step, trim.
//
// Mapping: one lane = one aircraft; state is structure-of-arrays in HBM (x[k*N + i]) so every load / store is one coalesced
// 512-B request per wave. The three small tables are copied into LDS once per workgroup. The stepping kernels keep an aircraft's
// state on chip (LDS panels, see "the stepping kernel" below) for the `nsteps` RK4 steps of a launch, so HBM traffic per
// aircraft-step is (27 x 8 B read + 27 x 8 B write + flags) / nsteps. Cessna172Sv0 in fp64 (WA, ECEF, NED) is stepped by the wave-specialised
// k_step_duo<KIN> (512-thread workgroups: two waves per aircraft group, c172_duo_device.hpp), everything else by k_step_air.
//
// Stepper semantics (lib/FlightCore/src/sim.jl:204-218,301-328 + OrdinaryDiffEq RK4):
//   k1 = f(x_n) ; k2 = f(x_n + dt/2 k1) ; k3 = f(x_n + dt/2 k2) ; k4 = f(x_n + dt k3)
//   x_{n+1} = x_n + dt/6 (2 (k2 + k3) + (k1 + k4)) ; y = f(x_{n+1}) ; f_step!(x_{n+1}, y)
// The reference evaluates f six times per step (k1..k4, the evaluation at x_{n+1} that feeds the
// callbacks, and a re-evaluation after the state-modifying callback). f is a pure function of
// (x, u, s), so the evaluation at x_{n+1} IS k1 of the next step whenever f_step! changed nothing;
// the kernel reuses it (4 evaluations per step) and re-evaluates only for the lanes whose f_step!
// did modify x or s (quaternion renormalisation, stall flip, engine state change).
#pragma once
#include "c172_device.hpp"
#include "c172x_ctl_device.hpp"
#include "c172_duo_device.hpp"

namespace fbd {

// ---- ground contact (landinggear.jl:260-328, 426-476) -----------------------------------------
// Two forms of each function. !FAST follows the reference operation by operation with the library's acos / atan2 / sincos (f_ode!, the
// output record). FAST is what the stepping kernels run — a batch on the ground spends half of every evaluation here (three units x
// ~1 080 instructions before, profiles/r03_ground_*): the same quantities with
//   * what the three units share (q_en, strut axis and terrain normal in ECEF axes, the tilt criterion) formed once (ground_common);
//   * rotations of a coordinate axis and products with a z-rotation written out for their structural zeros (the generic forms multiply
//     by literal 0 and 1, which the compiler may not fold without fast-math flags); same operations on the non-zero terms, so the values
//     are the reference's (a state quaternion off unit norm keeps its (1 - s^2) v + s^2 R v behaviour, DESIGN.md "Faithfulness rule");
//   * the wheel frame's quaternion from its z-rotation matrix in closed form (RQuat(RMatrix) has two live cases for such a matrix);
//   * a castoring nose wheel's cos / sin of HALF the velocity azimuth by half_angle_cs (no atan2 + sincos), a steered one's by a short
//     series (|psi| <= pi / 12);
//   * the skid azimuth by the stepping kernels' table atan2; reciprocal square roots where the reference divides by a square root;
//   * F_b = N (q_sc f_c) from the rotation already made for N, instead of rotating N f_c again.
template <bool FAST>
__device__ FB_GROUND_ATTR void ground_common(quat q_eb, quat q_en, GroundCommon& c) {
    c.q_en = q_en;
    if constexpr (FAST) {
        // qrot(q, (0,0,1)) = (2 (q_y q_w + q_z q_x), 2 (q_z q_y - q_x q_w), 1 - 2 (q_x^2 + q_y^2)), term by term as the generic form makes them
        auto rot_z = [](quat q) { return v3{(2 * q.y) * q.w - (2 * q.z) * (-q.x), (2 * q.z) * q.y - (2 * q.x) * q.w, 1 + ((2 * q.x) * (-q.x) - (2 * q.y) * q.y)}; };
        c.ks_e = rot_z(q_eb);
        c.ut_e = rot_z(c.q_en);
    } else {
        c.ks_e = qrot(q_eb, v3{0, 0, 1});
        c.ut_e = qrot(c.q_en, v3{0, 0, 1});
    }
    c.ut_ks = dot(c.ut_e, c.ks_e);
    if constexpr (FAST) {
        const bool lo = c.ut_ks < 0.5 - 1e-7, hi = c.ut_ks > 0.5 + 1e-7;
        c.tilt_crash = lo;
        if (__builtin_amdgcn_ballot_w64(!lo && !hi) != 0) {
            const double a = acos(fmax(fmin(c.ut_ks, 1.0), -1.0));
            if (!lo && !hi) c.tilt_crash = a * (180 / PI) > 60;
        }
        c.alpha_ts = 0;
    } else {
        c.alpha_ts = acos(fmax(fmin(c.ut_ks, 1.0), -1.0));
        c.tilt_crash = c.alpha_ts * (180 / PI) > 60;
    }
}
template <bool FAST>
__device__ FB_GROUND_ATTR void gear_ground_kinematics(const GroundIn& in, GroundOut& o) {
    using namespace c172;
    const int g = in.g;
    const v3 r_bs_b = {ldg_r[g][0], ldg_r[g][1], ldg_r[g][2]};
    // terrain point under the wheel, ECEF (geodesy.jl:418-428)
    // (kept as the reference's a / sqrt(...) in both forms: the strut's compression is the difference of two ECEF positions 6.4e6 m from
    // the origin, and a second ulp on this radius is a nanometre of compression in front of 4e4 N/m)
    const double RE = wgs::a / sqrt(1 - wgs::e2 * in.loc_Ot.z * in.loc_Ot.z);
    const v3 r_et_e = {(RE + in.he_Ot) * in.loc_Ot.x, (RE + in.he_Ot) * in.loc_Ot.y, (RE * (1 - wgs::e2) + in.he_Ot) * in.loc_Ot.z};
    const v3 r_es_e = in.r_eb_e + in.r_bs_e;
    const v3 r_st_e = r_et_e - r_es_e;
    const double l = dot(in.ut_e, r_st_e) / in.ut_ks;
    o.xi = fmin(0.0, l);
    const v3 r_bc_b = v3{0, 0, o.xi} + r_bs_b;
    const v3 v_body = in.v_eb_b + cross(in.w_eb_b, r_bc_b);
    quat q_sc;
    if constexpr (FAST) {
        // q_nw = q_nb o Rz(psi_sw): main wheels psi_sw = 0; nose wheel steered (psi = steer psi_max) or castoring (psi = azimuth of v)
        quat q_nw = in.q_nb;
        if (g == 2) {
            double c, s;
            if (in.steer_engaged) {
                const double h = 0.5 * (in.steer_in * psi_max), z = h * h;   // |h| <= pi / 12: sin to h^13, cos to h^14 (< 1e-18)
                double ps = -1.0 / 6227020800.0, pc = -1.0 / 87178291200.0;
                ps = __builtin_fma(z, ps, 1.0 / 39916800.0); ps = __builtin_fma(z, ps, -1.0 / 362880.0); ps = __builtin_fma(z, ps, 1.0 / 5040.0);
                ps = __builtin_fma(z, ps, -1.0 / 120.0); ps = __builtin_fma(z, ps, 1.0 / 6.0);
                pc = __builtin_fma(z, pc, 1.0 / 479001600.0); pc = __builtin_fma(z, pc, -1.0 / 3628800.0); pc = __builtin_fma(z, pc, 1.0 / 40320.0);
                pc = __builtin_fma(z, pc, -1.0 / 720.0); pc = __builtin_fma(z, pc, 1.0 / 24.0); pc = __builtin_fma(z, pc, -0.5);
                s = __builtin_fma(-(h * z), ps, h);
                c = __builtin_fma(z, pc, 1.0);
            } else {
                half_angle_cs(v_body.y, v_body.x, c, s);
            }
            const quat a = in.q_nb;   // qmul(a, {c, 0, 0, s}) without its zero terms
            q_nw = {a.w * c - a.z * s, c * a.x + a.y * s, c * a.y - a.x * s, a.w * s + c * a.z};
        }
        // iw_n = qrot(q_nw, (1,0,0)), of which the horizontal part is kept: ic = (x, y, 0) / |(x, y)|, jc = kc x ic, kc = (0,0,1)
        const double iwx = 1 + ((2 * q_nw.y) * (-q_nw.y) - (2 * q_nw.z) * q_nw.z);
        const double iwy = (2 * q_nw.z) * q_nw.w - (2 * q_nw.x) * (-q_nw.y);
        const double in_ = rsqrt(iwx * iwx + iwy * iwy);
        const double icx = in_ * iwx, icy = in_ * iwy;
        // RQuat(RMatrix([ic jc kc])) (attitude.jl:192-233) for R = Rz: trace 1 + 2 ic.x; the largest-diagonal search ends on the trace
        // unless 1 > trace (then on R33 = 1)
        const double tr = (icx + icx) + 1;
        const bool on_trace = !(1 > tr);
        const double vw = on_trace ? 1 + tr : icy + icy, vz = on_trace ? icy + icy : 1 + 2 * 1.0 - tr;
        const double inv = rsqrt(vw * vw + vz * vz);
        const double cw = vw * inv, sz = vz * inv;   // q_nc = (cw, 0, 0, sz)
        const quat a = qconj(in.q_nb);               // q_sc = q_bn o q_nc (q_ns = q_nb: q_bs = 1)
        q_sc = {a.w * cw - a.z * sz, cw * a.x + a.y * sz, cw * a.y - a.x * sz, a.w * sz + cw * a.z};
    } else {
        const double psi_v = atan2(v_body.y, v_body.x);
        double psi_sw = 0.0;
        if (g == 2) psi_sw = in.steer_engaged ? in.steer_in * psi_max : psi_v;
        double s, c;
        sincos(0.5 * psi_sw, &s, &c);
        const quat q_nw = qmul(in.q_nb, quat{c, 0, 0, s});
        const v3 iw_n = qrot(q_nw, v3{1, 0, 0});
        const v3 kc = {0, 0, 1};
        const v3 iw_t = iw_n - dot(iw_n, kc) * kc;
        const v3 ic = (1 / norm(iw_t)) * iw_t;
        const v3 jc = cross(kc, ic);
        // RQuat(RMatrix([ic jc kc])) (attitude.jl:192-233)
        const double R[3][3] = {{ic.x, jc.x, kc.x}, {ic.y, jc.y, kc.y}, {ic.z, jc.z, kc.z}};
        const double tr = R[0][0] + R[1][1] + R[2][2];
        int imax = 0;
        double best = tr;
        if (R[0][0] > best) { best = R[0][0]; imax = 1; }
        if (R[1][1] > best) { best = R[1][1]; imax = 2; }
        if (R[2][2] > best) { best = R[2][2]; imax = 3; }
        quat v;
        if (imax == 0) v = {1 + tr, R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]};
        else if (imax == 1) v = {R[2][1] - R[1][2], 1 + 2 * R[0][0] - tr, R[0][1] + R[1][0], R[2][0] + R[0][2]};
        else if (imax == 2) v = {R[0][2] - R[2][0], R[0][1] + R[1][0], 1 + 2 * R[1][1] - tr, R[1][2] + R[2][1]};
        else v = {R[1][0] - R[0][1], R[2][0] + R[0][2], R[1][2] + R[2][1], 1 + 2 * R[2][2] - tr};
        const double inv = 1 / sqrt(v.w * v.w + v.x * v.x + v.y * v.y + v.z * v.z);
        const quat q_nc = {v.w * inv, v.x * inv, v.y * inv, v.z * inv};
        q_sc = qmul(qconj(in.q_nb), q_nc);  // q_ns = q_nb (q_bs = 1)
    }
    const v3 v_c_body = qrot_inv(q_sc, v_body);
    v3 ks_c;
    if constexpr (FAST) {
        const quat p = qconj(q_sc);   // qrot(p, (0,0,1)), as in ground_common
        ks_c = {(2 * p.y) * p.w - (2 * p.z) * (-p.x), (2 * p.z) * p.y - (2 * p.x) * p.w, 1 + ((2 * p.x) * (-p.x) - (2 * p.y) * p.y)};
    } else {
        ks_c = qrot_inv(q_sc, v3{0, 0, 1});
    }
    o.xi_dot = -v_c_body.z / ks_c.z;
    const double kd = ldg_kd[g];  // k_d_ext == k_d_cmp for both C172 dampers (c172.jl:444-451)
    o.F_dmp = -(ldg_ks[g] * o.xi + kd * o.xi_dot);
    const v3 v_c = v_c_body + o.xi_dot * ks_c;
    o.st = (fabs(v_c.z) < 1e-8) ? 0 : FB_ST_CONTACT_ASSERT;
    o.v_xy0 = v_c.x;
    o.v_xy1 = v_c.y;
    o.q_sc = q_sc;
    o.r_bc_b = r_bc_b;
}

template <bool FAST>
__device__ FB_GROUND_ATTR void gear_ground_force(const GroundIn& in, GroundOut& o) {
    const double nv = sqrt(o.v_xy0 * o.v_xy0 + o.v_xy1 * o.v_xy1);
    auto mu = [&](double mu_s, double mu_d) {
        const double k = fmin(fmax((nv - 0.005) / (0.01 - 0.005), 0.0), 1.0);
        return k * mu_d + (1 - k) * mu_s;
    };
    const double mu_roll = mu(0.03, 0.02);
    const double mu_skid = in.surface == 0 ? mu(0.75, 0.25) : (in.surface == 1 ? mu(0.25, 0.15) : mu(0.075, 0.025));
    const double k_br = in.brake_in;  // η_br = 1 (landinggear.jl:106-108); nose gear has NoBraking -> 0
    const double mu_x = mu_roll + (mu_skid - mu_roll) * k_br;
    double psi_abs;   // |atan2(v_y, v_x)|
    if constexpr (FAST) psi_abs = (nv < 1e-3) ? PI / 2 : atan2_tab(fabs(o.v_xy1), o.v_xy0, in.atan_tab);
    else psi_abs = fabs((nv < 1e-3) ? PI / 2 : atan2(o.v_xy1, o.v_xy0));
    const double psi_skid = 10 * (PI / 180);
    double mu_y;
    if (psi_abs < psi_skid) mu_y = mu_skid * psi_abs / psi_skid;
    else if (psi_abs > PI - psi_skid) mu_y = mu_skid * (1 - (psi_skid + psi_abs - PI) / psi_skid);
    else mu_y = mu_skid;
    double sc;
    if constexpr (FAST) sc = fmin(1.0, mu_skid * rsqrt(mu_x * mu_x + mu_y * mu_y));
    else sc = fmin(1.0, mu_skid / sqrt(mu_x * mu_x + mu_y * mu_y));
    const v3 f_c = {in.frc_out0 * (mu_x * sc), in.frc_out1 * (mu_y * sc), -1.0};
    const v3 f_s = qrot(o.q_sc, f_c);
    const double N = fmax(0.0, -o.F_dmp / f_s.z);
    v3 F_b;
    if constexpr (FAST) F_b = N * f_s;      // = q_sc(N f_c): the rotation is linear (q_bc = q_sc)
    else F_b = qrot(o.q_sc, N * f_c);
    o.F_b = F_b;
    o.tau_b = cross(o.r_bc_b, F_b);
}

template <bool FAST> __device__ __noinline__ void gear_ground_kinematics_call(const GroundIn& in, GroundOut& o) { gear_ground_kinematics<FAST>(in, o); }
template <bool FAST> __device__ __noinline__ void gear_ground_force_call(const GroundIn& in, GroundOut& o) { gear_ground_force<FAST>(in, o); }

// ---- kernel arguments -----------------------------------------------------------------------
struct KArgs {
    double* x;          // [NX x n]  (NX = 27, or 34 for Cessna172X: rows 27..33 = actuator positions)
    int32_t* s;         // [FB_NS x n]
    const double* u;    // [FB_NU x n]
    const int32_t* ui;  // [n]
    int32_t* status;    // [n]
    const double* tables;  // LDS_TABLE_DOUBLES doubles
    const float* tables_f32;  // the same blob in fp32 (fp32 stepper only)
    const float* egm96;
    int64_t n;
    Env env;
    const double* env_rows;   // [ENV_DEV_ROWS x n] per-aircraft environment (fb_set_env), or null: the batch-wide block `env` (fb_params)
    double dt;
    // Cessna172X only
    double* cs;         // [FB_NCS x n] control-law record (actuator commands live here)
    const double* cu;   // [FB_NCU x n] control-law inputs
    double* q_pre;      // [8 x n] q_wb, q_ew of the last evaluation before f_step! (what the periodic update must see)
    int32_t* redo;      // [n] set by the airborne pass of the stepping kernel for lanes that came within reach of the ground
    double* k1;         // [FB_NX x n] Cessna172X: derivative left by the last evaluation of the previous launch (FSAL across launches)
    int32_t* k1_valid;  // [n] 1 when k1 is the derivative at the current x, s, u
    // Cessna172X: f_periodic!(avionics, vehicle) runs INSIDE the stepping kernels, after the f_step! of every step n with
    // (ctl_phase + n) % ctl_ratio == 0 (cb_periodic after cb_step, FC/sim.jl:204-218, 366-381)
    const double* gains;  // FB_TABLE_CTL_GAINS blob (global memory, read through scalar loads)
    CtlOffsets ctl_off;
    double ctl_dT;        // the control laws' sample period
    int ctl_ratio;        // Δt / dt (0: never)
    int ctl_phase;        // steps taken since the last init, modulo ctl_ratio, when the launch starts
    double* ctl_bak;      // [(FB_NCS + FB_NCU) x n] the airborne pass's copy of cs | cu at launch start (restored for lanes it hands over)
    double* duo_pld;      // [DUO_NCONST x n] k_step_duo: per-aircraft constants of the launch (the deflection-only aerodynamic terms),
                          // written by its prologue and fetched at the start of every evaluation's aerodynamics block
                          // (Cessna172Xv2: of the EVALUATION — role P forms them from the actuators' stage positions and hands them over here)
    double* duo_tap;      // [DUO_NTAP x n] k_step_duo<KIN, true>: what the two halves of a control update hand each other (DUO_TAP_*)
    // the termination record (fb_get_termination): written once, when an aircraft's simulation ends
    long long* term_step; // [n] RK updates completed since the last init when the exception was thrown
    int32_t* term_where;  // [n] FB_TERM_*
    long long step0;      // steps taken since the last init when the launch starts
};
// The one exception the reference would have thrown out of an f_ode! whose checks raised `bits`: kinematics and air data come first
// (altitude below h_min: FP/kinematics.jl:190,199 -> geodesy.jl:218-221; ISA range: atmosphere.jl:133), the struts after them
// (landinggear.jl:240, 321), the centre of mass last (dynamics.jl:477-486). (A strut's assertion ahead of a LATER strut's altitude
// error is not told apart: both need the wheels on the ground AND below h_min.)
// The kernel's own argument block, re-read where it is needed: a stepping kernel that keeps a dozen of its arguments in SGPRs across its
// loop for something that happens once per control period (the arguments of a control update, the base of the tap rows) pays for them
// with SGPR spills inside the loop. kernarg() returns the kernarg segment — KArgs is the first parameter of every stepping kernel — through
// an opaque copy, so that loads through it are issued where they stand (scalar loads, one exposed round trip per use).
typedef __attribute__((address_space(4))) const KArgs* kargs_cptr;
FBD kargs_cptr kernarg() {
    kargs_cptr p = (kargs_cptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
// Per-aircraft environment (fb_set_env): in the reference wind, sea-level conditions and terrain elevation are inputs of EACH simulation's own
// atmosphere / terrain models (FP/atmosphere.jl:75-84,156-165, FP/terrain.jl:34-48) — N simulations, N environments. Rows FB_ENV_* of the
// caller's panel, then two derived rows the host fills like make_args fills Env::ln_p_sl / k_rt. The surface type stays batch-wide.
// A kernel instance with PERENV holds the lane's nine values in VGPRs (the batch-wide block sits in SGPRs): k_step_air (one wave per SIMD,
// registers to spare) in every mechanisation; k_step_duo in the WA mechanisation, where role P keeps its four values in registers and role
// D reads its four from an 8 KB LDS panel (see there); the single-call verbs choose at run time.
constexpr int ENV_DEV_LN_P = FB_NENV, ENV_DEV_K_RT = FB_NENV + 1, ENV_DEV_ROWS = FB_NENV + 2;
FBD Env env_of(const KArgs& a, int64_t i) {
    const double* e = a.env_rows + i;
    const int64_t n = a.n;
    return {e[(int64_t)FB_ENV_T_SL * n], e[(int64_t)FB_ENV_P_SL * n], e[(int64_t)FB_ENV_WIND_N * n], e[(int64_t)FB_ENV_WIND_E * n], e[(int64_t)FB_ENV_WIND_D * n],
            e[(int64_t)FB_ENV_H_TERRAIN * n], a.env.surface, e[(int64_t)ENV_DEV_LN_P * n], e[(int64_t)ENV_DEV_K_RT * n]};
}
// the single-call verbs: one instance, the choice made at run time (the nine values in VGPRs either way)
FBD Env env_any(const KArgs& a, int64_t i) { return a.env_rows ? env_of(a, i) : a.env; }
// A status bit raised outside fb_step (the single-call verbs): the record says so, with the step count of the moment — unless the
// aircraft already carries a record (fb_get_termination promises a valid step and place next to every termination bit)
FBD void mark_outside_step(int32_t* status, long long* term_step, int32_t* term_where, long long step0, int64_t i, int32_t st) {
    if (st == 0) return;
    if ((status[i] & ~FB_ST_NAN) == 0 && (st & ~FB_ST_NAN) != 0) { term_where[i] = FB_TERM_OUTSIDE_STEP; term_step[i] = step0; }
    status[i] |= st;
}
FBD int32_t first_exception(int32_t bits) {
    if (bits & FB_ST_ALT_RANGE) return FB_ST_ALT_RANGE;
    if (bits & FB_ST_ISA_RANGE) return FB_ST_ISA_RANGE;
    if (bits & FB_ST_CONTACT_ASSERT) return FB_ST_CONTACT_ASSERT;
    return bits;
}

constexpr int STEP_BLOCK = 256;  // lanes per workgroup of the stepping kernel
constexpr int DUO_NCONST = 12;   // rows of KArgs::duo_pld: 8 aerodynamic sums, 2 x (interval, weight) of the flap-axis locations

// Stage the [aero | piston | propeller] blob into LDS. NC = propeller coefficients kept per grid point: all six when the
// full output record is produced, the first four (C_Fx, C_Mx, C_Fz_α, C_Mz_α) otherwise — 7 KB of LDS less.
template <int NC>
FBD void stage_tables(double* lds, double* rk, const double* tables) {
    for (int k = threadIdx.x; k < AT_SIZE + PT_SIZE; k += blockDim.x) lds[k] = tables[k];
    for (int k = threadIdx.x; k < PR_NJ * PR_NM * NC; k += blockDim.x) lds[LDS_PROP + k] = tables[LDS_PROP + (k / NC) * PR_NC + (k % NC)];
    __syncthreads();
    // reciprocal knot spacings for every position of the aero|piston blob (only knot positions are ever read)
    for (int k = threadIdx.x; k < LDS_RK_KNOTS; k += blockDim.x) rk[k] = 1.0 / (lds[k + 1] - lds[k]);
    for (int k = threadIdx.x; k < ATAN_N; k += blockDim.x) rk[LDS_ATAN + k] = atan(k * (1.0 / 32));   // atan2_tab()'s table
    __syncthreads();
}
FBD void load_inputs(const KArgs& a, int64_t i, Inputs& in) {
    make_inputs(in, a.u + i, a.n, a.ui[i]);
    in.u_glob = a.u + i;
    in.n = a.n;
}
// Cessna172X: the seven actuator commands in force — the four control-law outputs (assign!, c172x_ctl.jl:449-458, 986-995)
// and the flaps / brake commands of u — saturated to the actuators' Ranged input types (c172x.jl:113-119)
// An actuator's position at an RK4 stage, from its closed form between control updates (x_n, the command c, z = dt / tau): the stage's argument is
// c + (x_n - c) ms with ms = 1, 1 - z/2, 1 - z/2 + z^2/4, 1 - z + z^2/2 - z^3/4. Written from the other end — x_n + (c - x_n) (1 - ms) — the step's
// FIRST stage is the state itself, bit for bit (1 - ms = 0), as in the reference, whose RK4 evaluates k1 at x_n; c + (x_n - c) rounds differently for
// another c, and then the derivative at x_{n+1} that one launch carries over to the next (k_step_air<X>'s k1, evaluated ahead of the control update with
// the OLD commands) and the one a launch evaluates for itself (behind it, with the NEW ones) are not the same numbers: a run cut into launches
// differently — a host callback after every step against a scenario table on the device — differed in the last place
// (tests/test_gpu_scenarios.py compares the two bit for bit). FB_ACT_STAGE_FORM: 2 this form; 1 the old one with the first stage selected
// (+0.9 % on the Cessna172Xv2 launch, profiles/r06_ab_stage0.txt); 0 the old one (rounds 3-5).
#ifndef FB_ACT_STAGE_FORM
#define FB_ACT_STAGE_FORM 2
#endif
FBD double act_stage_mc(int stage, double z) {   // 1 - ms (form 2), ms (forms 0, 1)
    if constexpr (FB_ACT_STAGE_FORM == 2) return stage == 0 ? 0.0 : (stage == 1 ? z / 2 : (stage == 2 ? z / 2 - z * z / 4 : z - z * z / 2 + z * z * z / 4));
    else return stage == 0 ? 1.0 : (stage == 1 ? 1 - z / 2 : (stage == 2 ? 1 - z / 2 + z * z / 4 : 1 - z + z * z / 2 - z * z * z / 4));
}
FBD double act_stage_pos(double x_n, double c, double mc, [[maybe_unused]] bool first_stage) {
    if constexpr (FB_ACT_STAGE_FORM == 2) return __builtin_fma(c - x_n, mc, x_n);
    else if constexpr (FB_ACT_STAGE_FORM == 1) return first_stage ? x_n : c + (x_n - c) * mc;
    else return c + (x_n - c) * mc;
}
// (in two parts — the row as it stands in memory, and its saturation — for callers that want all seven loads in flight before the first clamp)
FBD double x2_command_row(const KArgs& a, int64_t i, int k) {
    const int64_t n = a.n;
    switch (k) {
        case FB_ACT_THROTTLE: return a.cs[(int64_t)FB_CS_THROTTLE_CMD * n + i];
        case FB_ACT_AILERON: return a.cs[(int64_t)FB_CS_AILERON_CMD * n + i];
        case FB_ACT_ELEVATOR: return a.cs[(int64_t)FB_CS_ELEVATOR_CMD * n + i];
        case FB_ACT_RUDDER: return a.cs[(int64_t)FB_CS_RUDDER_CMD * n + i];
        case FB_ACT_FLAPS: return a.u[(int64_t)FB_U_FLAPS * n + i];
        case FB_ACT_BRAKE_LEFT: return a.u[(int64_t)FB_U_BRAKE_LEFT * n + i];
        default: return a.u[(int64_t)FB_U_BRAKE_RIGHT * n + i];
    }
}
FBD double x2_command_sat(int k, double v) { return clampd(v, (k == FB_ACT_AILERON || k == FB_ACT_ELEVATOR || k == FB_ACT_RUDDER) ? -1.0 : 0.0, 1.0); }
FBD double x2_command(const KArgs& a, int64_t i, int k) { return x2_command_sat(k, x2_command_row(a, i, k)); }
// dst[k n + i] = src[k n + i], k < ROWS, G rows at a time: G loads in flight, then G stores. (Row by row — a load, a wait, a store, the next
// load behind the store it may alias — the launch-start copy of the control-law record was 94 dependent memory round trips per workgroup.)
// (k_step_duo's two copies: rows per batch, and a diagnostic switch — FB_X2_BAK = 0 leaves the copies out, which is WRONG for a lane that is handed
// over behind a control update and only measures what they cost)
#ifndef FB_X2_BAK
#define FB_X2_BAK 1
#endif
#ifndef FB_X2_BAK_G_CS
#define FB_X2_BAK_G_CS 11
#endif
#ifndef FB_X2_BAK_G_CU
#define FB_X2_BAK_G_CU 14
#endif
template <int ROWS, int G>
__device__ __forceinline__ void copy_rows_batched(double* dst, const double* src, int64_t n, int64_t i) {
    static_assert(ROWS % G == 0, "");
#pragma unroll 1
    for (int k0 = 0; k0 < ROWS; k0 += G) {
        double v[G];
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = src[(int64_t)(k0 + g) * n + i];
#pragma unroll
        for (int g = 0; g < G; g++) dst[(int64_t)(k0 + g) * n + i] = v[g];
    }
}
template <bool X> struct Dims { static constexpr int NXT = X ? (int)FB_X2_NX : (int)FB_NX; };

// vehicle.y as the control laws see it, from the state x (device row order): one RHS evaluation with a partial sink — everything
// that does not feed the twelve tapped outputs is dead code. GROUND = false: the caller knows the aircraft is clear of the terrain.
// the row of the ellipsoidal altitude in each mechanisation's own state block (c172_device_impl.inc, rhs())
template <int KIN> constexpr int h_e_row() { return KIN == FB_KIN_WA ? (int)FB_X_H_E : (KIN == FB_KIN_ECEF ? FB_X_Q_WB + 7 : FB_X_Q_WB + 5); }
template <bool GROUND, int KIN, class CmdFn>
FBD CtlIn x2_ctl_inputs(const KArgs& a, int64_t i, const Tables& T, const double (&x)[FB_X2_NX], int stall, int eng, int ui, CmdFn&& cmd_of) {
    const InputsX in = {&x[X2_ACT], a.u + i, a.n, ui};
    StepAux aux;
    CtlSink tap;
    rhs<KIN, GROUND>(x, stall, eng, in, env_any(a, i), T, [](int, double) {}, aux, tap);
    CtlIn v;
    v.lat = tap.lat; v.lon = tap.lon;
    v.EAS = tap.EAS; v.h_e = x[h_e_row<KIN>()]; v.theta = tap.theta; v.phi = tap.phi; v.clm = -tap.vd; v.chi = tap.chi;
    v.w_wb_b = {tap.wx, tap.wy, tap.wz};
    v.w_eb_b = {x[FB_X_OMEGA_EB_B], x[FB_X_OMEGA_EB_B + 1], x[FB_X_OMEGA_EB_B + 2]};
    v.alpha = tap.alpha; v.beta = tap.beta; v.alpha_filt = x[FB_X_ALPHA_FILT]; v.beta_filt = x[FB_X_BETA_FILT];
    v.n_eng = x[FB_X_ENG_OMEGA] / c172::w_rated;
#pragma unroll
    for (int k = 0; k < 4; k++) { v.pos[k] = in.pos(k); v.cmd[k] = cmd_of(k); }
    v.on_gnd = GROUND && aux.wow != 0;   // is_on_gnd: any strut with weight on wheels (c172.jl:998-1001)
    return v;
}
// Avionics f_periodic! inside a stepping kernel (c172x2.jl:27-37): guidance, then the control laws, on the outputs of the step's last
// f_ode! — which the stepping loop has tapped from that very evaluation (the one at x_{n+1}, before f_step! renormalises the
// quaternions) through a CtlSink, so nothing is evaluated again here (a second evaluation just for the tap was 12 % of a launch at
// two steps per control period). OUT OF LINE on purpose: it runs once per control period, and inlined into the stepping loop its
// ~9 k instructions cost the loop its register allocation (512 registers + 1.7 KB of scratch); as a call, only the call site pays
// (live registers saved around it).
// wave-uniform values that reach a function through the call ABI arrive in VGPRs and the compiler must treat them as per-lane:
// v_readfirstlane turns them back into SGPR values (scalar loads, SGPR base addresses)
FBD int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
FBD int64_t uni(int64_t v) { return (int64_t)(((uint64_t)(uint32_t)uni((int)((uint64_t)v >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)(uint64_t)v)); }
FBD double uni(double v) { return __builtin_bit_cast(double, uni(__builtin_bit_cast(int64_t, v))); }
template <class P> FBD P* uni(P* p) { return (P*)(uintptr_t)uni((int64_t)(uintptr_t)p); }
// The kernel arguments it reads travel as SCALAR arguments, i.e. in registers (COPIES: handing the callee the kernel's own KArgs by
// reference moved the kernel arguments to scratch for the whole kernel, +25 % on every step; and a struct argument, whatever its
// size, goes through private memory: the callee then starts with an exposed round trip). The ten lookup offsets are packed two
// to a word (the blob is at most CTL_GAINS_MAX = 6144 doubles), `tsg` = total | same_grid << 16.
// Returns the four commands it leaves in cs — in registers: re-reading the rows just written waits for the stores to land first.
struct CtlOut { double cmd[4]; };   // throttle, aileron, elevator, rudder, as x2_command() would read them back
// Inlined into k_step_air<KIN, true, GROUND> since round 4: out of line (rounds 2-3, "the call costs 4.9 k cycles") it was a callee that saved and
// restored the callee-saved registers it used through scratch; inlined, the one-wave airborne pass runs 14.05 -> 12.8 ms per launch of 524 288
// (profiles/r04_x2_inline_ab.txt, second table). -DFB_X2_PERIODIC_ATTR=__noinline__ builds the old form.
#ifndef FB_X2_PERIODIC_ATTR
#define FB_X2_PERIODIC_ATTR __forceinline__
#endif
__device__ FB_X2_PERIODIC_ATTR CtlOut x2_periodic(const double* a_cu, double* a_cs, const double* a_gains, int64_t a_n, double a_dT, uint32_t o01, uint32_t o23,
                                           uint32_t o45, uint32_t o67, uint32_t o89, uint32_t tsg, int64_t i, const CtlIn& v) {
    FB_X2_STAMP(21);
    const double* cu = uni(a_cu); double* cs = uni(a_cs); const double* gains = uni(a_gains);
    const int64_t n = uni(a_n);
    const double dT = uni(a_dT);
    const uint32_t op[5] = {o01, o23, o45, o67, o89};
    CtlOffsets off;
#pragma unroll
    for (int k = 0; k < 10; k++) off.off[k] = uni((int)((op[k / 2] >> (16 * (k % 2))) & 0xffffu));
    off.total = uni((int)(tsg & 0xffffu));
    off.same_grid = uni((int)(tsg >> 16));
    // cs / cu rows are read and written where the laws use them, through global pointers with wave-uniform bases (only the rows of
    // the active modes move; prefetching the whole 94-row record into registers and writing back what changed was measured 2x
    // slower: the copy spills). The gains come by per-lane gather from the L2-resident blob, one table's corner records per burst
    // (scalar loads through a wave-uniform loop over the distinct grid cells were also measured 2x slower: every s_load batch
    // is an exposed L2 round trip behind a 16 KB scalar cache that the 46 KB blob does not fit).
    typedef __attribute__((address_space(1))) double* gptr;
    typedef ctlg_cptr gcptr;
    double lu[FB_NCU], ls[FB_NCS];
    {
        const gcptr u0 = (gcptr)(uintptr_t)cu + i, c0 = (gcptr)(uintptr_t)cs + i;
#pragma unroll
        for (int k = 0; k < FB_NCU; k++) lu[k] = u0[(int64_t)k * n];   // one burst: all 94 rows of the record in flight together
#pragma unroll
        for (int k = 0; k < FB_NCS; k++) ls[k] = c0[(int64_t)k * n];
    }
    const CtlMemCachedT<gptr> M = {(gptr)(uintptr_t)cu + i, (gptr)(uintptr_t)cs + i, n, lu, ls};
    FB_X2_STAMP(22);
    gdc_update(M, v);
    FB_X2_STAMP(23);
    const CtlTabT<gcptr> tab = ctl_tab((gcptr)(uintptr_t)gains, off, v.EAS, v.h_e);
    ctl_lon(tab, M, dT, v, (int)M.U(FB_CU_LON_MODE_REQ));
    FB_X2_STAMP(24);
    // (fetching the lateral gains ahead of the longitudinal channel, to hide their gather behind its dependent chains, holds 40 more
    // values across it: 460 registers in this function, and the calling kernel's allocation pays — 14.2 -> 15.0 ms per launch)
    const int lat_req = (int)M.U(FB_CU_LAT_MODE_REQ);
    ctl_lat(tab, M, dT, v, lat_req, ctl_lat_gains(tab, v, lat_req));
    FB_X2_STAMP(25);
    return {{clampd(M.S(FB_CS_THROTTLE_CMD), 0, 1), clampd(M.S(FB_CS_AILERON_CMD), -1, 1), clampd(M.S(FB_CS_ELEVATOR_CMD), -1, 1), clampd(M.S(FB_CS_RUDDER_CMD), -1, 1)}};
}

// FB_VERB_FAST: the single-call verbs with f_ode! in the stepping kernels' form (their atan2 / log / sincos, knot scans through scalar loads) instead
// of the form with the library's functions. Measured, no gain (k_f_ode 0.404 against 0.405-0.413 ms per 1 048 576 aircraft: it is bound by its
// 1.4 KB of output per aircraft, 3.6 TB/s of stores), so the verbs keep the library form.
#ifndef FB_VERB_FAST
#define FB_VERB_FAST false
#endif
// f_ode!(world): xdot (optional) and the output record y
template <bool X, int KIN>
__global__ __launch_bounds__(256) void k_f_ode(KArgs a, double* xdot, double* y) {
    constexpr int NXT = Dims<X>::NXT;
    __shared__ double lds[LDS_TABLE_DOUBLES];
    __shared__ double rk[LDS_RK_DOUBLES];
    stage_tables<PR_NC>(lds, rk, a.tables);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Tables T = {(lds_cptr)lds, a.egm96, (lds_cptr)rk, (gk_cptr)a.tables};
    double x[NXT], xd[NXT];
#pragma unroll
    for (int k = 0; k < NXT; k++) x[k] = a.x[(int64_t)k * a.n + i];
    StepAux aux;
    auto emit = [&](int j, double v) { xd[j] = v; };
    int32_t st;
    if constexpr (X) {
        const InputsX in = {&x[X2_ACT], a.u + i, a.n, a.ui[i]};
        st = rhs<KIN, true, FB_VERB_FAST>(x, a.s[i], a.s[a.n + i], in, env_any(a, i), T, emit, aux, PanelSink{y + i, a.n});
#pragma unroll
        for (int k = 0; k < FB_NACT; k++) xd[X2_ACT + k] = 1 / ACT_TAU * (x2_command(a, i, k) - x[X2_ACT + k]);   // Actuator1.f_ode!, c172x.jl:39-52
    } else {
        Inputs in;
        load_inputs(a, i, in);
        st = rhs<KIN, true, FB_VERB_FAST>(x, a.s[i], a.s[a.n + i], in, env_any(a, i), T, emit, aux, PanelSink{y + i, a.n});
    }
    if (xdot) {
#pragma unroll
        for (int k = 0; k < NXT; k++) xdot[(int64_t)k * a.n + i] = xd[k];
    }
    mark_outside_step(a.status, a.term_step, a.term_where, a.step0, i, st);
}

// f_step!(world). The reference acts on the y left behind by the last f_ode!; f_ode! is a pure
// function of (x,u,s), so it is recomputed here from the current x.
template <bool X, int KIN>
__global__ __launch_bounds__(256) void k_f_step(KArgs a) {
    constexpr int NXT = Dims<X>::NXT;
    __shared__ double lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ double rk[LDS_RK_DOUBLES];
    stage_tables<PR_NC_STEP>(lds, rk, a.tables);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Tables T = {(lds_cptr)lds, a.egm96, (lds_cptr)rk, (gk_cptr)a.tables};
    double x[NXT], xd[NXT];
#pragma unroll
    for (int k = 0; k < NXT; k++) x[k] = a.x[(int64_t)k * a.n + i];
    int stall = a.s[i], eng = a.s[a.n + i];
    StepAux aux;
    auto emit = [&](int j, double v) { xd[j] = v; };
    int32_t st;
    if constexpr (X) {
        const InputsX in = {&x[X2_ACT], a.u + i, a.n, a.ui[i]};
        st = rhs<KIN, true, FB_VERB_FAST>(x, stall, eng, in, env_any(a, i), T, emit, aux, NoSink{});
        f_step<KIN>(x, stall, eng, in, aux, st);
    } else {
        Inputs in;
        load_inputs(a, i, in);
        st = rhs<KIN, true, FB_VERB_FAST>(x, stall, eng, in, env_any(a, i), T, emit, aux, NoSink{});
        f_step<KIN>(x, stall, eng, in, aux, st);
    }
#pragma unroll
    for (int k = 0; k < FB_NX; k++) a.x[(int64_t)k * a.n + i] = x[k];
    a.s[i] = stall;
    a.s[a.n + i] = eng;
    mark_outside_step(a.status, a.term_step, a.term_where, a.step0, i, st);
}

// ---- nsteps x step!(sim), fused: the stepping kernel ------------------------------------------------------------------------------
// Registers are the limiter: VALU instructions address 256 VGPRs (the AGPR half of the file is only a spill cache), one fp64 value
// takes two, and a single RHS keeps ~200 fp64 values alive. So the RK4 bookkeeping does not live in registers at all: per lane, x_n,
// the running stage sum and the state being evaluated sit in [rows][256] LDS panels (lane-contiguous rows: conflict-free ds_read_b64 /
// ds_write_b64), and the stage update is fused INTO the RHS: each derivative k_j is consumed the moment it is produced (emit), so no
// k[NX] array is ever alive. Stage sum: acc = k1 + 2 k2 + 2 k3, x_{n+1} = x_n + dt/6 (acc + k4) — the classic RK4 combination
// (OrdinaryDiffEq writes it as dt/6 (2 (k2 + k3) + (k1 + k4)); same value up to the last bit).
// Two passes per launch: the AIRBORNE instance (GROUND = false: no ground-contact code) steps every lane; a lane that comes within
// 10 m of the terrain at any evaluation stops without committing anything and raises its redo flag; the GROUND-capable instance
// (same source, GROUND = true) then re-runs exactly those lanes from the same launch-start state (workgroups without such a lane
// leave at once). History: the first generations kept the stage state in registers (k_step<X, KIN, GROUND>, 108 of the 256 VGPRs,
// a fifth of its VALU instructions v_accvgpr moves, a per-lane stage machine); the ground-capable pass was the last user of that
// form (4.9e8 aircraft-steps/s on the ground; 7.9e8 on this design).
// The stage state lives in a third LDS panel and is read at the point of use; emit() updates it IN PLACE (every row's derivative is produced after the last
// read of that row within one evaluation — checked against rhs()'s source order), so neither array exists in registers.
// LDS room for the third panel comes from two facts about an airborne aircraft: its six contact-regulator states are
// identically zero (reset by f_step! whenever a wheel is off the ground, zero derivative at zero) so they need no rows —
// a lane that arrives with a non-zero one is handed to the ground-capable pass — and the eleven per-lane inputs fit in the
// registers the panels freed. 3 x 21 rows x 2 KB + 22 KB of tables = 151 KB.
#ifndef FB_AIR_SCALAR_KNOTS
#define FB_AIR_SCALAR_KNOTS true
#endif
// GROUND = true is the same kernel for the lanes the airborne pass hands over (second pass of a launch): all 27 rows and the
// ground-contact branch compiled in (see step_block() for its workgroup size).
template <int STRIDE, bool GROUND = false>
struct StateLds {
    lds_cptr p;   // &panel[lane]
    __device__ __forceinline__ static constexpr bool skip(int k) { return !GROUND && k >= FB_X_LDG_FRC && k < FB_X_LDG_FRC + 6; }
    __device__ __forceinline__ static constexpr int row(int k) { return GROUND ? k : (k < FB_X_LDG_FRC ? k : k - 6); }
    __device__ __forceinline__ double operator[](int k) const { return skip(k) ? 0.0 : p[row(k) * STRIDE]; }
};
// X = Cessna172X: the seven first-order actuators (ẋ = (cmd - x)/τ, commands constant during a launch) are linear, so classic
// RK4 on them has a closed form — stage values cmd + (x_n - cmd) m_s with m = 1, 1 - z/2, 1 - z/2 + z²/4, 1 - z + z²/2 - z³/4 and
// x_{n+1} = cmd + (x_n - cmd)(1 - z + z²/2 - z³/6 + z⁴/24), z = dt/τ — equal to the stage-by-stage evaluation up to rounding.
// They therefore need no panel rows at all (x_n and the commands ride in registers), and the Sv0 LDS budget holds for Xv2.
// The airborne kernels' emit: one branch-free body for all four stages (the stage only enters through wave-uniform operands):
//   A = acc + b k        b = 1, 2, 2, 1; acc is 0 whenever a stage 0 starts (zeroed at load and by every stage 3)
//   stage 0-2: acc <- A, x_eval <- x_n + c dt k;      stage 3: acc <- 0, x_n <- x_n + dt/6 A
// Bit-identical to the branched form (acc + 1 k = k exactly at stage 0, A 1 = A, same fma's) and to the ground-capable kernel's:
// near the ground a one-ulp difference in this combination is amplified to 1e-6 within a few hundred steps (tried:
// x_n + dt/6 acc + dt/6 k4 saves the select and fails test_approach_crosses_the_air_ground_handover).
// the stage sum k1 + 2 k2 + 2 k3: an LDS panel [rows][B], or (REGS, see step_acc_in_regs) NR registers per lane
template <int B, bool REGS> struct AccStore;
template <int B> struct AccStore<B, false> {
    lds_ptr l; int t;
    __device__ __forceinline__ double get(int r) const { return l[r * B + t]; }
    __device__ __forceinline__ void set(int r, double v) const { l[r * B + t] = v; }
};
template <int B> struct AccStore<B, true> {
    double* r_;   // the kernel's local array (every index is a compile-time constant after unrolling: registers)
    __device__ __forceinline__ double get(int r) const { return r_[r]; }
    __device__ __forceinline__ void set(int r, double v) const { r_[r] = v; }
};
template <int B, bool GROUND = false, bool ACC_REGS = false>
struct AirEmit {
    typedef void batched_tag;
    lds_cptr xs_l;            // x_n panel
    AccStore<B, ACC_REGS> acc;  // stage sum
    lds_ptr xwr_l;            // the panel this stage writes (evaluation panel, at stage 3 x_n itself)
    double eb, ee, em;
    bool last;
    int t;
    using SV = StateLds<B, GROUND>;
    __device__ __forceinline__ void operator()(int j, double kj) const {
        if (SV::skip(j)) return;   // identically zero in the air
        const int r = SV::row(j), idx = r * B + t;
        const double xs = xs_l[idx];
        const double A = __builtin_fma(eb, kj, acc.get(r));
        acc.set(r, A * em);
        xwr_l[idx] = __builtin_fma(ee, last ? A : kj, xs);
    }
    template <int NE>
    __device__ __forceinline__ void batch(int j0, const double (&k)[NE]) const {   // NE consecutive rows (no contact rows among them in the air)
        double xs[NE], A[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) { const int r = SV::row(j0 + e); xs[e] = xs_l[r * B + t]; A[e] = acc.get(r); }
#pragma unroll
        for (int e = 0; e < NE; e++) A[e] = __builtin_fma(eb, k[e], A[e]);
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int r = SV::row(j0 + e);
            acc.set(r, A[e] * em);
            xwr_l[r * B + t] = __builtin_fma(ee, last ? A[e] : k[e], xs[e]);
        }
    }
};
// Lanes per workgroup, and where the stage sum lives. The airborne instances: 256 lanes, three 21-row panels (151 KB). The
// ground-capable instances need all 27 rows: 256 lanes, two 27-row panels + the stage sum in 54 registers — Cessna172Sv0 450-458
// registers, no scratch; Cessna172Xv2 512 registers + ~1 KB of scratch.
// History: at full register pressure this LLVM places spill code before the exec restore of control-flow join blocks in some
// instances (tools/check_mir_spills.py, which the build enforces): lanes then reload garbage and results change from run to run
// (seen in the scripted crosswind landing, tools/det_check.py). With machine LICM on (the fp64 literal pairs hoisted out of the loop,
// see __graft_entry__.py) the 256-lane form tripped that check and every ground instance ran 192 lanes (7.9e8 aircraft-steps/s on a
// batch sitting on the ground); without it the Sv0 instances have registers to spare and no spill code at all. The Cessna172Xv2 instances
// stayed at 192 lanes (three panels in LDS: three one-wave SIMDs of four) with the gear units' ground contact behind calls until round 6:
// inlined (FB_X2_GROUND_CALLS = 0) and at 256 lanes they pass the check, and a batch on the ground steps 2.9 x as fast
// (4.31e8 -> 8.95e8 -> 1.24e9 aircraft-steps/s, profiles/r06_ab_x2_ground_inline.txt).
#ifndef FB_GROUND_BLOCK_X
#define FB_GROUND_BLOCK_X 256
#endif
#ifndef FB_GROUND_BLOCK_S
#define FB_GROUND_BLOCK_S 256
#endif
template <bool X, bool GROUND> constexpr int step_block() { return GROUND ? (X ? FB_GROUND_BLOCK_X : FB_GROUND_BLOCK_S) : STEP_BLOCK; }
template <bool X, bool GROUND> constexpr bool step_acc_in_regs() { return GROUND && step_block<X, GROUND>() > 192; }
#ifndef FB_STEP_ATTR
#define FB_STEP_ATTR
#endif
// diagnostic builds (-DFB_STAMP, tools/stamp_ground_launch.py): where a launch of the ground-capable Cessna172Xv2 pass spends its cycles outside the
// evaluations — slots 13 entry, 14 tables staged, 15 x_n in LDS, 30 loop entered (inputs, commands, the carried k1), 31 state written back
#ifdef FB_STAMP
#define FB_LAUNCH_STAMP(k) do { if constexpr (X && GROUND) fb_stamp(k); } while (0)
#else
#define FB_LAUNCH_STAMP(k) do { } while (0)
#endif
template <int KIN, bool X = false, bool GROUND = false, bool PERENV = false>
__global__ __launch_bounds__((step_block<X, GROUND>())) FB_STEP_ATTR void k_step_air(KArgs a, int nsteps) {
    constexpr int B = step_block<X, GROUND>(), NR = GROUND ? (int)FB_NX : FB_NX - 6;
    constexpr bool ACC_REGS = step_acc_in_regs<X, GROUND>();
    using SV = StateLds<B, GROUND>;
    using InT = typename std::conditional<X, InputsXAgg, InputsAgg>::type;
    __shared__ double lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ double rk[LDS_RK_DOUBLES];
    __shared__ double xs_l[NR * B];    // x_n
    __shared__ double acc_l[ACC_REGS ? 1 : NR * B];   // k1 + 2 k2 + 2 k3 (an LDS panel, or the registers acc_r: see step_acc_in_regs)
    __shared__ double xc_l[NR * B];    // the state being evaluated (x_n, or x_n + c dt k_j), updated in place by emit()
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (GROUND) {   // second pass: only the lanes the airborne pass handed over
        const int mine = (i < a.n) && a.redo[i] != 0;
        if (!__syncthreads_or(mine)) return;
        FB_LAUNCH_STAMP(13);
        stage_tables<PR_NC_STEP>(lds, rk, a.tables);
        FB_LAUNCH_STAMP(14);
        if (!mine) return;
        a.redo[i] = 0;
    } else {
        stage_tables<PR_NC_STEP>(lds, rk, a.tables);
        if (i >= a.n) return;
    }
    if (a.status[i] != 0) return;
    const int t = threadIdx.x;
    const Env env = [&] { if constexpr (PERENV) return env_of(a, i); else return a.env; }();   // per lane (VGPRs) or the batch-wide block (SGPRs)
    double acc_r[ACC_REGS ? NR : 1];
    const AccStore<B, ACC_REGS> acc = [&] { if constexpr (ACC_REGS) return AccStore<B, true>{acc_r}; else return AccStore<B, false>{(lds_ptr)acc_l, t}; }();
    // ---- termination (FC/sim.jl:561-570; include/flightbatch.h FB_TERM_*) ----
    // The reference stops at the first exception and leaves mdl.x as it stands at the throw: for an f_ode! that threw at an RK stage,
    // that stage's ARGUMENT. An evaluation here updates its argument panel in place, so by the time its status bits are known the
    // argument is gone. The airborne instance therefore hands every lane that raises a bit over (nothing committed, like a lane that
    // comes within reach of the ground); the ground-capable instance notes WHERE the lane threw (tkey = completed steps * 8 + FB_TERM_*
    // code) and, once the wave is through its launch, steps the lane again from the launch-start state — same inputs, same code, bit
    // for bit the same path — up to the evaluation that threw, and freezes it in front of it (`replaying`, below).
    constexpr int TKEY_NONE = 0x7fffffff;
    [[maybe_unused]] int tkey = TKEY_NONE;     // per lane
    [[maybe_unused]] bool replaying = false;   // wave-uniform
    bool mine = true;                          // per lane: this pass steps the lane
    // What a lane carries from the loop into the epilogue is declared ABOVE the replay's entry point and (re)loaded for the replayed lanes
    // only: the other lanes of the wave — the ones that finished the launch, or crashed in f_step! — sit the second pass out and must
    // write back what THEIR launch left (stall flag, engine state, actuator positions, steps completed), not the launch-start values.
    constexpr int NAL = GROUND ? (int)FB_NACT : (int)FB_ACT_BRAKE_LEFT;   // actuators tracked through the stages (on the ground the brakes are inputs of the RHS like the rest)
    static_assert(FB_ACT_BRAKE_LEFT == 5 && FB_ACT_BRAKE_RIGHT == 6 && FB_NACT == 7, "brakes are the last two actuators");
    double xa[X ? NAL : 1], ca[X ? NAL : 1];
    int steps_alive = 0;
    int stall = 0, eng = 0;
    // (-DFB_REPLAY_BYSTANDER_DEFECT rebuilds the defect this guards against — every lane of a replayed wave reloading the launch-start
    // values — so that the regression tests can be seen to fail on it: tests/test_gpu_termination.py, profiles/r04_replay_defect.txt)
#ifdef FB_REPLAY_BYSTANDER_DEFECT
#define FB_REPLAY_MINE true
#else
#define FB_REPLAY_MINE mine
#endif
restart:
    // The lane's index, opaque to the optimiser from here to the loop: the ~100 row addresses of this block (state, carried k1, record copies, commands)
    // are invariant in the replay loop that `restart` heads, and hoisted in front of it they were all live across it — 83 of them spilled in the
    // ground-capable Cessna172Xv2 instance and reloaded one by one, each row's load waiting for its address: load, wait, store, 27 + 27 times
    // (tools/stamp_ground_launch.py: 36 k + 27 k cycles of a one-step launch's 286 k).
    int64_t il = i;
    asm volatile("" : "+v"(il));
    if (mine) {
        bool to_ground = false;
        // nine rows in flight at a time (left to itself the scheduler pairs them: load, load, wait, store — thirteen round trips to memory in a row)
        static_assert(FB_NX % 9 == 0, "row batches");
#pragma unroll
        for (int k0 = 0; k0 < FB_NX; k0 += 9) {
            double v[9];
#pragma unroll
            for (int g = 0; g < 9; g++) v[g] = a.x[(int64_t)(k0 + g) * a.n + il];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 9; g++) {
                const int k = k0 + g;
                if (SV::skip(k)) to_ground = to_ground || (v[g] != 0.0);
                else { xs_l[SV::row(k) * B + t] = v[g]; acc.set(SV::row(k), 0.0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (to_ground) { a.redo[il] = 1; return; }   // (airborne instance only: the ground-capable one skips no row)
    }
    FB_LAUNCH_STAMP(15);
    if constexpr (X) {
        // the control laws run inside this launch and rewrite cs / cu; a lane that is later handed to the ground-capable pass, or stepped
        // a second time up to the evaluation that threw, must find them as they were at launch start
        // (a launch of ONE step needs neither the copy nor the way back: its only control update is the last thing it does — behind the step's last
        // evaluation and f_step!, with no re-evaluation of k1 behind it in this launch — so a lane that is handed over, or that throws, has not had
        // one. At one step per launch, the shape of every host callback and of a scenario table evaluated after every step, the two copies were
        // 1.5 KB per aircraft and launch, as much again as the state and the record the step itself moves.)
        if (a.ctl_ratio > 0 && nsteps > 1 && mine) {
            if (!replaying) {
                copy_rows_batched<FB_NCS, 11>(a.ctl_bak, a.cs, a.n, il);
                copy_rows_batched<FB_NCU, 14>(a.ctl_bak + (int64_t)FB_NCS * a.n, a.cu, a.n, il);
            } else {
                copy_rows_batched<FB_NCS, 11>(a.cs, a.ctl_bak, a.n, il);
                copy_rows_batched<FB_NCU, 14>(const_cast<double*>(a.cu), a.ctl_bak + (int64_t)FB_NCS * a.n, a.n, il);
            }
        }
    }
    InT in;
    // actuator positions x_n and commands. Only the five the airborne RHS reads are tracked through the stages; the two brake
    // actuators are advanced at the end, over the steps this lane completed, with the same closed form.
    if (FB_REPLAY_MINE) steps_alive = 0;
    if constexpr (X) {
        if (FB_REPLAY_MINE) {
            double cr[NAL];
#pragma unroll
            for (int k = 0; k < NAL; k++) { xa[k] = a.x[(int64_t)(X2_ACT + k) * a.n + il]; cr[k] = x2_command_row(a, il, k); }
            __builtin_amdgcn_sched_barrier(0);   // (all fourteen rows requested before the first saturation waits for its own)
#pragma unroll
            for (int k = 0; k < NAL; k++) ca[k] = x2_command_sat(k, cr[k]);
        }
        in.xa = nullptr; in.u_glob = a.u + il; in.n = a.n; in.ui = a.ui[il];
        sum_payload_of(in);
    } else {
        load_inputs(a, il, in);
        if constexpr (GROUND) in.load_ground_inputs();
        in.u_glob = nullptr;   // (ground-only inputs are never read in the air, and the ground-capable pass has just fetched them)
        in.sum_payload();
        in.sum_aero((lds_cptr)lds + LDS_AERO, (lds_cptr)rk + LDS_AERO);
    }
    if (FB_REPLAY_MINE) { stall = a.s[il]; eng = a.s[a.n + il]; }
    const double dt = a.dt, hdt = a.dt / 2, dt6 = a.dt / 6;
    const double z = dt / ACT_TAU;
    // The loop below is a WAVE-uniform state machine: stage and step live in SGPRs, so the three-way choice inside emit() is a
    // scalar branch (2 SALU instructions) and not an exec-mask dance (14 scalar instructions per emitted row when the stage
    // was per-lane). The price: when f_step! modifies a lane's state (renormalisation, stall flag, engine state) and k1 has to
    // be re-evaluated for it, the other lanes of its wave sit out that one evaluation (`run` false) instead of moving on.
    int stage = 0, step = 0;
    bool pending_cb = false, redoing = false;      // uniform
    bool alive = mine, dead = false, run = mine, handoff = false;   // per lane
    if constexpr (X) {
        // FSAL across launches: the previous launch's last evaluation sat at this very state. A wave whose lanes all
        // hold a valid k1 starts at stage 1; otherwise the lanes without one evaluate it first while the others sit out.
        const bool have_k1 = mine && a.k1 && a.k1_valid[il];
        if (have_k1) {
#pragma unroll
            for (int j0 = 0; j0 < FB_NX; j0 += 9) {   // (nine rows in flight at a time, like x_n above)
                double kv[9];
#pragma unroll
                for (int g = 0; g < 9; g++) kv[g] = SV::skip(j0 + g) ? 0.0 : a.k1[(int64_t)(j0 + g) * a.n + il];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 9; g++) {
                    const int j = j0 + g;
                    if (SV::skip(j)) continue;
                    const int idx = SV::row(j) * B + t;
                    acc.set(SV::row(j), kv[g]);
                    xc_l[idx] = xs_l[idx] + hdt * kv[g];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (__builtin_amdgcn_ballot_w64(mine && !have_k1) == 0) stage = 1;
        else { run = mine && !have_k1; redoing = true; }
    }
    FB_LAUNCH_STAMP(30);
#pragma unroll 1
    while (true) {
        StepAux aux;
        int lds_off = 0;
        asm volatile("" : "+s"(lds_off));   // an opaque zero offset: without it LICM hoists every loop-invariant table / input load (~150 values) into VGPRs: 1.7 KB/lane of spills
        const Tables T = {(lds_cptr)lds + lds_off, a.egm96, (lds_cptr)rk + lds_off, (gk_cptr)a.tables + lds_off};
        const bool last = stage == 3;
        const double eb = (stage == 1 || stage == 2) ? 2.0 : 1.0, ee = last ? dt6 : (stage == 2 ? dt : hdt), em = last ? 0.0 : 1.0;
        // panel roles: stage 0 evaluates x_n straight from xs_l, stages 1-3 the state the previous stage left in xc_l; stages 0-2
        // write the next evaluation state to xc_l, stage 3 the new x_n to xs_l (each row read before it is overwritten)
        const lds_cptr xrd_l = stage == 0 ? (lds_cptr)xs_l : (lds_cptr)xc_l;
        const lds_ptr xwr_l = last ? (lds_ptr)xs_l : (lds_ptr)xc_l;
        int32_t bits = 0;
        // (the tap's twelve values: in an LDS panel where the LDS has room — the ground-capable instances at 256 lanes —, in registers otherwise)
        constexpr bool TAP_LDS = X && GROUND && B == 256;
        __shared__ double tap_l[TAP_LDS ? 12 * B : 1];
        [[maybe_unused]] typename std::conditional<TAP_LDS, CtlSinkLds<B>, CtlSinkOpt>::type tap;
        tap.on = false;
        if constexpr (TAP_LDS) tap.base = (lds_ptr)tap_l + t;
        [[maybe_unused]] const bool tap_now = X && a.ctl_ratio > 0 && stage == 0 && pending_cb && !redoing && (a.ctl_phase + step + 1) % a.ctl_ratio == 0;   // wave-uniform
        // which evaluation of the reference's schedule this is (wave-uniform): the one at the new state of an RK update that has just
        // been made — `step` counts the callbacks that have run, so one update more is complete — a re-evaluation at the accepted state, or
        // RK stage k2..k4 (FB_TERM_* codes)
        [[maybe_unused]] const bool at_new = stage == 0 && pending_cb && !redoing;
        [[maybe_unused]] const int ev_done = step + (at_new ? 1 : 0);
        [[maybe_unused]] const int ev_key = ev_done * 8 + (stage != 0 ? stage + 1 : (at_new ? (int)FB_TERM_F_ODE_NEW : (int)FB_TERM_F_ODE_REEVAL));
        static_assert(FB_TERM_F_ODE_K2 == 2 && FB_TERM_F_ODE_K3 == 3 && FB_TERM_F_ODE_K4 == 4, "stage + 1");
        if constexpr (GROUND) {
            if (replaying && run && tkey == ev_key) {
                // this is the evaluation that threw: the lane stops in front of it, with mdl.x = its argument (FC/sim.jl:306)
                if (stage != 0) {
#pragma unroll
                    for (int r = 0; r < NR; r++) xs_l[r * B + t] = xc_l[r * B + t];
                    if constexpr (X) {
                        const double ms = act_stage_mc(stage, z);
#pragma unroll
                        for (int k = 0; k < NAL; k++) xa[k] = act_stage_pos(xa[k], ca[k], ms, false);   // (the argument the evaluation was given: the same expression)
                    }
                }
                if constexpr (X) { if (a.k1) a.k1_valid[i] = 0; }
                alive = false; run = false; dead = true;
            }
        }
        if (run) {
            InT inl = in;                   // and keeps products of the per-lane inputs from being hoisted out of it
            double xa_s[X ? FB_NACT : 1];
            if constexpr (X) {
                const double ms = act_stage_mc(stage, z);
#pragma unroll
                for (int k = 0; k < NAL; k++) xa_s[k] = act_stage_pos(xa[k], ca[k], ms, stage == 0);
                if constexpr (!GROUND) { xa_s[FB_ACT_BRAKE_LEFT] = 0; xa_s[FB_ACT_BRAKE_RIGHT] = 0; }   // (never read in the air)
                inl.xa = xa_s;
                inl.u_glob = in.u_glob + lds_off;
            } else {
                asm volatile("" : "+v"(inl.throttle), "+v"(inl.mixture));
            }
            const AirEmit<B, GROUND, ACC_REGS> emit = {(lds_cptr)xs_l, acc, xwr_l, eb, ee, em, last, t};
            const SV xv = {xrd_l + t + lds_off};
            if constexpr (X) {
                // the evaluation at x_{n+1} of a step that closes a control period is the "last f_ode!" whose outputs the control
                // laws read: the partial sink is switched on for it (ONE instance of rhs() in the loop: CtlSinkOpt)
                tap.on = tap_now;
                bits = rhs<KIN, GROUND, FB_AIR_SCALAR_KNOTS>(xv, stall, eng, inl, env, T, emit, aux, tap);
            } else
                bits = rhs<KIN, GROUND, FB_AIR_SCALAR_KNOTS>(xv, stall, eng, inl, env, T, emit, aux, NoSink{});
            if constexpr (!GROUND) {
                // within reach of the ground, or an exception: nothing is committed for this lane, the ground-capable pass takes it over
                if (bits != 0) { handoff = true; alive = false; run = false; bits = 0; }
            } else {
                if (bits != 0) {   // f_ode! threw (altitude / ISA range, contact assertion): the simulation of this aircraft ends HERE
                    if (!replaying) {
                        tkey = ev_key;
                        a.status[i] |= first_exception(bits);
                        a.term_where[i] = ev_key & 7;
                        a.term_step[i] = a.step0 + ev_done;
                    }
                    // (second pass: the lane follows the first pass bit for bit and is frozen in front of this evaluation, so nothing
                    // can be raised there; a lane that did would simply stop as it stands)
                    alive = false; run = false; bits = 0;
                }
            }
            if constexpr (X) {
                if (last && run) {
                    const double P = 1 - z + z * z / 2 - z * z * z / 6 + z * z * z * z / 24;
#pragma unroll
                    for (int k = 0; k < NAL; k++) xa[k] = ca[k] + (xa[k] - ca[k]) * P;
                    steps_alive++;
                }
            }
        }
        if (redoing) { redoing = false; run = alive; }   // the lanes that sat out the re-evaluation of k1 join again
        else if (stage == 0 && pending_cb) {
            // f_step! on x_{n+1}, which sits in xs_l (this evaluation's emits have already moved xc_l on to the next stage)
            pending_cb = false;
            step++;
            bool mod = false;
            // Cessna172X: f_periodic! follows f_step! (FC/sim.jl:204-218) but reads the outputs of the step's last f_ode!, i.e. of this
            // evaluation at x_{n+1}, taken before f_step! renormalises the quaternions: tap them first
            // (a lane that f_step! is about to terminate — a crash flag — gets no update: the reference throws out of cb_step before
            // cb_periodic runs; a lane whose evaluation threw is not running any more)
            const bool ctl_now = X && a.ctl_ratio > 0 && (a.ctl_phase + step) % a.ctl_ratio == 0;   // wave-uniform
            if (run) {
                if constexpr (X) {
                    if (ctl_now && !aux.crash) {
                        CtlIn v;
                        {
                            const CtlSink tv = tap_values(tap);
                            v.lat = tv.lat; v.lon = tv.lon;
                            v.EAS = tv.EAS; v.theta = tv.theta; v.phi = tv.phi; v.clm = -tv.vd; v.chi = tv.chi;
                            v.w_wb_b = {tv.wx, tv.wy, tv.wz};
                            v.alpha = tv.alpha; v.beta = tv.beta;
                        }
                        auto XS = [&](int k) { return xs_l[SV::row(k) * B + t]; };   // x_{n+1}, before f_step! touches it
                        v.h_e = XS(h_e_row<KIN>());
                        v.w_eb_b = {XS(FB_X_OMEGA_EB_B), XS(FB_X_OMEGA_EB_B + 1), XS(FB_X_OMEGA_EB_B + 2)};
                        v.alpha_filt = XS(FB_X_ALPHA_FILT); v.beta_filt = XS(FB_X_BETA_FILT);
                        v.n_eng = XS(FB_X_ENG_OMEGA) / c172::w_rated;
#pragma unroll
                        for (int k = 0; k < 4; k++) { v.pos[k] = clampd(xa[k], k == FB_ACT_THROTTLE ? 0.0 : -1.0, 1.0); v.cmd[k] = ca[k]; }   // (InputsX::pos)
                        v.on_gnd = GROUND && aux.wow != 0;   // is_on_gnd: any strut with weight on wheels (c172.jl:998-1001)
                        auto pk = [&](int k) { return (uint32_t)a.ctl_off.off[k] | ((uint32_t)a.ctl_off.off[k + 1] << 16); };
                        const CtlOut co = x2_periodic(a.cu, a.cs, a.gains, a.n, a.ctl_dT, pk(0), pk(2), pk(4), pk(6), pk(8),
                                                      (uint32_t)a.ctl_off.total | ((uint32_t)a.ctl_off.same_grid << 16), i, v);
                        FB_X2_STAMP(26);
                        static_assert(FB_ACT_THROTTLE == 0 && FB_ACT_AILERON == 1 && FB_ACT_ELEVATOR == 2 && FB_ACT_RUDDER == 3, "CtlOut order");
#pragma unroll
                        for (int k = 0; k < 4; k++) ca[k] = co.cmd[k];   // the commands in force from the next stage on (flaps and brakes are inputs: unchanged)
                    }
                }
                auto renorm = [&](int k0, int len) {   // normalize_block!(v, 1e-8), kinematics.jl:114-118; WA :226-229, ECEF :317-320, NED: none
                    double q[4] = {0, 0, 0, 0}, n2 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < len) { q[k] = xs_l[SV::row(k0 + k) * B + t]; n2 += q[k] * q[k]; }
                    const double nr = ::sqrt(n2);
                    if (fabs(nr - 1.0) > 1e-8) {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (k < len) xs_l[SV::row(k0 + k) * B + t] = q[k] / nr;
                        mod = true;
                    }
                };
                if constexpr (KIN == FB_KIN_WA) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_EW, 4); }
                else if constexpr (KIN == FB_KIN_ECEF) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_WB + 4, 3); }
                const int stall0 = stall, eng0 = eng;
                if (aux.alpha > c172::alpha_stall_hi) stall = 1;
                else if (aux.alpha < c172::alpha_stall_lo) stall = 0;
                [[maybe_unused]] bool crashed = false;
                if constexpr (GROUND) {
                    // landing gear, unit by unit (c172.jl:485; landinggear.jl:539-548): a strut's crash check THROWS GroundCrash, and what
                    // follows it in the reference's order — this unit's regulator reset, the later units, the engine's state machine,
                    // f_periodic! — does not run; the friction regulators of a unit without weight on its wheel are reset (:479-483)
#pragma unroll
                    for (int g = 0; g < 3; g++) {
                        if (crashed) continue;
                        if (aux.crash & (1 << g)) { crashed = true; continue; }
                        if (!(aux.wow & (1 << g))) {
                            const int i0 = SV::row(FB_X_LDG_FRC + 2 * g) * B + t, i1 = SV::row(FB_X_LDG_FRC + 2 * g + 1) * B + t;
                            if (xs_l[i0] != 0.0 || xs_l[i1] != 0.0) mod = true;
                            xs_l[i0] = 0.0; xs_l[i1] = 0.0;
                        }
                    }
                }
                if (!crashed) {
                    const double w = xs_l[SV::row(FB_X_ENG_OMEGA) * B + t];
                    const bool fuel = aux.m_avail > 0;
                    const bool start = in.ui & FB_UI_ENG_START, stop = in.ui & FB_UI_ENG_STOP;
                    if (eng == 0) { if (start) eng = 1; }
                    else if (eng == 1) { if (!start) eng = 0; if (w > c172::w_idle && fuel) eng = 2; }
                    else if (stop || w < c172::w_stall || !fuel) eng = 0;
                }
                mod = mod || stall != stall0 || eng != eng0;
                if constexpr (GROUND) {
                    if (crashed) {   // the simulation of this aircraft ends inside f_step! (FB_TERM_F_STEP), `step` RK updates complete
                        a.status[i] |= FB_ST_GROUND_CRASH;
                        a.term_where[i] = FB_TERM_F_STEP;
                        a.term_step[i] = a.step0 + step;
                        dead = true;
                    }
                }
                if constexpr (X) {
                    if (a.k1 && (dead || step == nsteps)) {   // this evaluation's derivatives (acc = k at stage 0) are the next launch's k1 unless something changed
                        const bool keep = !dead && !mod;
                        int64_t ik = i;
                        asm volatile("" : "+v"(ik));   // (the 27 row addresses formed here, not carried — spilled — through the whole loop: see `il`)
                        if (keep) {
#pragma unroll
                            for (int j = 0; j < FB_NX; j++)
                                a.k1[(int64_t)j * a.n + ik] = SV::skip(j) ? 0.0 : acc.get(SV::row(j));
                        }
                        a.k1_valid[ik] = keep ? 1 : 0;
                    }
                }
                if (dead) { alive = false; run = false; mod = false; }
            }
            if (step == nsteps || __builtin_amdgcn_ballot_w64(alive) == 0) {
                if constexpr (GROUND) {
                    // lanes whose f_ode! threw are stepped once more, up to that evaluation (see `tkey` above)
                    if (!replaying && __builtin_amdgcn_ballot_w64(tkey != TKEY_NONE) != 0) {
                        replaying = true;
                        mine = tkey != TKEY_NONE;
                        goto restart;
                    }
                }
                break;
            }
            if (__builtin_amdgcn_ballot_w64(mod) != 0) {   // k1 must be re-evaluated on the modified x_{n+1}
                if (mod) {
#pragma unroll
                    for (int r = 0; r < NR; r++) acc.set(r, 0.0);   // (acc held the discarded k1; stage 0 reads x_n from xs_l itself)
                }
                run = mod; redoing = true;
                continue;
            }
        }
        stage = (stage + 1) & 3;
        pending_cb = (stage == 0);
    }
    if (!GROUND && handoff) {
        if constexpr (X) {
            if (a.ctl_ratio > 0 && nsteps > 1) {   // nothing of this lane's launch is committed: undo the control-law updates it has made (one step: it has made none)
#pragma unroll 1
                for (int k = 0; k < FB_NCS; k++) a.cs[(int64_t)k * a.n + i] = a.ctl_bak[(int64_t)k * a.n + i];
#pragma unroll 1
                for (int k = 0; k < FB_NCU; k++) const_cast<double*>(a.cu)[(int64_t)k * a.n + i] = a.ctl_bak[(int64_t)(FB_NCS + k) * a.n + i];
            }
        }
        a.redo[i] = 1;
        return;
    }
    bool bad = false;
    int64_t ie = i;
    asm volatile("" : "+v"(ie));   // (the write-back's row addresses formed here: see `il`)
#pragma unroll
    for (int k = 0; k < FB_NX; k++) {
        if (SV::skip(k)) continue;
        const double v = xs_l[SV::row(k) * B + t];
        bad = bad || !isfinite(v);
        a.x[(int64_t)k * a.n + ie] = v;
    }
    if constexpr (X) {
#pragma unroll
        for (int k = 0; k < NAL; k++) { bad = bad || !isfinite(xa[k]); a.x[(int64_t)(X2_ACT + k) * a.n + ie] = xa[k]; }
        const double P = 1 - z + z * z / 2 - z * z * z / 6 + z * z * z * z / 24;
#pragma unroll
        for (int k = NAL; k < FB_NACT; k++) {
            const double c = x2_command(a, ie, k), x0 = a.x[(int64_t)(X2_ACT + k) * a.n + ie];
            double v = x0;
            for (int m = 0; m < steps_alive; m++) v = c + (v - c) * P;   // step by step: bit-identical to the per-step update
            bad = bad || !isfinite(v);
            a.x[(int64_t)(X2_ACT + k) * a.n + ie] = v;
        }

    }
    if (bad) a.status[ie] |= FB_ST_NAN;
    a.s[ie] = stall;
    a.s[a.n + ie] = eng;
    FB_LAUNCH_STAMP(31);
}

// ---- the same update in two halves, for the wave-specialised stepper (k_step_duo<KIN, true>) ---------------------------------------
// Two waves serve an aircraft there, and an update on ONE of them is a 36 k-cycle dependent chain (profiles/r03_x2_update_stamps.txt) with
// the other idle. The longitudinal laws (c172x_ctl.jl:286-446: throttle, elevator) and the lateral ones (:880-983: aileron, rudder) are
// independent once the guidance has run, so role P's wave runs x2_periodic_lon and role D's x2_periodic_lat, side by side. What they read
// of vehicle.y was tapped from the step's last f_ode! by BOTH roles (each the outputs its own part of the evaluation forms) into the rows
// of KArgs::duo_tap; each half runs the guidance for itself (CtlMemHalfT says who writes what to memory). Same arguments-in-registers
// discipline as x2_periodic; the state rows a half needs it reads from the x_n panel in LDS (x_{n+1}, before f_step! touches it).
enum { DUO_TAP_THETA = 0, DUO_TAP_PHI, DUO_TAP_WX, DUO_TAP_WY, DUO_TAP_WZ, DUO_TAP_VD, DUO_TAP_CHI, DUO_TAP_EAS, DUO_TAP_ALPHA, DUO_TAP_BETA,   // role D's
       DUO_TAP_LAT, DUO_TAP_LON, DUO_TAP_POS, DUO_TAP_CMD = DUO_TAP_POS + 4, DUO_NTAP = DUO_TAP_CMD + 4 };                                  // role P's
static_assert(DUO_TAP_THETA == DUO_TAP_THETA_ROW && DUO_TAP_PHI == DUO_TAP_THETA_ROW + 1 && DUO_TAP_WX == DUO_TAP_WX_ROW && DUO_TAP_WZ == DUO_TAP_WX_ROW + 2 &&
              DUO_TAP_VD == DUO_TAP_VD_ROW && DUO_TAP_CHI == DUO_TAP_VD_ROW + 1 && DUO_TAP_EAS == DUO_TAP_EAS_ROW && DUO_TAP_ALPHA == DUO_TAP_ALPHA_ROW &&
              DUO_TAP_BETA == DUO_TAP_ALPHA_ROW + 1 && DUO_TAP_LAT == DUO_TAP_LAT_ROW && DUO_TAP_LON == DUO_TAP_LAT_ROW + 1, "rhs_duo's tap rows");
struct Ctl2 { double c0, c1; };
struct CtlHalfArgs {   // (unpacked from the scalar arguments inside the callee: see x2_periodic)
    const double* cu; double* cs; const double* gains; const double* tap;
    int64_t n; double dT; CtlOffsets off;
};
FBD CtlHalfArgs ctl_half_args(const double* a_cu, double* a_cs, const double* a_gains, const double* a_tap, int64_t a_n, double a_dT, uint32_t o01, uint32_t o23,
                              uint32_t o45, uint32_t o67, uint32_t o89, uint32_t tsg) {
    CtlHalfArgs h;
    h.cu = uni(a_cu); h.cs = uni(a_cs); h.gains = uni(a_gains); h.tap = uni(a_tap); h.n = uni(a_n); h.dT = uni(a_dT);
    const uint32_t op[5] = {o01, o23, o45, o67, o89};
#pragma unroll
    for (int k = 0; k < 10; k++) h.off.off[k] = uni((int)((op[k / 2] >> (16 * (k % 2))) & 0xffffu));
    h.off.total = uni((int)(tsg & 0xffffu));
    h.off.same_grid = uni((int)(tsg >> 16));
    return h;
}
#ifdef FB_STAMP
// diagnostic builds (tools/stamp_x2_duo.py): per-phase cycles of the two halves of an update, wave 0 (longitudinal, slots 21-25 + ctl_lon's 27-29)
// and wave 4 (lateral, slots 16-20) of workgroup 0
__device__ unsigned long long g_half_last[2];
template <int HALF> __device__ __forceinline__ void half_stamp(int k) {
    __builtin_amdgcn_sched_barrier(0);
    if (blockIdx.x == 0 && (threadIdx.x & 255) < 64) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 255) == 0) {
            if (k >= 0) { g_stamp_acc[k] += t - g_half_last[HALF - 1]; g_stamp_cnt[k] += 1; }
            g_half_last[HALF - 1] = __builtin_amdgcn_s_memtime();
            if (HALF == CTL_HALF_LON) g_stamp_last = g_half_last[0];   // (ctl_lon's own fences measure from here)
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define FB_HALF_STAMP(HALF, k) half_stamp<HALF>(k)
#else
#define FB_HALF_STAMP(HALF, k) do { } while (0)
#endif
// xs: &x_n panel[lane] of the calling workgroup (rows of DUO_B lanes, the airborne row numbering); h_e: the mechanisation's altitude state
// Inlined into k_step_duo<KIN,true>: as an out-of-line callee each half saved and restored the callee-saved VGPR groups it used through scratch
// (~170 stores + 170 loads per pair-update, ~9 GB each way per launch); inlined the kernel holds 249 registers and has no scratch frame at all.
// Same-box alternating A/B (profiles/r04_x2_inline_ab.txt): ratio 2 10.90 -> 10.37 ms, ratio 1 13.9 -> 12.5 ms, ratio 50 8.25 -> 8.35 ms.
#ifndef FB_X2_HALF_ATTR
#define FB_X2_HALF_ATTR __forceinline__
#endif
// FB_X2_SPLIT_PITCH = 1 (round 6; OFF: measured, no gain): the longitudinal half is the long one of an update — 27 k cycles
// against the lateral half's 14 k, profiles/r05_x2_half_stamps.txt — so its pitch-axis outer loops (c2θ PID, θ -> q, q2e integrator + PID: 5 k of its
// 8 k of outer loops) run on the LATERAL half's wave, ahead of the lateral laws, and their elevator reference reaches the te2te LQR — restructured so
// that everything that does not need it comes first (lqr_run_g_late) — through an LDS slot and a fifth point Z of the pair's counters. Correct
// (26 GPU tests, 2.5e-12 against the oracle) and NOT faster: 9.96-9.99 ms per launch against 9.84-9.86, at every combination of issue priorities
// (profiles/r06_ab_x2_split_pitch.txt). The two halves share one SIMD's issue slots: an update costs what the pair's ~4 900 instructions cost,
// whichever wave issues them — the lateral wave's "waiting" was never idle time that work could be moved into. Kept for the record.
#ifndef FB_X2_SPLIT_PITCH
#define FB_X2_SPLIT_PITCH 0
#endif
// hand: how the pitch-axis outer loops' elevator reference crosses from the lateral half's wave to the longitudinal half's (k_step_duo: an LDS
// slot and the pair's counters): hand.put(v) in the lateral half — store, then publish —, hand.get() in the longitudinal one — wait, then load
template <int HALF, class Hand>
__device__ FB_X2_HALF_ATTR Ctl2 x2_periodic_half(const double* a_cu, double* a_cs, const double* a_gains, const double* a_tap, int64_t a_n, double a_dT, uint32_t o01,
                                              uint32_t o23, uint32_t o45, uint32_t o67, uint32_t o89, uint32_t tsg, int64_t i, double h_e, lds_cptr xs, const Hand& hand) {
    constexpr int B = 256;   // (= DUO_B, defined below)
    using SV = StateLds<B, false>;
    FB_HALF_STAMP(HALF, -1);
    const CtlHalfArgs A = ctl_half_args(a_cu, a_cs, a_gains, a_tap, a_n, a_dT, o01, o23, o45, o67, o89, tsg);
    const int64_t n = A.n;
    typedef __attribute__((address_space(1))) double* gptr;
    typedef ctlg_cptr gcptr;
    double lu[FB_NCU];
    const gcptr t0 = (gcptr)(uintptr_t)A.tap + i;
    auto TAP = [&](int k) { return t0[(int64_t)k * n]; };
    auto XS = [&](int k) { return xs[SV::row(k) * B]; };
    CtlIn v;
    {   // one burst: the tapped outputs and the input rows in flight together (the compiler drops the rows this half never reads)
        const gcptr u0 = (gcptr)(uintptr_t)A.cu + i;
        v.lat = TAP(DUO_TAP_LAT); v.lon = TAP(DUO_TAP_LON); v.EAS = TAP(DUO_TAP_EAS); v.phi = TAP(DUO_TAP_PHI);
        if constexpr (HALF == CTL_HALF_LON) {
            v.theta = TAP(DUO_TAP_THETA); v.clm = -TAP(DUO_TAP_VD); v.alpha = TAP(DUO_TAP_ALPHA);
            v.w_wb_b = {0.0, TAP(DUO_TAP_WY), TAP(DUO_TAP_WZ)};
            v.pos[0] = TAP(DUO_TAP_POS + 0); v.pos[2] = TAP(DUO_TAP_POS + 2); v.cmd[0] = TAP(DUO_TAP_CMD + 0); v.cmd[2] = TAP(DUO_TAP_CMD + 2);
            v.pos[1] = v.pos[3] = v.cmd[1] = v.cmd[3] = 0; v.chi = 0; v.beta = 0;
        } else {
            v.chi = TAP(DUO_TAP_CHI); v.beta = TAP(DUO_TAP_BETA);
            v.theta = 0; v.clm = 0;
            v.w_wb_b = {TAP(DUO_TAP_WX), 0.0, 0.0};
            if constexpr (FB_X2_SPLIT_PITCH != 0) {   // (+ pitch and yaw rate: what the pitch-axis outer loops read, ctl_lon_pitch_half)
                v.theta = TAP(DUO_TAP_THETA); v.clm = -TAP(DUO_TAP_VD);
                v.w_wb_b = {TAP(DUO_TAP_WX), TAP(DUO_TAP_WY), TAP(DUO_TAP_WZ)};
            }
            v.pos[1] = TAP(DUO_TAP_POS + 1); v.pos[3] = TAP(DUO_TAP_POS + 3); v.cmd[1] = TAP(DUO_TAP_CMD + 1); v.cmd[3] = TAP(DUO_TAP_CMD + 3);
            v.pos[0] = v.pos[2] = v.cmd[0] = v.cmd[2] = 0; v.alpha = 0;
        }
#pragma unroll
        for (int k = 0; k < FB_NCU; k++) lu[k] = u0[(int64_t)k * n];
    }
    v.h_e = h_e;
    v.w_eb_b = {XS(FB_X_OMEGA_EB_B), XS(FB_X_OMEGA_EB_B + 1), XS(FB_X_OMEGA_EB_B + 2)};
    v.alpha_filt = XS(FB_X_ALPHA_FILT); v.beta_filt = XS(FB_X_BETA_FILT);
    v.n_eng = XS(FB_X_ENG_OMEGA) / c172::w_rated;
    v.on_gnd = false;   // (the airborne pass: no strut has weight on its wheel)
    const CtlMemHalfT<gptr, HALF> M = {(gptr)(uintptr_t)A.cu + i, (gptr)(uintptr_t)A.cs + i, n, lu};
    FB_HALF_STAMP(HALF, HALF == CTL_HALF_LON ? 22 : 16);   // entry, arguments, the burst of loads
    gdc_update(M, v);
    FB_HALF_STAMP(HALF, HALF == CTL_HALF_LON ? 23 : 17);   // guidance
    const CtlTabT<gcptr> tab = ctl_tab((gcptr)(uintptr_t)A.gains, A.off, v.EAS, v.h_e);
    if constexpr (HALF == CTL_HALF_LON) {
        // (FB_X2_SPLIT_PITCH: the pitch-axis outer loops run on the partner wave; their elevator reference arrives inside the te2te LQR's run)
        ctl_lon<true, FB_X2_SPLIT_PITCH != 0>(tab, M, A.dT, v, (int)M.U(FB_CU_LON_MODE_REQ), [&]() { return hand.get(); });
        FB_HALF_STAMP(HALF, 24);   // (behind ctl_lon's fence 29: the LQR run and the stores)
        return {clampd(M.S(FB_CS_THROTTLE_CMD), 0, 1), clampd(M.S(FB_CS_ELEVATOR_CMD), -1, 1)};
    } else {
        if constexpr (FB_X2_SPLIT_PITCH != 0) {
            // first the longitudinal channel's pitch-axis outer loops (c2θ, θ -> q, q2e), whose result the partner wave's te2te LQR waits for
            hand.put(ctl_lon_pitch_half(tab, M, A.dT, v, (int)M.U(FB_CU_LON_MODE_REQ)));
            FB_HALF_STAMP(HALF, 18);   // pitch-axis outer loops, reference handed over
        }
        const int lat_req = (int)M.U(FB_CU_LAT_MODE_REQ);
        LatGains G;   // (not read: the gains are taken where they are used)
        G.P = {0, 0, 0, 0};
        ctl_lat<true>(tab, M, A.dT, v, lat_req, G);
        FB_HALF_STAMP(HALF, 19);   // lateral laws
        return {clampd(M.S(FB_CS_AILERON_CMD), -1, 1), clampd(M.S(FB_CS_RUDDER_CMD), -1, 1)};
    }
}

// ---- the wave-specialised airborne stepper (Cessna172Sv0) --------------------------------------------------------------------------
// k_step_air is bound by what ONE wave per SIMD can issue: its 151 KB of LDS panels and ~400 registers leave room for no second
// wave, every instruction costs the lone wave ~4.3 cycles and nothing hides an LDS round trip. k_step_duo serves the
// same 256 aircraft per workgroup with EIGHT waves: waves 0-3 ("P") evaluate geoid, atmosphere, propeller and engine, waves 4-7
// ("D") the airframe: aerodynamics, kinematics, mass, gravity, dynamics — rhs_duo() in c172_duo_device.hpp. Wave w and wave w + 4
// work on the same 64 aircraft, so a SIMD always has a second instruction stream to issue from; each role fits 256 registers.
// The two waves of a pair hand four things over per evaluation (velocity at the propeller D -> P; density and orthometric altitude,
// then the propeller's wrench P -> D; the control words of the next evaluation D -> P) and keep in step through two counters in LDS,
// without a barrier (DuoSync, below). LDS (full to the last byte): aero and piston tables 8 KB, x_n and the evaluation state 2 x 42 KB,
// D's seventeen stage sums 34 KB and three of P's four 6 KB, the payload's mass-property sums 20 KB, exchange 6 KB, flags 2 KB.
// The step's bookkeeping (stage machine, f_step!, hand-over to the ground-capable pass) is D's, exactly as in k_step_air;
// P follows through a per-lane flag word and a per-pair control word in LDS.
// NO WORKGROUP BARRIER inside the stepping loop: nothing crosses wave pairs. A pair whose 64 aircraft are all done (terminated before
// the launch, or handed over) leaves the kernel while the others go on, and pairs drift apart in their stage machines (a pair
// re-evaluates k1 after an f_step! that modified one of its lanes); neither needs anything of anybody. (Up to round 3 the pair met at
// __syncthreads(), which made both a matter of how s_barrier treats ended waves and mismatched call sites on this hardware.)
// tests/test_gpu_duo.py runs both situations (pairs running dry at different steps of one launch, lanes terminated before it) against
// the one-wave stepper and the oracle.
constexpr int DUO_B = 256;
constexpr int DUO_NP = 4, DUO_ND = 17;   // state rows per role
#ifndef FB_DUO_NPL
#define FB_DUO_NPL 0
#endif
// role P's own x_n rows (fuel, engine: nobody else writes them) in registers, refreshed from the panel at every stage 0: its emits — the
// last thing of its evaluation, behind role D's point X — then start without an LDS round trip (the Cessna172Sv0 instances: the Xv2 ones spill with it)
#ifndef FB_DUO_P_XN_REGS
#define FB_DUO_P_XN_REGS 1
#endif
constexpr int DUO_NPL = FB_DUO_NPL;   // how many of role P's four stage sums live in LDS (what is left of the 160 KB)
// how many of role D's seventeen live in LDS; the rest — from the end: the angular / linear velocity rows, whose emit closes the evaluation
// behind role P's point W, on the critical path of the pair — in registers (a ds_read + ds_write less per row and evaluation)
// (the WA Cessna172Sv0 instance: 13 in LDS, four in registers, 248 registers: 14.36 -> 14.26 ms per launch, profiles/r04_ab_acc_regs.txt; the
// ECEF / NED Cessna172Sv0 instances sit at 252-256 registers and would spill: all seventeen in LDS)
#ifndef FB_DUO_NDL
#define FB_DUO_NDL 13
#endif
// (the Cessna172Xv2 instances, since the update halves are inlined: 14 in LDS, three in registers, no spill: 9.95 -> 9.86 ms per launch)
#ifndef FB_DUO_NDL_X
#define FB_DUO_NDL_X 14
#endif
template <int KIN, bool X> constexpr int duo_ndl() { return (KIN == FB_KIN_WA && !X) ? FB_DUO_NDL : (X ? FB_DUO_NDL_X : DUO_ND); }
static_assert(FB_DUO_NDL >= 1 && FB_DUO_NDL <= DUO_ND, "");
// Per-aircraft launch constants cost a role twenty registers each if they ride through the evaluation. Role D reads the payload's
// ten mass-property sums from an LDS panel at the point of use and fetches its aerodynamic sums from global memory at the start of
// the evaluation (one batch of loads, consumed after the table locations).
struct InputsDuoP {
    double throttle, mixture;
    int ui;
    __device__ __forceinline__ double get_throttle() const { return throttle; }
    __device__ __forceinline__ double get_mixture() const { return mixture; }
};
struct InputsDuoD {
    static constexpr bool pld_precomputed = true, aero_precomputed = true;
    lds_cptr pld_l;         // &panel[lane], rows: M, Mr[3], J[6]
    const double* aero_g;   // &duo_pld[0 * n + i]
    int64_t n;
    int ui;
    __device__ __forceinline__ double get_pld_M() const { return pld_l[0]; }
    __device__ __forceinline__ double get_pld_Mr(int k) const { return pld_l[(1 + k) * DUO_B]; }
    __device__ __forceinline__ double get_pld_J(int k) const { return pld_l[(4 + k) * DUO_B]; }
    // Cessna172Xv2 (the two uses of panel and memory rows are exchanged there: the payload's ten sums — launch constants — come from the rows
    // of KArgs::duo_pld, like the Sv0 instance's aerodynamic constants, and the LDS panel carries the evaluation's aerodynamic sums from role P):
    static constexpr int SUMS_ROWS = 10;   // cd_in cd_df cl_df cm_in | de da dr | w_df4 w_df2 | the two table intervals in one word
    __device__ __forceinline__ void fetch_pld_raw(double (&v)[10]) const {
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = aero_g[(int64_t)k * n];
    }
    __device__ __forceinline__ void fetch_aero(AeroC& c) const {
        double v[DUO_NCONST];
        fetch_aero_raw(v);
        aero_from_raw(v, c);
    }
    // (in two steps for the Cessna172Xv2 instance, which fetches behind a wait and wants the loads in flight while it locates the knots)
    __device__ __forceinline__ void fetch_aero_raw(double (&v)[DUO_NCONST]) const {
#pragma unroll
        for (int k = 0; k < DUO_NCONST; k++) v[k] = aero_g[(int64_t)k * n];
    }
    __device__ __forceinline__ static void aero_from_raw(const double (&v)[DUO_NCONST], AeroC& c) {
        c.cd_in = v[0]; c.cd_df = v[1]; c.cy_in = v[2]; c.cl_in = v[3]; c.cl_df = v[4]; c.croll_in = v[5]; c.cm_in = v[6]; c.cn_in = v[7];
        c.l_df4 = {(int)v[8], v[9]}; c.l_df2 = {(int)v[10], v[11]};
    }
};
// role D's inputs in the Cessna172Xv2 instance: InputsDuoD with the payload's sums in registers (fetched from memory ahead of the kinematics
// block, consumed by the mass properties behind it)
struct InputsDuoDX : InputsDuoD {
    double pldv[10];
    __device__ __forceinline__ double get_pld_M() const { return pldv[0]; }
    __device__ __forceinline__ double get_pld_Mr(int k) const { return pldv[1 + k]; }
    __device__ __forceinline__ double get_pld_J(int k) const { return pldv[4 + k]; }
};
// How the two waves of a PAIR of the wave-specialised stepper (role P: wave w, role D: wave w + 4, the same 64 aircraft) keep in step.
// Nothing crosses pairs, so they do not meet at workgroup barriers — behind s_barrier a pair also waited for the three pairs on the other
// SIMDs, ~0.5 k cycles at each of four barriers per evaluation even with its partner already there (profiles/r03_duo_alone.txt) —
// but through two counters in LDS, one per wave: a wave PUBLISHES the points of its evaluation it has passed (one ds_write), and WAITS,
// where it needs something of its partner, until the partner's counter has reached the point that produces it. Three points per
// evaluation and role, numbered 3 i + k + 1 in iteration i of the evaluation loop:
//     role D publishes  T (k = 0)  top of the loop: control and flag words of this evaluation written, every row of the previous one emitted
//                       V (k = 1)  velocity at the propeller put (its own state rows read)
//                       X (k = 2)  propeller wrench and fuel row read
//     role P publishes  R (k = 0)  state rows read (so: its previous evaluation is complete, fuel row included)
//                       A (k = 1)  density and orthometric altitude put
//                       W (k = 2)  propeller wrench put
//     role D waits for  R before it emits the kinematics rows (role P reads q_ew, h_e) and reads the fuel row;  A before the part of the
//                       aerodynamics that needs the atmosphere;  W before the last part of the dynamics
//     role P waits for  T before it reads the control words and the state;  V before the propeller;  X before it emits the engine-speed and
//                       fuel rows (in place, at the end of its evaluation)
// Each wait sits where the value is needed and no earlier, so the waves slide against each other: role D — the critical path
// (tools/duo_alone.sh: 12.7 ms per launch alone on its SIMD, role P 10.7) — almost never finds itself waiting (tools/duo_waitprof.py).
// The LDS serves one wave's instructions in order: what a wave wrote ahead of its counter is there when the partner sees the count, and
// what it read ahead of it has been read. The waits are bounded (a partner that never arrives would hang the GPU): the pair then goes on
// and its aircraft are flagged FB_ST_NAN when the kernel ends.
// Diagnostic builds (-DFB_STAMP -DFB_DUO_WAITPROF, tools/duo_waitprof.py): wave 0 (role P) and wave 4 (role D) of workgroup 0 add the
// cycles they spend in wait k to g_stamp_acc[k] / [8 + k].
#ifdef FB_DUO_SYNC_DEBUG
__device__ unsigned g_duo_sync_dbg[40];
#endif
struct DuoSync {
    volatile __attribute__((address_space(3))) int* mine;
    volatile __attribute__((address_space(3))) int* other;
    int base;     // DUO_NPT x the iteration of the evaluation loop. WAVE-UNIFORM: advanced by the loops, in uniform code (most points sit inside
                  // the divergent `if (run)` of the evaluation, where a counter incremented in place would advance for the running lanes only)
    int failed;
};
// Release / acquire. The counter store is a workgroup-scope RELEASE of this wave's LDS traffic: `s_waitcnt lgkmcnt(0)` ahead of it — every
// ds_write above has been performed, every ds_read above has returned — which is what the LLVM AMDGPU memory model prescribes for a
// workgroup-scope release fence over the local address space on gfx9 (the builtin fence, also with the "local" address-space argument,
// adds vmcnt(0) on this compiler: it would stall role D on its aerodynamic constants' global loads at every point, so the wait is
// written out). The ACQUIRE side needs no instruction of its own: the poll's ds_read has returned (lgkmcnt(0) ahead of the
// v_readfirstlane that consumes it) before any LDS access below the loop is issued, and the LDS performs one wave's accesses in issue
// order. -DFB_DUO_RELEASE_WAIT=0 builds the round-3 form (compiler barriers only; it rests on that in-order service for the stores
// too) for A/B timing: profiles/r04_ab_fence.txt.
#ifndef FB_DUO_RELEASE_WAIT
#define FB_DUO_RELEASE_WAIT 1
#endif
// GLOBAL: the release also covers this wave's global-memory traffic (vmcnt(0): the Cessna172Xv2 instance hands values over through global
// memory too — the evaluation's aerodynamic sums, the tapped outputs of a control update; the two waves of a pair share their CU's vector
// L1, so a workgroup-scope release / acquire needs no cache maintenance).
template <bool GLOBAL = false>
FBD void duo_publish(DuoSync& sy, int k) {
    if constexpr (GLOBAL) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else {
#if FB_DUO_RELEASE_WAIT
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (and, for the compiler: every LDS access above stays above ...)
#else
        asm volatile("" ::: "memory");   // (compiler: every LDS access above stays above ...)
#endif
    }
    *sy.mine = sy.base + k + 1;
    asm volatile("" ::: "memory");   // (... and every one below stays below)
}
FBD void duo_wait(DuoSync& sy, int k) {
#if defined(FB_STAMP) && defined(FB_DUO_WAITPROF)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    const int c = sy.base + k + 1;
    asm volatile("" ::: "memory");
    int spins = 0;
#pragma unroll 1
    while (__builtin_amdgcn_readfirstlane(*sy.other) < c) {
        __builtin_amdgcn_s_sleep(1);
#ifdef FB_DUO_SYNC_DEBUG
        if (++spins > (1 << 10)) {
            sy.failed = 1;
            if ((threadIdx.x & 63) == 0 && atomicAdd(&g_duo_sync_dbg[0], 1u) < 7) {
                const unsigned q = atomicAdd(&g_duo_sync_dbg[1], 4u);
                g_duo_sync_dbg[2 + q] = blockIdx.x; g_duo_sync_dbg[3 + q] = threadIdx.x; g_duo_sync_dbg[4 + q] = c; g_duo_sync_dbg[5 + q] = *sy.other;
            }
            break;
        }
#else
        if (++spins > (1 << 20)) { sy.failed = 1; break; }
#endif
    }
    asm volatile("" ::: "memory");
#if defined(FB_STAMP) && defined(FB_DUO_WAITPROF)
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == DUO_B)) {
        const int q = k + (threadIdx.x ? 8 : 0);
        g_stamp_acc[q] += t1 - t0; g_stamp_cnt[q] += 1;
    }
#endif
}

template <int ROLE, bool X = false, int NDL = DUO_ND>   // NDL: how many of role D's stage sums live in LDS (duo_ndl())
struct DuoEmit {
    static constexpr int role = ROLE;
    static constexpr bool x2 = X;   // the Cessna172Xv2 instance: role D fetches the evaluation's aerodynamic sums behind role P's point R, every publication
                                    // releases global memory too, and a tapped evaluation (`tap`) stores the control laws' inputs to rows of KArgs::duo_tap
    typedef void batched_tag;
    using SV = StateLds<DUO_B, false>;
    lds_cptr xs_l;     // x_n panel
    lds_ptr xwr_l;     // the panel this stage writes
    lds_ptr acc_l;     // this role's stage sums in LDS: D's seventeen [17][DUO_B], the first DUO_NPL of P's four
    double* acc_r;     // the rest of the role's stage sums (registers)
    lds_ptr xch_l;     // exchange rows 6.. [XD_ROWS - 6][DUO_B]
    lds_ptr xov_l;     // exchange rows 0-5: the angular / linear velocity rows of the evaluation panel (see c172_duo_device.hpp)
    double eb, ee, em;
    bool last;
    int t;
    DuoSync* sync;     // this wave's side of the pair's synchronisation counters
    bool tap;          // wave-uniform (Cessna172Xv2): this is the step's last f_ode! and a control update follows it
    int64_t ai;        // the lane's aircraft
    const double* xn_r = nullptr;   // role P (FB_DUO_P_XN_REGS): x_n of its own rows
    GeoidCache* gcache = nullptr;   // role P (FB_DUO_GEOID_CACHE): the EGM96 cell of the lane's aircraft, kept from evaluation to evaluation
    // rows k0 .. k0 + N - 1 of KArgs::duo_tap (base and stride re-read from the kernel's arguments: once per control period, see kernarg())
    template <int N>
    __device__ __forceinline__ void tap_rows(int k0, const double (&v)[N]) const {
        const kargs_cptr ka = kernarg();
        double* const g = ka->duo_tap + ai;
        const int64_t n = ka->n;
#pragma unroll
        for (int e = 0; e < N; e++) g[(int64_t)(k0 + e) * n] = v[e];
    }
    // panel rows: 0-1 filters (D), 2 fuel, 3-5 engine (P), 6-14 q_wb q_ew h_e (D), 15-20 angular and linear velocity (D)
    __device__ __forceinline__ static constexpr bool owned(int j) {
        if (SV::skip(j)) return false;
        const int r = SV::row(j);
        return ROLE == 1 ? (r >= 2 && r <= 5) : (r < 2 || r >= 6);
    }
    __device__ __forceinline__ static constexpr int slot(int r) { return ROLE == 1 ? r - 2 : (r < 2 ? r : r - 4); }
    __device__ __forceinline__ double aget(int r) const {
        if (ROLE == 1 && slot(r) >= DUO_NPL) return acc_r[slot(r) - DUO_NPL];
        if (ROLE == 2 && slot(r) >= NDL) return acc_r[slot(r) - NDL];
        return acc_l[slot(r) * DUO_B + t];
    }
    __device__ __forceinline__ void aset(int r, double v) const {
        if (ROLE == 1 && slot(r) >= DUO_NPL) acc_r[slot(r) - DUO_NPL] = v;
        else if (ROLE == 2 && slot(r) >= NDL) acc_r[slot(r) - NDL] = v;
        else acc_l[slot(r) * DUO_B + t] = v;
    }
    __device__ __forceinline__ void operator()(int j, double kj) const {
        static_assert(ROLE == 1 || ROLE == 2, "");
        const int r = SV::row(j), idx = r * DUO_B + t;
        const double xs = (ROLE == 1 && FB_DUO_P_XN_REGS && !X) ? xn_r[slot(r)] : xs_l[idx];
        const double A = __builtin_fma(eb, kj, aget(r));
        aset(r, A * em);
        xwr_l[idx] = __builtin_fma(ee, last ? A : kj, xs);
    }
    template <int NE>
    __device__ __forceinline__ void batch(int j0, const double (&k)[NE]) const {
        double xs[NE], A[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) { const int r = SV::row(j0 + e); xs[e] = (ROLE == 1 && FB_DUO_P_XN_REGS && !X) ? xn_r[slot(r)] : xs_l[r * DUO_B + t]; A[e] = aget(r); }
#pragma unroll
        for (int e = 0; e < NE; e++) A[e] = __builtin_fma(eb, k[e], A[e]);
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int r = SV::row(j0 + e);
            aset(r, A[e] * em);
            xwr_l[r * DUO_B + t] = __builtin_fma(ee, last ? A[e] : k[e], xs[e]);
        }
    }
    __device__ __forceinline__ void xput(int k, double v) const { if (k < 6) xov_l[k * DUO_B + t] = v; else xch_l[(k - 6) * DUO_B + t] = v; }
    __device__ __forceinline__ double xget(int k) const { return k < 6 ? xov_l[k * DUO_B + t] : xch_l[(k - 6) * DUO_B + t]; }
    // (Cessna172Xv2: a tapped evaluation hands values over through global memory too — role P's tap rows are released at its point W)
    __device__ __forceinline__ void xpub(int k) const { if (X && tap && ROLE == 1) duo_publish<true>(*sync, k); else duo_publish<false>(*sync, k); }
    __device__ __forceinline__ void xwait(int k) const { duo_wait(*sync, k); }
};
// every state row of an airborne aircraft belongs to exactly one role, and a role's rows fill its stage-sum slots exactly once
constexpr bool duo_rows_ok() {
    int np = 0, nd = 0;
    bool seen_p[DUO_NP] = {}, seen_d[DUO_ND] = {};
    for (int j = 0; j < FB_NX; j++) {
        if (StateLds<DUO_B, false>::skip(j)) { if (DuoEmit<1>::owned(j) || DuoEmit<2>::owned(j)) return false; continue; }
        const int r = StateLds<DUO_B, false>::row(j);
        if (DuoEmit<1>::owned(j) == DuoEmit<2>::owned(j)) return false;
        if (DuoEmit<1>::owned(j)) { const int k = DuoEmit<1>::slot(r); if (k < 0 || k >= DUO_NP || seen_p[k]) return false; seen_p[k] = true; np++; }
        else { const int k = DuoEmit<2>::slot(r); if (k < 0 || k >= DUO_ND || seen_d[k]) return false; seen_d[k] = true; nd++; }
    }
    return np == DUO_NP && nd == DUO_ND;
}
static_assert(duo_rows_ok(), "k_step_duo: row ownership / stage-sum slots");
static_assert(DuoEmit<1>::owned(FB_X_FUEL) && DuoEmit<1>::owned(FB_X_ENG_OMEGA) && DuoEmit<2>::owned(FB_X_Q_WB) && DuoEmit<2>::owned(FB_X_V_EB_B + 2),
              "rhs_duo emits the engine and fuel rows in role P, everything else in role D");
enum { DUO_F_RUN = 1, DUO_F_ZERO_ACC = 2, DUO_F_ENG_SHIFT = 2, DUO_F_CTL = 16 };   // per-lane flag word (CTL: this lane takes part in the control update that follows)
enum { DUO_C_EXIT = 4, DUO_C_TAP = 8, DUO_C_CMD = 16 };            // per-pair control word: stage | EXIT | TAP (this evaluation is tapped) | CMD (new lateral commands put)
// Cessna172Xv2 (X = true; configs[3]): the same pair of waves, and
//   * the five actuators the airborne evaluation reads (throttle, aileron, elevator, rudder, flaps: closed-form RK4, see k_step_air) ride in
//     role P's registers — it has ~85 to spare, role D none. Role P forms the throttle position for its engine and, at the head of every
//     evaluation, the deflection-only aerodynamic sums of that stage's surface positions (InputsAgg::sum_aero, ~100 instructions that role
//     D's critical path does not pay), and hands them to role D through the rows of KArgs::duo_pld — the rows role D fetches its launch
//     constants from in the Sv0 instance; role D fetches them behind role P's point R (which therefore releases global memory too);
//   * f_periodic!(avionics, vehicle) runs inside the launch like in k_step_air, in two halves side by side: the step's last f_ode! is TAPPED
//     by both roles (each stores the outputs its part of the evaluation forms to the rows of KArgs::duo_tap), role D keeps the book as
//     always (f_step!, then flags with DUO_F_CTL for the lanes that take part) and publishes a fourth point U; then role P's wave runs the
//     longitudinal laws and role D's the lateral ones (x2_periodic_half). Role P keeps its two new commands, role D puts its two into the
//     exchange rows that carry density and altitude during an evaluation (DUO_C_CMD); role P publishes F when its half is done, which role
//     D waits for before the next evaluation's T (so the record's rows are at rest whenever an evaluation runs, and when a lane's record is
//     put back from ctl_bak at the end). Callback order as in the reference: cb_step, then cb_periodic (FC/sim.jl:204-218).
//   * no derivative is carried across launches (k_step_air<X> saves one evaluation in 201 that way): a.k1_valid is cleared for the lanes
//     whose steps are committed here; a lane handed over keeps the one the ground-capable pass left it (see the epilogue).
constexpr int DUO_PT_U = DUO_NPT, DUO_PT_F = DUO_NPT;   // (behind the evaluation's points) role D: flags of the update written; role P: its half of the update done
constexpr int DUO_PT_Z = DUO_NPT + 1;                   // role D (inside its half of an update): the pitch-axis outer loops' elevator reference put (x2_periodic_half)
// Diagnostic builds (-DFB_STAMP -DFB_DUO_PHASES, tools/duo_phases.py): the shader clock when role D's first wave of workgroup 0 passes the
// phases of a launch (g_stamp_acc[8 + k]: 0 entry, 1 tables staged, 2 state loaded and launch constants formed, 3 last evaluation done, 4 exit)
#if defined(FB_STAMP) && defined(FB_DUO_PHASES)
#define DUO_PHASE(k) do { if (blockIdx.x == 0 && threadIdx.x == 256) { __builtin_amdgcn_s_waitcnt(0); g_stamp_acc[8 + (k)] = __builtin_amdgcn_s_memtime(); g_stamp_cnt[8 + (k)] += 1; } } while (0)
#else
#define DUO_PHASE(k) do { } while (0)
#endif
// PERENV: every aircraft in its own environment (KArgs::env_rows, fb_set_env). Role P — ISA atmosphere, engine — keeps its four values (sea-level
// T and p and the two derived from them) in registers, of which it has ~130 to spare; role D — wind-relative velocity, height over the terrain —
// has none, and reads its four (wind N / E / D, terrain elevation) from an LDS panel of its own at the point of use (4 rows x 256 lanes: 8 KB
// of the 10 KB the WA instances leave free; the ECEF / NED instances have no room and are stepped by k_step_air<.., PERENV>).
template <int KIN, bool X = false, bool PERENV = false>
__global__ __launch_bounds__(2 * DUO_B) void k_step_duo(KArgs a, int nsteps) {
    constexpr int B = DUO_B, NR = FB_NX - 6, NP = DUO_NP, ND = DUO_ND;
    constexpr int NPT = X ? DUO_NPT + 2 : DUO_NPT;   // points per iteration of the evaluation loop
    constexpr int NAL = FB_ACT_BRAKE_LEFT;           // actuators the airborne evaluation reads
    using SV = StateLds<B, false>;
    __shared__ double lds[AT_SIZE + PT_SIZE];   // aero | piston tables (the propeller table stays in global memory, see rhs_duo())
    __shared__ double rk[LDS_RK_DOUBLES];
    __shared__ double xs_l[NR * B];    // x_n
    __shared__ double xc_l[NR * B];    // the state being evaluated, updated in place by the emits
    constexpr int DUO_NDL = duo_ndl<KIN, X>();
    __shared__ double accd_l[DUO_NDL * B];  // role D's stage sums, as far as they live in LDS (duo_ndl())
    __shared__ double accp_l[(DUO_NPL > 0 ? DUO_NPL : 1) * B];   // role P's, as far as the LDS goes
    __shared__ double pld_l[10 * B];   // role D: the payload's mass-property sums
    __shared__ double xch_l[(XD_ROWS - 6) * B];
    __shared__ int flags_l[B];
    __shared__ int dst_l[B];           // role D's per-lane bookkeeping word
    // PERENV: role D's wind N / E / D and terrain elevation — in the WA instances from an LDS panel [4][B] of its own (written and read by role D
    // alone); the ECEF / NED Cessna172Sv0 instances, whose seventeen stage sums fill the LDS, fetch the four rows from memory at every evaluation (L2-resident,
    // requested at the head of the evaluation like the Sv0 instance's aerodynamic constants)
    constexpr bool ENV_LDS = PERENV && (KIN == FB_KIN_WA || X);   // (the Cessna172Xv2 instances keep fourteen stage sums in LDS: 8 KB are free in every mechanisation)
    __shared__ double envd_l[ENV_LDS ? 4 * B : 1];
    static_assert(LDS_RK_DOUBLES >= LDS_ATAN + ATAN_N + 2, "room for the control words behind the atan table");
    int* ctrl_l = (int*)&rk[LDS_ATAN + ATAN_N];   // one control word per wave pair (the LDS is full to the last 16 bytes)
    int* sync_l = ctrl_l + 4;                      // two synchronisation counters per wave pair: [pair] role P's, [4 + pair] role D's
    static_assert(LDS_RK_DOUBLES >= LDS_ATAN + ATAN_N + 6, "room for the control words and the synchronisation counters behind the atan table");
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) + 1;   // 1: P (waves 0-3), 2: D (waves 4-7)
    // Issue priority. The SIMD's arbiter serves the OLDER wave first, i.e. role P (waves 0-3), while role D is the critical path of
    // every barrier-to-barrier segment (profiles/r02_duo_barrier_wait.txt: P waits 6.0 k cycles per evaluation inside the barriers, D 0.3 k):
    // P raced ahead on the shared fp64 pipe, then sat in the barrier while D ran alone with nobody to fill its stalls. With D ahead in
    // priority P fills D's gaps instead: 16.64 -> 15.95 ms per launch (profiles/r03_ab_prio.txt; s_setprio 1, 2, 3 and D in waves 0-3
    // measure alike).
#ifndef FB_DUO_PRIO_D
#define FB_DUO_PRIO_D 2
#endif
#ifndef FB_X2_LON_PRIO
#define FB_X2_LON_PRIO 3   // Cessna172Xv2: role P's wave during its half of a control update (see there)
#endif
#ifndef FB_X2_HEAD_PRIO
#define FB_X2_HEAD_PRIO 3  // ... and from the top of its loop to its point R
#endif
#ifndef FB_X2_UPD_PRIO
#define FB_X2_UPD_PRIO 0   // ... and role D's wave during ITS half (the shorter one): 9.86 -> 9.82 ms per launch
#endif
#ifndef FB_X2_PITCH_PRIO
#define FB_X2_PITCH_PRIO 3 // ... and role D's wave from the start of its half to its point Z (the pitch-axis outer loops, which role P's LQR waits for)
#endif
#ifndef FB_X2_SPEC_PRIO
#define FB_X2_SPEC_PRIO 3  // ... and while it forms the next stage's aerodynamic sums at the end of an iteration
#endif
    if (role == 2) __builtin_amdgcn_s_setprio(FB_DUO_PRIO_D);
#ifdef FB_DUO_PRIO_P
    if (role == 1) __builtin_amdgcn_s_setprio(FB_DUO_PRIO_P);
#endif
    const int t = threadIdx.x & (B - 1);
    const int pair = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t i = (int64_t)blockIdx.x * B + t;
    DUO_PHASE(0);
    for (int k = threadIdx.x; k < AT_SIZE + PT_SIZE; k += blockDim.x) lds[k] = a.tables[k];
    __syncthreads();
    for (int k = threadIdx.x; k < LDS_RK_KNOTS; k += blockDim.x) rk[k] = 1.0 / (lds[k + 1] - lds[k]);
    for (int k = threadIdx.x; k < ATAN_N; k += blockDim.x) rk[LDS_ATAN + k] = atan(k * (1.0 / 32));
    if (threadIdx.x < 8) sync_l[threadIdx.x] = 0;
    __syncthreads();
    DUO_PHASE(1);
    DuoSync sy = {(volatile __attribute__((address_space(3))) int*)(sync_l + (role == 1 ? pair : 4 + pair)),
                  (volatile __attribute__((address_space(3))) int*)(sync_l + (role == 1 ? 4 + pair : pair)), 0, 0};
    const bool valid = i < a.n && a.status[i] == 0;
    const double dt = a.dt, hdt = a.dt / 2, dt6 = a.dt / 6;
    Env env_p = a.env;   // what role P's evaluation reads of the environment: T_sl, p_sl, ln_p_sl, k_rt (rhs_duo<KIN, 1>)
    if constexpr (PERENV) {
        const Env e = (i < a.n) ? env_of(a, i) : a.env;
        if (role == 1) env_p = e;
        else if constexpr (ENV_LDS) { envd_l[0 * B + t] = e.wind_n; envd_l[1 * B + t] = e.wind_e; envd_l[2 * B + t] = e.wind_d; envd_l[3 * B + t] = e.h_trn; }
    }
    // role D's per-lane bookkeeping word, kept in LDS between evaluations (role P reads it once, when the launch is over)
    enum { D_ALIVE = 1, D_HANDOFF = 4, D_STALL = 8, D_ACTIVE = 16, D_ENG_SHIFT = 5 };   // (no lane ends its simulation here: a status bit is a hand-over)
    // what an evaluation at stage `stg` needs (wave-uniform)
    struct StageK { bool last; double eb, ee, em; lds_cptr xrd_l; lds_ptr xwr_l; };
    auto stage_k = [&](int stg) {
        StageK k;
        k.last = stg == 3;
        if constexpr (X) {
            // (formed here from an opaque copy of dt: as loop invariants in VGPRs, dt / 2 and dt / 6 are what the Cessna172Xv2 instance's
            // allocation spills — and reloads in every evaluation)
            double dtl = a.dt;
            asm volatile("" : "+s"(dtl));
            k.eb = (stg == 1 || stg == 2) ? 2.0 : 1.0; k.ee = k.last ? dtl / 6 : (stg == 2 ? dtl : dtl / 2); k.em = k.last ? 0.0 : 1.0;
        } else {
            k.eb = (stg == 1 || stg == 2) ? 2.0 : 1.0; k.ee = k.last ? dt6 : (stg == 2 ? dt : hdt); k.em = k.last ? 0.0 : 1.0;
        }
        // panel roles: stage 0 evaluates x_n straight from xs_l, stages 1-3 the state the previous stage left in xc_l; stages 0-2
        // write the next evaluation state to xc_l, stage 3 the new x_n to xs_l
        k.xrd_l = stg == 0 ? (lds_cptr)xs_l : (lds_cptr)xc_l;
        k.xwr_l = k.last ? (lds_ptr)xs_l : (lds_ptr)xc_l;
        return k;
    };
    // one half of a control update (the arguments re-read from the kernel's argument block where the call stands: kernarg())
    // The elevator reference of the pitch-axis outer loops, from the lateral half's wave (role D's) to the longitudinal half's (role P's): through
    // the lane's slot of the exchange row that carries h_rot during an evaluation (role P writes it ahead of its point W, role D reads it behind W:
    // idle between evaluations, like the rho / h_o rows that carry the lateral commands) and role D's point Z (an LDS-only release: what crosses is
    // this one row). NOT the evaluation panel's velocity rows: the tapped evaluation is stage 0 of the next step, its emits have left the stage-1
    // state there.
    struct CtlHand {
        lds_ptr slot; DuoSync* sy;
        __device__ __forceinline__ void put(double v) const { *slot = v; duo_publish<false>(*sy, DUO_PT_Z); __builtin_amdgcn_s_setprio(FB_X2_UPD_PRIO); }
        __device__ __forceinline__ double get() const { duo_wait(*sy, DUO_PT_Z); return *slot; }
    };
    [[maybe_unused]] auto ctl_half = [&](auto half, int64_t lane) {
        const kargs_cptr ka = kernarg();
        auto pk = [&](int k) { return (uint32_t)ka->ctl_off.off[k] | ((uint32_t)ka->ctl_off.off[k + 1] << 16); };
        const CtlHand hand = {(lds_ptr)xch_l + (XD_HROT - 6) * B + t, &sy};
        return x2_periodic_half<decltype(half)::value>(ka->cu, ka->cs, ka->gains, ka->duo_tap, ka->n, ka->ctl_dT, pk(0), pk(2), pk(4), pk(6), pk(8),
                                                       (uint32_t)ka->ctl_off.total | ((uint32_t)ka->ctl_off.same_grid << 16), lane,
                                                       xs_l[SV::row(h_e_row<KIN>()) * B + t], (lds_cptr)xs_l + t, hand);
    };
    if (role == 1) {
        // ================= role P =================
        InputsDuoP in;
        in.throttle = 0; in.mixture = 0; in.ui = 0;
        [[maybe_unused]] double xa[X ? NAL : 1], ca[X ? NAL : 1];   // Cessna172Xv2: actuator positions x_n and the commands in force
        if constexpr (X) {
#pragma unroll
            for (int k = 0; k < NAL; k++) { xa[k] = 0; ca[k] = 0; }
            if (valid) {
#pragma unroll
                for (int k = 0; k < NAL; k++) { xa[k] = a.x[(int64_t)(X2_ACT + k) * a.n + i]; ca[k] = x2_command(a, i, k); }
                in.mixture = clampd(a.u[(int64_t)FB_U_MIXTURE * a.n + i], 0, 1);
                in.ui = a.ui[i];
                if (FB_X2_BAK && a.ctl_ratio > 0 && nsteps > 1) {   // its half of the launch-start copy of the control-law record (role D copies cu); not for a launch of one step: see k_step_air
                    static_assert(FB_NCS % 11 == 0 && FB_NCU % 14 == 0 && FB_NCS % FB_X2_BAK_G_CS == 0 && FB_NCU % FB_X2_BAK_G_CU == 0, "batch sizes of the record copies");
                    copy_rows_batched<FB_NCS, FB_X2_BAK_G_CS>(a.ctl_bak, a.cs, a.n, i);
                }
            }
            // once per launch: the backup rows have left this wave before its first publication, whichever kind that is — role D restores a
            // handed-over lane from ctl_bak (D_HANDOFF), and an ordinary publication (duo_publish<false>) waits on lgkmcnt only
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        } else if (valid) {
            Inputs in0;
            load_inputs(a, i, in0);
            in.throttle = in0.throttle; in.mixture = in0.mixture; in.ui = in0.ui;
        }
        double acc_r[NP - DUO_NPL];
        double xn_r[NP] = {0, 0, 0, 0};
        GeoidCache gcache = {-1, -1, 0.0f, 0.0f, 0.0f, 0.0f};   // (no cell yet: the first evaluation gathers)
#pragma unroll
        for (int k = 0; k < NP - DUO_NPL; k++) acc_r[k] = 0.0;
#pragma unroll
        for (int k = 0; k < DUO_NPL; k++) accp_l[k * B + t] = 0.0;
        [[maybe_unused]] const double z0 = dt / ACT_TAU;
        // Cessna172Xv2: the deflection-only aerodynamic sums (InputsAgg::sum_aero) of the surface actuators' positions at stage `stg_for`, for
        // role D, through the LDS panel that carries the payload's sums in the Sv0 instance (role D reads them behind this role's point R). Formed
        // for every lane, running or not, from the lane's own actuator registers — so what is in memory is valid for the whole wave — and
        // formed AHEAD: at the end of an iteration for the stage that normally follows, while role D finishes its evaluation and keeps the
        // book; at the head of an evaluation only when that guess was wrong (a re-evaluation of k1, or new commands from a control update).
        [[maybe_unused]] int sums_for = -1;   // wave-uniform: the stage whose sums are in the panel (-1: none)
        [[maybe_unused]] double df_last = __builtin_nan(""), cm_df = 0;   // per lane: the flap deflection the flap-dependent rows were formed for; its C_m term
        [[maybe_unused]] auto form_sums = [&](int stg_for, int lds_off) {
            double z = z0;
            asm volatile("" : "+v"(z));
            const double ms = act_stage_mc(stg_for, z);
            double xa_s[FB_NACT];
#pragma unroll
            for (int k = 0; k < NAL; k++) xa_s[k] = act_stage_pos(xa[k], ca[k], ms, stg_for == 0);
            xa_s[FB_ACT_BRAKE_LEFT] = 0; xa_s[FB_ACT_BRAKE_RIGHT] = 0;   // (never read in the air)
            const InputsX ix = {xa_s, nullptr, a.n, in.ui};
            // InputsAgg::sum_aero's expressions, term by term; the flap-dependent ones (two table locations, three lookups, five of the ten
            // rows) only when a lane's flap position has changed since they were formed — flaps move in seconds, a stage follows a stage in
            // microseconds
            const double de = ix.get_de(), da = ix.get_da(), dr = ix.get_dr(), df = ix.get_df();
            lds_cptr A = (lds_cptr)lds + LDS_AERO + lds_off, RA = (lds_cptr)rk + LDS_AERO + lds_off;
            auto S_ = [&](int k) -> double { return A[AT_SCALARS + k]; };
            lds_ptr sp = (lds_ptr)pld_l + t + lds_off;
            if (__builtin_amdgcn_ballot_w64(!(df == df_last)) != 0) {
                const loc l_df4 = grid_locate<4>(A + AT_DF4_K, RA + AT_DF4_K, df, true, true);
                const loc l_df2 = grid_locate<2>(A + AT_DF2_K, RA + AT_DF2_K, df, true, true);
                cm_df = lerp1(A + AT_CM_DF_V, l_df4);
                const uint64_t iw = (uint64_t)(uint32_t)l_df4.i | ((uint64_t)(uint32_t)l_df2.i << 32);
                sp[1 * B] = lerp1(A + AT_CD_DF_V, l_df4); sp[2 * B] = lerp1(A + AT_CL_DF_V, l_df4);
                sp[7 * B] = l_df4.w; sp[8 * B] = l_df2.w; sp[9 * B] = __builtin_bit_cast(double, iw);
                df_last = df;
            }
            const loc l_de = grid_locate<3>(A + AT_UNIT3_K, RA + AT_UNIT3_K, de, true, true);
            // ten rows of the panel: the sums that need a table (0 cd_in, 1 cd_df, 2 cl_df, 3 cm_in), the three deflections the linear ones are formed
            // from by role D (4-6: cy_in, cl_in, croll_in, cn_in are seven multiply-adds with scalar-loaded derivatives there), the flap-axis
            // weights (7, 8) and, in one word, the two intervals (9)
            sp[0] = S_(AS_CD_ZERO) + lerp1(A + AT_CD_DE_V, l_de);
            sp[3 * B] = S_(AS_CM_ZERO) + S_(AS_CM_DE) * de + cm_df;
            sp[4 * B] = de; sp[5 * B] = da; sp[6 * B] = dr;
            static_assert(InputsDuoD::SUMS_ROWS == 10, "panel rows");
            sums_for = stg_for;
        };
#pragma unroll 1
        while (true) {
            DUO_MARK(1, 15);  // (arrival at the top of the loop, counted from the previous evaluation's start)
            duo_wait(sy, DUO_PT_T);   // role D's control and flag words of this evaluation are written, every row of the previous one emitted
            DUO_MARK(1, 0);
            if constexpr (X) __builtin_amdgcn_s_setprio(FB_X2_HEAD_PRIO);   // (until its point R: role D reads the evaluation's sums behind it, early in its own evaluation)
            const int c = __builtin_amdgcn_readfirstlane(ctrl_l[pair]);
            const int f = flags_l[t];
            if constexpr (X) {
                if (c & DUO_C_CMD) {   // the lateral half of the update that has just run has put its two commands (role D's wave, rows rho / h_o)
                    if (f & DUO_F_CTL) { ca[FB_ACT_AILERON] = xch_l[(XD_RHO - 6) * B + t]; ca[FB_ACT_RUDDER] = xch_l[(XD_HO - 6) * B + t]; }
                }
            }
            if (c & DUO_C_EXIT) break;
            const bool run = f & DUO_F_RUN;
            const int eng = (f >> DUO_F_ENG_SHIFT) & 3;
            if (f & DUO_F_ZERO_ACC) {
#pragma unroll
                for (int k = 0; k < NP - DUO_NPL; k++) acc_r[k] = 0.0;
#pragma unroll
                for (int k = 0; k < DUO_NPL; k++) accp_l[k * B + t] = 0.0;
            }
            const int stg = c & 3;
            if (FB_DUO_P_XN_REGS && !X && stg == 0) {
#pragma unroll
                for (int k = 0; k < NP; k++) xn_r[k] = xs_l[(2 + k) * B + t];
            }
            const StageK sk = stage_k(stg);
            int lds_off = 0;
            asm volatile("" : "+s"(lds_off));   // (see k_step_air: keeps the loop-invariant table / input loads from being hoisted into registers)
            const Tables T = {(lds_cptr)lds + lds_off, a.egm96, (lds_cptr)rk + lds_off, (gk_cptr)a.tables + lds_off};
            [[maybe_unused]] const bool tap = X && (c & DUO_C_TAP);
            if constexpr (X) { if (sums_for != stg || (c & DUO_C_CMD)) form_sums(stg, lds_off); }
            if (__builtin_amdgcn_ballot_w64(run) != 0) {   // (the same in both waves of a pair)
                if (run) {
                    StepAux aux;
                    InputsDuoP inl = in;
                    if constexpr (X) {
                        // the throttle actuator's stage position (closed-form RK4, see k_step_air) for the engine; the deflection-only aerodynamic
                        // sums of this stage's surface positions are in memory already when the speculation at the end of the previous
                        // iteration was right (the usual case: the next stage, the same commands)
                        double z = z0;
                        asm volatile("" : "+v"(z));   // (opaque: the stage multipliers are formed here, not hoisted out of the loop into registers that then spill)
                        const double ms = act_stage_mc(stg, z);
                        inl.throttle = clampd(act_stage_pos(xa[FB_ACT_THROTTLE], ca[FB_ACT_THROTTLE], ms, stg == 0), 0.0, 1.0);   // InputsX::get_throttle
                    }
                    DUO_MARK(1, 12);   // (Cessna172Xv2: stage positions and aerodynamic sums formed and stored)
                    asm volatile("" : "+v"(inl.throttle), "+v"(inl.mixture));
                    const DuoEmit<1, X> emit = {(lds_cptr)xs_l, sk.xwr_l, (lds_ptr)accp_l, acc_r, (lds_ptr)xch_l, (lds_ptr)xc_l + 15 * B, sk.eb, sk.ee, sk.em, sk.last, t, &sy,
                                                tap, i, xn_r, &gcache};
                    if constexpr (X) {
                        if (tap) {   // what the control laws read of the actuators: the Ranged positions of the state x_{n+1} (InputsX::pos), and the commands this f_ode! saw
                            static_assert(DUO_TAP_CMD == DUO_TAP_POS + 4, "positions, then commands");
                            const double pc[8] = {clampd(xa[0], 0.0, 1.0), clampd(xa[1], -1.0, 1.0), clampd(xa[2], -1.0, 1.0), clampd(xa[3], -1.0, 1.0), ca[0], ca[1], ca[2], ca[3]};
                            emit.template tap_rows<8>(DUO_TAP_POS, pc);
                        }
                    }
                    const SV xv = {sk.xrd_l + t + lds_off};
#if defined(FB_DUO_ONLY) && FB_DUO_ONLY == 2
                    // timing diagnostic (tools/duo_alone.sh): role D alone on its SIMD — role P puts plausible constants and publishes its points
                    emit.xpub(DUO_PT_R);
                    emit.xput(XD_RHO, 1.1); emit.xput(XD_HO, 1000.0);
                    emit.xpub(DUO_PT_A);
                    emit.xwait(DUO_PT_V);
                    emit.xput(XD_FP, 900.0); emit.xput(XD_FP + 1, 0.0); emit.xput(XD_FP + 2, 0.0);
                    emit.xput(XD_TAUP, -150.0); emit.xput(XD_TAUP + 1, 0.0); emit.xput(XD_TAUP + 2, 700.0); emit.xput(XD_HROT, 80.0);
                    emit.xpub(DUO_PT_W);
                    emit.xwait(DUO_PT_X);
                    (void)xv; (void)aux; (void)inl;
#else
                    rhs_duo<KIN, 1>(xv, 0, eng, inl, env_p, T, emit, aux);
#endif
                    if constexpr (X) {
                        if (sk.last) {
                            double z = z0;
                            asm volatile("" : "+v"(z));
                            const double P = 1 - z + z * z / 2 - z * z * z / 6 + z * z * z * z / 24;
#pragma unroll
                            for (int k = 0; k < NAL; k++) xa[k] = ca[k] + (xa[k] - ca[k]) * P;
                        }
                    }
                }
            } else duo_publish(sy, DUO_PT_W);   // (an evaluation nobody runs: role D must not wait for it)
            if constexpr (X) {
                if (tap) {
                    // f_periodic!(avionics, vehicle), the longitudinal half: role D has run f_step! on x_{n+1} and written the flags (DUO_F_CTL: the
                    // lanes that take part — a lane its evaluation has just handed over does not)
                    DUO_MARK(1, 13);   // arrives at U
                    duo_wait(sy, DUO_PT_U);
                    DUO_MARK(1, 14);   // past U
                    const bool ctl = flags_l[t] & DUO_F_CTL;
                    // (issue priority: this half is the longer one and runs ahead of role D's; role D waits for it at priority 0. Measured —
                    // profiles/r04_x2_update_ab.txt — the two halves do not overlap at role D's stepping priority: 0.156 ms per update of
                    // 524 288 aircraft against 0.078 + 0.061 for the halves alone; with this half ahead 0.115)
                    __builtin_amdgcn_s_setprio(FB_X2_LON_PRIO);
#ifdef FB_X2_SKIP_LON   // (timing diagnostics: one half of the update alone)
                    if (false) {
#else
                    if (ctl) {
#endif
                        const Ctl2 co = ctl_half(std::integral_constant<int, CTL_HALF_LON>{}, i);
                        ca[FB_ACT_THROTTLE] = co.c0; ca[FB_ACT_ELEVATOR] = co.c1;   // in force from the next stage on
                    }
                    DUO_MARK(1, 7);    // longitudinal half done
                    __builtin_amdgcn_s_setprio(0);
                    duo_publish<true>(sy, DUO_PT_F);
                }
                // the sums of the stage that normally comes next, while role D finishes its evaluation and keeps the book — unless a control
                // update has just run: role D's two new commands arrive with the next control word, and the sums are formed then, once
#ifndef FB_X2_NO_SPECULATION
                if (tap) sums_for = -1;
                else {
                    // (ahead of role D in issue priority: this wave must be back at the top, and through its state reads, before role D needs
                    // point R — at priority 0 it got there late: stepping alone 8.52 -> 8.33 ms, profiles/r04_x2_update_ab.txt)
                    __builtin_amdgcn_s_setprio(FB_X2_SPEC_PRIO);
                    form_sums((stg + 1) & 3, lds_off);
                    __builtin_amdgcn_s_setprio(0);
                }
#endif
            }
            sy.base += NPT;
        }
        if constexpr (X) {
            // the actuator positions of the lanes this launch commits (role D's bookkeeping word says which: it is final behind the EXIT word)
            const int d = dst_l[t];
            if (valid && (d & D_ACTIVE) && !(d & D_HANDOFF)) {
                bool bad = false;
#pragma unroll
                for (int k = 0; k < NAL; k++) { bad = bad || !isfinite(xa[k]); a.x[(int64_t)(X2_ACT + k) * a.n + i] = xa[k]; }
                if (bad) atomicOr(&a.status[i], (int32_t)FB_ST_NAN);
            }
        }
        // a wait of this role that ran into its bound (role D never arrived): the pair went on with a stale hand-over, so its aircraft
        // are flagged here too (role D flags them when ITS waits fail; atomics: both roles may write the word)
        if (__builtin_amdgcn_ballot_w64(sy.failed != 0) != 0 && valid) atomicOr(&a.status[i], (int32_t)FB_ST_NAN);
        return;
    }
    // ================= role D =================
    // The lane's bookkeeping state lives in an LDS word between evaluations (D_* bits): kept in registers it is what the allocator
    // spills around the evaluation, and the reloads land in the divergent bookkeeping code (tools/check_isa_spills.py).
    InputsDuoD in;
    in.pld_l = (lds_cptr)pld_l + t; in.aero_g = a.duo_pld + (valid ? i : 0); in.n = a.n; in.ui = 0;
    double accd_r[ND > DUO_NDL ? ND - DUO_NDL : 1];   // the stage sums that do not live in LDS (DUO_NDL)
#pragma unroll
    for (int k = 0; k < (ND > DUO_NDL ? ND - DUO_NDL : 1); k++) accd_r[k] = 0.0;
    {
        int stall = 0, eng = 0;
        bool to_ground = false;
        if (valid) {
#pragma unroll
            for (int k = 0; k < FB_NX; k++) {
                const double v = a.x[(int64_t)k * a.n + i];
                if (SV::skip(k)) to_ground = to_ground || (v != 0.0);
                else xs_l[SV::row(k) * B + t] = v;
            }
            if (to_ground) a.redo[i] = 1;
            stall = a.s[i]; eng = a.s[a.n + i];
            DUO_PHASE(5);
            if constexpr (X) {
                InputsXAgg in0;
                in0.xa = nullptr; in0.u_glob = a.u + i; in0.n = a.n; in0.ui = a.ui[i];
                sum_payload_of(in0);
                {   // the payload's sums: launch constants, to the rows role D fetches them from at every evaluation (the panel is role P's here)
                    const double pv[10] = {in0.pld_M, in0.pld_Mr[0], in0.pld_Mr[1], in0.pld_Mr[2], in0.pld_J[0], in0.pld_J[1], in0.pld_J[2], in0.pld_J[3], in0.pld_J[4], in0.pld_J[5]};
#pragma unroll
                    for (int k = 0; k < 10; k++) a.duo_pld[(int64_t)k * a.n + i] = pv[k];
                }
                in.ui = in0.ui;
                if (FB_X2_BAK && a.ctl_ratio > 0 && nsteps > 1 && !to_ground) {   // its half of the launch-start copy of the control-law record (role P copies cs)
                    copy_rows_batched<FB_NCU, FB_X2_BAK_G_CU>(a.ctl_bak + (int64_t)FB_NCS * a.n, a.cu, a.n, i);
                }
            } else {
                InputsAgg in0;
                load_inputs(a, i, in0); DUO_PHASE(6); in0.sum_aero((lds_cptr)lds + LDS_AERO, (lds_cptr)rk + LDS_AERO); in0.sum_payload();
                DUO_PHASE(7);
                pld_l[t] = in0.pld_M;
#pragma unroll
                for (int k = 0; k < 3; k++) pld_l[(1 + k) * B + t] = in0.pld_Mr[k];
#pragma unroll
                for (int k = 0; k < 6; k++) pld_l[(4 + k) * B + t] = in0.pld_J[k];
                const double ac[DUO_NCONST] = {in0.cd_in, in0.cd_df, in0.cy_in, in0.cl_in, in0.cl_df, in0.croll_in, in0.cm_in, in0.cn_in,
                                               (double)in0.l_df4.i, in0.l_df4.w, (double)in0.l_df2.i, in0.l_df2.w};
#pragma unroll
                for (int k = 0; k < DUO_NCONST; k++) a.duo_pld[(int64_t)k * a.n + i] = ac[k];
                in.ui = in0.ui;
            }
        }
#pragma unroll
        for (int k = 0; k < DUO_NDL; k++) accd_l[k * B + t] = 0.0;
        const bool active = valid && !to_ground;   // this launch owns the lane's state
        dst_l[t] = (active ? (D_ALIVE | D_ACTIVE) : 0) | (stall ? D_STALL : 0) | (eng << D_ENG_SHIFT);
        flags_l[t] = (active ? DUO_F_RUN : 0) | (eng << DUO_F_ENG_SHIFT);
    }
    DUO_PHASE(2);
    // the wave-uniform stage machine (k_step_air's)
    int stage = 0, step = 0;
    bool pending_cb = false, redoing = false, exit_ = __builtin_amdgcn_ballot_w64(dst_l[t] & D_ALIVE) == 0;
    [[maybe_unused]] bool tap_now = false, cmd_put = false;   // Cessna172Xv2: this evaluation is tapped / the update before it has put new lateral commands
    if ((threadIdx.x & 63) == 0) ctrl_l[pair] = exit_ ? DUO_C_EXIT : 0;
    // (this lane's launch constants — Sv0: aerodynamic, Xv2: payload — are read back from memory by THIS lane: program order is all that needs,
    // no fence. The device-scope __threadfence() that stood here wrote the L2 back — buffer_wbl2 — in every wave of every workgroup.)
#ifdef FB_DUO_START_FENCE
    __threadfence();
#elif !defined(FB_DUO_START_NOWAIT)
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the stores have been performed at this XCD's L2, which every later load of this workgroup goes to
#endif
#pragma unroll 1
    while (true) {
        DUO_MARK(2, 15);
        duo_publish(sy, DUO_PT_T);   // the control and flag words of this evaluation are written, every row of the previous one emitted
        DUO_MARK(2, 0);
        if (exit_) { DUO_PHASE(3); break; }
        const StageK sk = stage_k(stage);
        int lds_off = 0;
        asm volatile("" : "+s"(lds_off));
        const Tables T = {(lds_cptr)lds + lds_off, a.egm96, (lds_cptr)rk + lds_off, (gk_cptr)a.tables + lds_off};
        StepAux aux;
        aux.alpha = 0; aux.m_avail = 0; aux.wow = 0; aux.crash = 0;
        int32_t bits = 0;
        bool run = flags_l[t] & DUO_F_RUN;
        if (__builtin_amdgcn_ballot_w64(run) != 0) {
            if (run) {
                const int d0 = dst_l[t];
                typename std::conditional<X, InputsDuoDX, InputsDuoD>::type inl;
                static_cast<InputsDuoD&>(inl) = in;
                inl.aero_g = in.aero_g + lds_off; inl.pld_l = in.pld_l + lds_off;
                if constexpr (X) {
                    // base and stride of the payload rows, read from the kernel's arguments HERE, at the head of the evaluation (scalar loads that
                    // complete long before the fetch: held in SGPRs across the whole loop they are spilled)
                    const kargs_cptr ka = kernarg();
                    int64_t il = i;
                    asm volatile("" : "+v"(il));   // (opaque: the row addresses are formed here — hoisted, they are spilled and reloaded)
                    inl.aero_g = ka->duo_pld + il;
                    inl.n = ka->n;
                }
                const DuoEmit<2, X, DUO_NDL> emit = {(lds_cptr)xs_l, sk.xwr_l, (lds_ptr)accd_l, accd_r, (lds_ptr)xch_l, (lds_ptr)xc_l + 15 * B, sk.eb, sk.ee, sk.em, sk.last, t, &sy,
                                            tap_now, i};
                const SV xv = {sk.xrd_l + t + lds_off};
#if defined(FB_DUO_ONLY) && FB_DUO_ONLY == 1
                // timing diagnostic: role P alone on its SIMD — role D hands the velocity at the propeller over and waits
                emit.xput(XD_VP, 50.0); emit.xput(XD_VP + 1, 0.0); emit.xput(XD_VP + 2, 2.0);
                emit.xpub(DUO_PT_V); emit.xwait(DUO_PT_R); emit.xwait(DUO_PT_A); emit.xwait(DUO_PT_W); emit.xpub(DUO_PT_X);
                (void)xv; (void)inl; (void)d0;
#else
                Env env_d = a.env;   // role D reads wind_n / wind_e / wind_d and h_trn
                if constexpr (ENV_LDS) {   // (from its LDS panel, at every evaluation: role D has no registers to carry them; lds_off keeps the reads in the loop)
                    lds_cptr ev = (lds_cptr)envd_l + t + lds_off;
                    env_d.wind_n = ev[0 * B]; env_d.wind_e = ev[1 * B]; env_d.wind_d = ev[2 * B]; env_d.h_trn = ev[3 * B];
                } else if constexpr (PERENV) {   // (from memory: base and stride re-read from the kernel's arguments here, like the Xv2 instance's payload rows)
                    const kargs_cptr ka = kernarg();
                    int64_t il = i;
                    asm volatile("" : "+v"(il));
                    const double* e = ka->env_rows + il;
                    const int64_t en = ka->n;
                    env_d.wind_n = e[(int64_t)FB_ENV_WIND_N * en]; env_d.wind_e = e[(int64_t)FB_ENV_WIND_E * en]; env_d.wind_d = e[(int64_t)FB_ENV_WIND_D * en];
                    env_d.h_trn = e[(int64_t)FB_ENV_H_TERRAIN * en];
                }
                bits = rhs_duo<KIN, 2>(xv, (d0 & D_STALL) ? 1 : 0, (d0 >> D_ENG_SHIFT) & 3, inl, env_d, T, emit, aux);
#endif
            }
        } else duo_publish(sy, DUO_PT_X);   // (an evaluation nobody runs: role P must not wait for it)
        // (f_step!, below, modifies x_{n+1} in place at the end of a step's last evaluation: role P has read what it reads of it — its
        // point R, which this wave has waited for in the evaluation)
        int d = dst_l[t];
        // within reach of the ground, or an exception (altitude / ISA range): nothing is committed for this lane, the ground-capable pass
        // steps it again from the launch-start state and ends its simulation where the reference would (see k_step_air, `tkey`)
        if (run && bits != 0) { d = (d | D_HANDOFF) & ~D_ALIVE; run = false; bits = 0; }
        bool zero_acc = false, advance = true;
        [[maybe_unused]] const bool ctl_now = X && tap_now;   // this step closes a control period: f_periodic! follows its f_step! (FC/sim.jl:204-218)
        [[maybe_unused]] const bool ctl_lane = ctl_now && run;   // (per lane: a lane handed over by this very evaluation gets no update — nothing of it is committed)
        if (redoing) { redoing = false; run = d & D_ALIVE; }   // the lanes that sat out the re-evaluation of k1 join again
        else if (stage == 0 && pending_cb) {
            // f_step! on x_{n+1}, which sits in xs_l (this evaluation's emits have already moved xc_l on to the next stage)
            pending_cb = false;
            step++;
            bool mod = false;
            if (run) {
                auto renorm = [&](int k0, int len) {   // normalize_block!(v, 1e-8), kinematics.jl:114-118; WA :226-229, ECEF :317-320, NED: none
                    double q[4] = {0, 0, 0, 0}, n2 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) if (k < len) { q[k] = xs_l[SV::row(k0 + k) * B + t]; n2 += q[k] * q[k]; }
                    const double nr = ::sqrt(n2);
                    if (fabs(nr - 1.0) > 1e-8) {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (k < len) xs_l[SV::row(k0 + k) * B + t] = q[k] / nr;
                        mod = true;
                    }
                };
                if constexpr (KIN == FB_KIN_WA) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_EW, 4); }
                else if constexpr (KIN == FB_KIN_ECEF) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_WB + 4, 3); }
                int stall = (d & D_STALL) ? 1 : 0, eng = (d >> D_ENG_SHIFT) & 3;
                const int stall0 = stall, eng0 = eng;
                if (aux.alpha > c172::alpha_stall_hi) stall = 1;
                else if (aux.alpha < c172::alpha_stall_lo) stall = 0;
                const double w = xs_l[SV::row(FB_X_ENG_OMEGA) * B + t];
                const bool fuel = aux.m_avail > 0;
                const bool start = in.ui & FB_UI_ENG_START, stop = in.ui & FB_UI_ENG_STOP;
                if (eng == 0) { if (start) eng = 1; }
                else if (eng == 1) { if (!start) eng = 0; if (w > c172::w_idle && fuel) eng = 2; }
                else if (stop || w < c172::w_stall || !fuel) eng = 0;
                mod = mod || stall != stall0 || eng != eng0;
                d = (d & ~(D_STALL | (3 << D_ENG_SHIFT))) | (stall ? D_STALL : 0) | (eng << D_ENG_SHIFT);
            }
            if (step == nsteps || __builtin_amdgcn_ballot_w64(d & D_ALIVE) == 0) { exit_ = true; advance = false; }
            else if (__builtin_amdgcn_ballot_w64(mod) != 0) {   // k1 must be re-evaluated on the modified x_{n+1}
                if (mod) {
#pragma unroll
                    for (int k = 0; k < DUO_NDL; k++) accd_l[k * B + t] = 0.0;   // (acc held the discarded k1; stage 0 reads x_n from xs_l itself)
#pragma unroll
                    for (int k = 0; k < (ND > DUO_NDL ? ND - DUO_NDL : 1); k++) accd_r[k] = 0.0;
                }
                zero_acc = mod; run = mod; redoing = true; advance = false;
            }
        }
        if (advance) {
            stage = (stage + 1) & 3;
            pending_cb = (stage == 0);
        }
        dst_l[t] = d;
        flags_l[t] = (run ? DUO_F_RUN : 0) | (zero_acc ? DUO_F_ZERO_ACC : 0) | (((d >> D_ENG_SHIFT) & 3) << DUO_F_ENG_SHIFT) | (ctl_lane ? DUO_F_CTL : 0);
        if constexpr (X) {
            cmd_put = false;
            if (ctl_now) {
                // f_periodic!(avionics, vehicle), the lateral half, on the outputs both roles tapped from the evaluation above (x_{n+1} as it was
                // before f_step! renormalised the quaternions; the state rows the laws read — rates, filters, engine speed, altitude — f_step!
                // does not touch). Role P's wave runs the longitudinal half meanwhile.
                DUO_MARK(2, 12);   // f_step! done, flags written: at U
                duo_publish<true>(sy, DUO_PT_U);
                __builtin_amdgcn_s_setprio(FB_X2_SPLIT_PITCH ? FB_X2_PITCH_PRIO : FB_X2_UPD_PRIO);   // (its half is the shorter one: behind role P's, which runs at FB_X2_LON_PRIO; FB_X2_SPLIT_PITCH: ahead up to its point Z, CtlHand::put)
#ifdef FB_X2_SERIAL   // (timing diagnostic: the lateral half only after the longitudinal one has finished)
                duo_wait(sy, DUO_PT_F);
#endif
#ifdef FB_X2_SKIP_LAT
                if (false) {
#else
                if (ctl_lane) {
#endif
                    const Ctl2 co = ctl_half(std::integral_constant<int, CTL_HALF_LAT>{}, i);
                    xch_l[(XD_RHO - 6) * B + t] = co.c0; xch_l[(XD_HO - 6) * B + t] = co.c1;   // aileron, rudder commands -> role P (these rows are idle between evaluations)
                }
                cmd_put = true;
                DUO_MARK(2, 13);   // lateral half done
                __builtin_amdgcn_s_setprio(0);
                duo_wait(sy, DUO_PT_F);   // role P's half is done: the record is at rest
                __builtin_amdgcn_s_setprio(FB_DUO_PRIO_D);
#ifdef FB_X2_UPD_PRIO
                __builtin_amdgcn_s_setprio(FB_DUO_PRIO_D);
#endif
                DUO_MARK(2, 14);   // past F
            }
            // the evaluation to come is the last f_ode! of a step that closes a control period?
            tap_now = a.ctl_ratio > 0 && !exit_ && stage == 0 && pending_cb && !redoing && (a.ctl_phase + step + 1) % a.ctl_ratio == 0;
        }
        if ((threadIdx.x & 63) == 0)
            ctrl_l[pair] = stage | (exit_ ? DUO_C_EXIT : 0) | (tap_now ? DUO_C_TAP : 0) | (cmd_put ? DUO_C_CMD : 0);
        sy.base += NPT;
    }
    const int d = dst_l[t];
    if (!(d & D_ACTIVE)) return;
    if (d & D_HANDOFF) {
        if constexpr (X) {
            if (a.ctl_ratio > 0 && nsteps > 1) {   // nothing of this lane's launch is committed: undo the control-law updates it has made (both halves are at rest: point F)
#pragma unroll 1
                for (int k = 0; k < FB_NCS; k++) a.cs[(int64_t)k * a.n + i] = a.ctl_bak[(int64_t)k * a.n + i];
#pragma unroll 1
                for (int k = 0; k < FB_NCU; k++) const_cast<double*>(a.cu)[(int64_t)k * a.n + i] = a.ctl_bak[(int64_t)(FB_NCS + k) * a.n + i];
            }
        }
        a.redo[i] = 1;
        return;
    }
    // Nothing is carried across launches HERE, so the derivative the ground-capable pass may have left for this lane (k_step_air<X>'s k1) is stale once
    // this launch's steps are committed — and only then: a lane that is handed over (at entry, because its contact states are set; or at its first
    // evaluation within reach of the ground) has not moved, and the pass that takes it over starts from the k1 it stored itself at the end of the
    // launch before. (Up to round 6 the flag was cleared at entry for every lane: the ground-capable pass then evaluated k1 again in EVERY launch —
    // five evaluations where four do in a one-step launch, the shape of every scenario table evaluated after each step; tools/stamp_ground_launch.py.)
    if constexpr (X) { if (a.k1) a.k1_valid[i] = 0; }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < FB_NX; k++) {
        if (SV::skip(k)) continue;
        const double v = xs_l[SV::row(k) * B + t];
        bad = bad || !isfinite(v);
        a.x[(int64_t)k * a.n + i] = v;
    }
    if constexpr (X) {
        // the two brake actuators, which the airborne evaluation never reads, over the steps this launch has completed (every committed lane: `step`)
        const double z = dt / ACT_TAU, P = 1 - z + z * z / 2 - z * z * z / 6 + z * z * z * z / 24;
#pragma unroll
        for (int k = NAL; k < FB_NACT; k++) {
            const double c = x2_command(a, i, k), x0 = a.x[(int64_t)(X2_ACT + k) * a.n + i];
            double v = x0;
            for (int m = 0; m < step; m++) v = c + (v - c) * P;   // step by step: bit-identical to the per-step update
            bad = bad || !isfinite(v);
            a.x[(int64_t)(X2_ACT + k) * a.n + i] = v;
        }
    }
    if (bad || __builtin_amdgcn_ballot_w64(sy.failed != 0) != 0) atomicOr(&a.status[i], (int32_t)FB_ST_NAN);   // (sy.failed: a synchronisation wait ran into its bound — the partner wave never arrived)
    DUO_PHASE(4);
    a.s[i] = (d & D_STALL) ? 1 : 0;
    a.s[a.n + i] = (d >> D_ENG_SHIFT) & 3;
}

// ---- trim: f_init!(vehicle, TrimParameters) (FlightApps/src/c172/c172.jl:796-942) --------------
// Diagnostic builds (-DFB_TRIM_STAMP, tools/stamp_trim.py): where the first wave of k_trim spends its cycles — the time since the
// previous mark goes to bucket 24 + k (0 other, 1 residual evaluations, 2 the active-set solver, 3 serving finished lanes)
#if defined(FB_TRIM_STAMP)
__device__ unsigned long long g_trim_last;
__device__ __forceinline__ void trim_mark(int k) {
    if (blockIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        const int lane = threadIdx.x;
        if (__builtin_amdgcn_readfirstlane(lane) == lane) {
            const unsigned long long old = atomicExch(&g_trim_last, t);
            if (old != 0) { atomicAdd(&g_stamp_acc[24 + k], t - old); atomicAdd(&g_stamp_cnt[24 + k], 1ull); }
        }
    }
}
#define TRIM_MARK(k) trim_mark(k)
#else
#define TRIM_MARK(k) do { } while (0)
#endif
struct TrimP {
    v3 n_e;
    double h_e, psi_nb, EAS, gamma_wb_n, psi_wb_dot, theta_wb_dot, beta_a, fuel_load, mixture, flaps, payload[5];
};
// assign!(vehicle, params, state): trim unknowns -> (x, u, s)   (c172s.jl:227-263,168-220; c172.jl:825-854;
// aircraftbase.jl:76-86,110-118; kinematics.jl:155-178)
__device__ __forceinline__ void trim_assign(const TrimP& p, const double* z, const Env& env, const Tables& T, double (&x)[FB_NX], Inputs& in, double (&uraw)[FB_NU]) {
    using namespace c172;
    const double alpha_a = z[FB_TS_ALPHA_A], phi = z[FB_TS_PHI_NB];
    // atmosphere at Ob (ellipsoidal -> orthometric -> geopotential)
    const double h_o = p.h_e - geoid_height(T, p.n_e);
    double T_air, p_air;
    int32_t st = 0;
    double lnp_air;
    isa_data(h_o * wgs::a / (wgs::a + h_o), env.T_sl, env.p_sl, T_air, p_air, lnp_air, st);
    const double rho = p_air / (isa::R * T_air);
    const double TAS = p.EAS * sqrt(isa::rho_std / rho);
    const double cb = cos(p.beta_a);
    const v3 v_wb_b = {TAS * (cos(alpha_a) * cb), TAS * sin(p.beta_a), TAS * (sin(alpha_a) * cb)};
    // θ constraint (aircraftbase.jl:110-118)
    const double a_ = v_wb_b.x / TAS;
    const double b_ = (v_wb_b.y * sin(phi) + v_wb_b.z * cos(phi)) / TAS;
    const double sg = sin(p.gamma_wb_n);
    const double theta = atan((a_ * b_ + sg * sqrt(a_ * a_ + b_ * b_ - sg * sg)) / (a_ * a_ - sg * sg));
    // q_nb = Rz(ψ) ∘ Ry(θ) ∘ Rx(φ)
    double s1, c1, s2, c2, s3, c3;
    sincos(0.5 * p.psi_nb, &s1, &c1);
    sincos(0.5 * theta, &s2, &c2);
    sincos(0.5 * phi, &s3, &c3);
    const quat q_nb = qmul(qmul(quat{c1, 0, 0, s1}, quat{c2, 0, s2, 0}), quat{c3, s3, 0, 0});
    // ω_wb_b from Euler rates (attitude.jl:460-474), ė = (ψ̇, θ̇, 0)
    const double sth = sin(theta), cth = cos(theta), sph = sin(phi), cph = cos(phi);
    const v3 w_wb_b = {-sth * p.psi_wb_dot, cth * sph * p.psi_wb_dot + cph * p.theta_wb_dot, cth * cph * p.psi_wb_dot - sph * p.theta_wb_dot};
    const v3 v_eb_n = v3{env.wind_n, env.wind_e, env.wind_d} + qrot(q_nb, v_wb_b);
    // WA initialisation
    const double f_den = sqrt(1 - wgs::e2 * p.n_e.z * p.n_e.z);
    const double R_E = wgs::a / f_den, R_N = wgs::a * (1 - wgs::e2) / (f_den * f_den * f_den);
    const v3 w_ew_n = {v_eb_n.y / (R_E + p.h_e), -v_eb_n.x / (R_N + p.h_e), 0.0};
    const v3 w_eb_b = qrot_inv(q_nb, w_ew_n) + w_wb_b;
    const v3 v_eb_b = qrot_inv(q_nb, v_eb_n);
    // q_ew = ltf(n_e) = Rz(λ) ∘ Ry(-(ϕ + π/2)) ∘ Rz(0)   (geodesy.jl:132-135)
    const double lat = atan2(p.n_e.z, sqrt(p.n_e.x * p.n_e.x + p.n_e.y * p.n_e.y)), lon = atan2(p.n_e.y, p.n_e.x);
    double sl, cl, sp, cp;
    sincos(0.5 * lon, &sl, &cl);
    sincos(0.5 * (-(lat + 0.5 * PI)), &sp, &cp);
    const quat q_ew = qmul(quat{cl, 0, 0, sl}, quat{cp, 0, sp, 0});
#pragma unroll
    for (int k = 0; k < FB_NX; k++) x[k] = 0.0;
    x[FB_X_Q_WB] = q_nb.w; x[FB_X_Q_WB + 1] = q_nb.x; x[FB_X_Q_WB + 2] = q_nb.y; x[FB_X_Q_WB + 3] = q_nb.z;
    x[FB_X_Q_EW] = q_ew.w; x[FB_X_Q_EW + 1] = q_ew.x; x[FB_X_Q_EW + 2] = q_ew.y; x[FB_X_Q_EW + 3] = q_ew.z;
    x[FB_X_H_E] = p.h_e;
    x[FB_X_OMEGA_EB_B] = w_eb_b.x; x[FB_X_OMEGA_EB_B + 1] = w_eb_b.y; x[FB_X_OMEGA_EB_B + 2] = w_eb_b.z;
    x[FB_X_V_EB_B] = v_eb_b.x; x[FB_X_V_EB_B + 1] = v_eb_b.y; x[FB_X_V_EB_B + 2] = v_eb_b.z;
    x[FB_X_ENG_OMEGA] = z[FB_TS_N_ENG] * w_rated;
    x[FB_X_ALPHA_FILT] = alpha_a;
    x[FB_X_BETA_FILT] = p.beta_a;
    x[FB_X_FUEL] = fmin(fmax(p.fuel_load, 0.0), 1.0);
#pragma unroll
    for (int k = 0; k < FB_NU; k++) uraw[k] = 0.0;
    uraw[FB_U_THROTTLE] = z[FB_TS_THROTTLE]; uraw[FB_U_MIXTURE] = p.mixture;
    uraw[FB_U_AILERON] = z[FB_TS_AILERON]; uraw[FB_U_ELEVATOR] = z[FB_TS_ELEVATOR]; uraw[FB_U_RUDDER] = z[FB_TS_RUDDER];
    uraw[FB_U_FLAPS] = p.flaps;
#pragma unroll
    for (int k = 0; k < 5; k++) uraw[FB_U_M_PILOT + k] = p.payload[k];
    make_inputs(in, uraw, 1, FB_UI_MIXTURE_AUTO | FB_UI_STEERING_ENGAGED);
    in.u_glob = nullptr;  // trim is an airborne condition (c172s.jl:256 asserts no weight on wheels)
    in.n = 0;
}
// residuals whose squared sum is the reference's cost (c172.jl:857-867)
// (FB_TRIM_FAST_RHS: the evaluation in the stepping kernels' form — their atan2 / log / sincos, knot scans through scalar loads — instead of
// the single-call verbs' form with the library's: a third of the instructions, and the trimmed state is the zero of the very arithmetic
// that steps it. The trim state moves by ~1e-13 against the library form.)
#ifndef FB_TRIM_FAST_RHS
#define FB_TRIM_FAST_RHS true
#endif
// GROUND = false: the airborne-only evaluation (388 registers against 504); answers FB_ST_INTERNAL_REDO where a wheel could reach the ground
template <bool GROUND = true>
__device__ __forceinline__ int32_t trim_resid_body(const TrimP& p, const double* z, const Env& env, const Tables& T, double* r) {
    TRIM_MARK(0);
    double x[FB_NX], xd[FB_NX];
    Inputs in;
    double uraw[FB_NU];
    trim_assign(p, z, env, T, x, in, uraw);
    StepAux aux;
    const int32_t st = rhs<FB_KIN_WA, GROUND, FB_TRIM_FAST_RHS>(x, 0, 2, in, env, T, [&](int j, double v) { xd[j] = v; }, aux, NoSink{});
    const double nv = sqrt(x[FB_X_V_EB_B] * x[FB_X_V_EB_B] + x[FB_X_V_EB_B + 1] * x[FB_X_V_EB_B + 1] + x[FB_X_V_EB_B + 2] * x[FB_X_V_EB_B + 2]);
    r[0] = xd[FB_X_V_EB_B] / nv; r[1] = xd[FB_X_V_EB_B + 1] / nv; r[2] = xd[FB_X_V_EB_B + 2] / nv;
    r[3] = xd[FB_X_OMEGA_EB_B]; r[4] = xd[FB_X_OMEGA_EB_B + 1]; r[5] = xd[FB_X_OMEGA_EB_B + 2];
    r[6] = xd[FB_X_ENG_OMEGA] / c172::w_rated;
    TRIM_MARK(1);
    return st;
}
// ---- the trim solver ---------------------------------------------------------------------------------------------
// The reference minimises cost = Σ r² with NLopt's :LN_BOBYQA inside box bounds, initial_step 0.05, stopval 1e-16,
// maxeval 1e5 (c172.jl:883-942). NLopt is a third-party optimiser; what is kept of it is its behaviour on this
// zero-residual problem — a trust region that starts at the initial step, never leaves the box and walks a continuous
// descent path from TrimState() — implemented as a bound-constrained Gauss-Newton trust-region iteration, one lane per
// aircraft:
//   * box trust region |dz|∞ <= D (D0 = 0.05) intersected with the bounds; inside it |r + J dz|² is minimised by an
//     active set (a variable that would leave the box is held at the face it hits and released when the model
//     gradient there points back inside), so an iterate slides along a bound instead of being clamped onto it;
//   * J by central differences (one-sided at a bound). The engine and aerodynamic maps are piecewise linear: where
//     the forward and backward differences of a column disagree (a knot within one difference step of the iterate,
//     e.g. n_eng = 1.074, piston.jl:110,140) the column is replaced by the pure one-sided slopes one step further
//     out, the step uses the side it moves to (semismooth Newton), and the mirror choice is tried too;
//   * D doubles after a well-predicted step, shrinks after a rejected one; iteration continues to the rounding
//     floor (cost <= 1e-27) so that the trim does not depend on the path; success <=> cost <= stopval = 1e-16.
// If the descent from TrimState() ends above stopval (a V-shaped minimum of the cost on a table knot), the solve is
// repeated by continuation in the trim PARAMETERS from the reference's default TrimParameters() (EAS 50, h 1050, ...:
// c172.jl:806-818, whose trim from TrimState() the reference's own test pins) towards the requested ones.
constexpr int TRIM_N = 7;
// The active-set solve inside the box. Every array index below is a compile-time constant after unrolling — the free set is a MASK, not
// a compacted index list, rows are exchanged by selects — so the 7 x 8 system, H, g and the step live in registers (round 3's form
// indexed them dynamically: 3.8 KB of scratch per lane, and the solver, not the ~140 residual evaluations, was where k_trim's time
// went: a dependent scratch round trip per matrix element). The ARITHMETIC is the compacted algorithm's, operation by operation and in
// its order (free variables in index order; partial pivoting over the free rows below, first maximum; elimination over the free columns
// from the pivot's on; back-substitution from the last free variable), which is also the oracle's (oracle/fo_trim.hpp): same bits.
__device__ __forceinline__ void trim_box_gauss_newton(const double (&H)[TRIM_N][TRIM_N], const double (&g)[TRIM_N], const double (&dl)[TRIM_N], const double (&du)[TRIM_N],
                                                      double ridge, double (&d)[TRIM_N]) {
    constexpr int N = TRIM_N;
    int fixed[N];
#pragma unroll
    for (int k = 0; k < N; k++) { d[k] = 0; fixed[k] = 0; }
#pragma unroll 1
    for (int pass = 0; pass < 4 * N; pass++) {
        double A[N][N + 1];   // rows / columns of fixed variables are never read
#pragma unroll
        for (int a = 0; a < N; a++) {   // H_FF d_F = -(g_F + H_FB d_B)
            double rhs = -g[a];
#pragma unroll
            for (int k = 0; k < N; k++) rhs = fixed[k] ? rhs - H[a][k] * d[k] : rhs;
#pragma unroll
            for (int b = 0; b < N; b++) A[a][b] = H[a][b] + (a == b ? ridge : 0.0);
            A[a][N] = rhs;
        }
#pragma unroll
        for (int c = 0; c < N; c++) {   // Gaussian elimination over the free variables, partial pivoting among the free rows at and below c
            if (fixed[c]) continue;
            int pv = c;
            double best = fabs(A[c][c]);
#pragma unroll
            for (int q = c + 1; q < N; q++) {
                const double v = fabs(A[q][c]);
                const bool take = !fixed[q] && v > best;
                pv = take ? q : pv; best = take ? v : best;
            }
#pragma unroll
            for (int q = c + 1; q < N; q++) {
                if (q == pv) {   // (per lane; a wave without such a lane skips it)
#pragma unroll
                    for (int w = 0; w <= N; w++) { const double t = A[q][w]; A[q][w] = A[c][w]; A[c][w] = t; }
                }
            }
            const double piv = A[c][c] != 0 ? A[c][c] : 1e-300;
#pragma unroll
            for (int q = c + 1; q < N; q++) {
                if (fixed[q]) continue;
                const double f = A[q][c] / piv;
#pragma unroll
                for (int w = c; w < N; w++) A[q][w] = fixed[w] ? A[q][w] : A[q][w] - f * A[c][w];
                A[q][N] -= f * A[c][N];
            }
        }
        double sol[N];
#pragma unroll
        for (int q = N - 1; q >= 0; q--) {
            double sum = A[q][N];
#pragma unroll
            for (int w = q + 1; w < N; w++) sum = fixed[w] ? sum : sum - A[q][w] * sol[w];
            sol[q] = fixed[q] ? 0.0 : sum / (A[q][q] != 0 ? A[q][q] : 1e-300);
        }
        double t = 1.0;   // longest feasible fraction of the move towards the free minimum
        int hit = -1, side = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            if (fixed[k]) continue;
            const double delta = sol[k] - d[k];
            if (delta > 0 && d[k] + delta > du[k]) { const double tt = (du[k] - d[k]) / delta; if (tt < t) { t = tt; hit = k; side = 1; } }
            if (delta < 0 && d[k] + delta < dl[k]) { const double tt = (dl[k] - d[k]) / delta; if (tt < t) { t = tt; hit = k; side = -1; } }
        }
#pragma unroll
        for (int k = 0; k < N; k++) d[k] = fixed[k] ? d[k] : d[k] + t * (sol[k] - d[k]);
        if (hit >= 0) {
#pragma unroll
            for (int k = 0; k < N; k++) if (k == hit) { fixed[k] = side; d[k] = side > 0 ? du[k] : dl[k]; }
            continue;
        }
        int rel = -1;
        double best = 0, g_rel = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            if (!fixed[k]) continue;
            double gm = g[k];
#pragma unroll
            for (int b = 0; b < N; b++) gm += H[k][b] * d[b];
            const double inward = fixed[k] > 0 ? gm : -gm;   // at the upper face a positive gradient wants to come back
            if (inward > best) { best = inward; rel = k; g_rel = g[k]; }
        }
        if (rel < 0 || best <= 1e-14 * (fabs(g_rel) + 1e-300)) break;
#pragma unroll
        for (int k = 0; k < N; k++) if (k == rel) fixed[k] = 0;
    }
}
// One aircraft's descent between iterations (k_trim keeps one per lane, so that a lane whose aircraft has converged takes the next aircraft while
// its neighbours go on iterating). Every loop over the seven unknowns in k_trim is unrolled, so that H, g and the step are indexed by constants
// (registers), and a lane-dependent index (the lane's Jacobian column) is a select chain or an address in the kernel's workspace.
struct TrimLane { double z[TRIM_N], r[TRIM_N], cost, D; int it; };
constexpr double TRIM_COST_FLOOR = 1e-27;
__device__ __forceinline__ bool trim_tr_goes_on(const TrimLane& S, int max_iter) { return S.it < max_iter && S.cost > TRIM_COST_FLOOR && S.D > 1e-13; }
// f_init!(vehicle, TrimParameters) (c172.jl:883-942), one lane per aircraft at a time.
__device__ __forceinline__ void trim_load_params(TrimP& p, const double* tp, int64_t n, int64_t i) {
    p.n_e = {tp[(int64_t)FB_TP_N_E * n + i], tp[(int64_t)(FB_TP_N_E + 1) * n + i], tp[(int64_t)(FB_TP_N_E + 2) * n + i]};
    p.h_e = tp[(int64_t)FB_TP_H_E * n + i]; p.psi_nb = tp[(int64_t)FB_TP_PSI_NB * n + i]; p.EAS = tp[(int64_t)FB_TP_EAS * n + i];
    p.gamma_wb_n = tp[(int64_t)FB_TP_GAMMA_WB_N * n + i]; p.psi_wb_dot = tp[(int64_t)FB_TP_PSI_WB_DOT * n + i];
    p.theta_wb_dot = tp[(int64_t)FB_TP_THETA_WB_DOT * n + i]; p.beta_a = tp[(int64_t)FB_TP_BETA_A * n + i];
    p.fuel_load = tp[(int64_t)FB_TP_FUEL_LOAD * n + i]; p.mixture = tp[(int64_t)FB_TP_MIXTURE * n + i]; p.flaps = tp[(int64_t)FB_TP_FLAPS * n + i];
    for (int k = 0; k < 5; k++) p.payload[k] = tp[(int64_t)(FB_TP_PAYLOAD + k) * n + i];
}
// assign!(vehicle, params, state_opt): leave the trimmed initial condition in x, u, s, the trim state in ts
__device__ __forceinline__ void trim_leave_body(const KArgs& a, const Env& env, const TrimP& p, const double (&z)[TRIM_N], const Tables& T, double* ts, int32_t* success, double* cost_out, double cost, int64_t i) {
    const int64_t n = a.n;
    double x[FB_NX], uraw[FB_NU];
    Inputs in;
    trim_assign(p, z, env, T, x, in, uraw);
    for (int k = 0; k < FB_NX; k++) a.x[(int64_t)k * n + i] = x[k];
    double* uw = const_cast<double*>(a.u);
    for (int k = 0; k < FB_NU; k++) uw[(int64_t)k * n + i] = uraw[k];
    const_cast<int32_t*>(a.ui)[i] = in.ui;
    a.s[i] = 0;        // stall = false
    a.s[n + i] = 2;    // EngineState.running
    for (int k = 0; k < TRIM_N; k++) ts[(int64_t)k * n + i] = z[k];
    if (success) success[i] = cost <= 1e-16;   // the reference's criterion: STOPVAL_REACHED, stopval = 1e-16 (c172.jl:926,934)
    if (cost_out) cost_out[i] = cost;
}
constexpr double TRIM_LO[TRIM_N] = {-PI / 12, -PI / 3, 0.4, 0, -1, -1, -1};                 // c172.jl:901-908
constexpr double TRIM_HI[TRIM_N] = {c172::alpha_stall_hi, PI / 3, 1.1, 1, 1, 1, 1};         // c172.jl:910-917
constexpr int TRIM_MAX_ITER = 500;
#ifndef FB_TRIM_REFILL_MIN
#define FB_TRIM_REFILL_MIN 16   // (4: 77.6 ms, 8: 73.8, 16: 72.0, 24: 74.5 per 1 048 576 aircraft of the bench lattice)
#endif
// The descent from the given trim state, for every aircraft.
//  * PERSISTENT: the aircraft differ in how many iterations they take (bench lattice: 64 to 606 residual evaluations, mean 144 — a wave that
//    trimmed 64 aircraft side by side waited for its slowest, ~420). One wave per SIMD; a wave takes aircraft from a queue (`next`, zeroed by the
//    host), every lane descends on its own aircraft, and when FB_TRIM_REFILL_MIN lanes have finished theirs they are served together: results
//    written, the next aircraft taken.
//  * ONE loop around ONE inlined residual evaluation, no call: with the residual out of line (an `f_ode!`: ~7.6 k instructions, every VGPR) each
//    call saved and restored the callee-saved registers and the caller's arrays through scratch — 800 scratch accesses per evaluation, a scratch
//    frame of 3.5 KB per lane whose 230 MB did not fit the L2: 138 k cycles per evaluation, 280 GB of HBM traffic per launch
//    (tools/stamp_trim.py, tools/pmc_trim.sh). Here an iteration's two parts are phases of a single loop — phase 0, the Jacobian: every
//    lane asks for the residual at its next difference point (its own column, its own kink probes; all lanes evaluate together, each at its
//    own point); phase 1, the trial steps: every lane forms its candidate step (the active-set solver) and asks for the residual there —
//    and a lane that has just begun a descent asks for the residual at the initial state first. The three Jacobians (central, forward,
//    backward: 147 values per lane, indexed by the lane's column) live in a workspace in memory, [row][lane]: written once per column, read once
//    per candidate. The algorithm, statement by statement, is oracle/fo_trim.hpp's.
//  * THE CONTINUATION FALLBACK is a sequence of descents of the same lane: when the descent from the given state ends above stopval, the lane
//    restarts from the given state with the parameters blended towards TrimParameters() (c172.jl:806-818; location and heading as requested)
//    and walks the blend back to the requested ones in steps that halve on failure and double on success.
// workspace rows per lane: the TrimParameters of the descent in progress; Jc | Jf | Jb; three trim states of the continuation (the given one, the
// first descent's result, the last good continuation point)
constexpr int TRIM_WS_J = FB_NTP, TRIM_WS_Z = TRIM_WS_J + 3 * TRIM_N * TRIM_N, TRIM_WS_ENV = TRIM_WS_Z + 3 * TRIM_N, TRIM_WS_ROWS = TRIM_WS_ENV + ENV_DEV_ROWS;   // (the last rows: the lane's aircraft's environment, k_trim<true>)
template <class A7>
__device__ __forceinline__ double pick7(const A7& v, int k) {   // v[k] for a lane-dependent k: a select chain (a dynamic register index would be scratch)
    double x = v[0];
#pragma unroll
    for (int m = 1; m < TRIM_N; m++) x = (k == m) ? v[m] : x;
    return x;
}
// PERENV: every aircraft in its own environment (fb_set_env): a lane parks its aircraft's rows in the workspace next to the trim parameters
// when it takes the aircraft from the queue and reads them back where the residual is evaluated (the parameters' own way)
template <bool PERENV>
__global__ __launch_bounds__(64) void k_trim(KArgs a, const double* tp, double* ts, int32_t* success, double* cost_out, unsigned long long* next, double* ws) {
    __shared__ double lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ double rk[LDS_RK_DOUBLES];
    stage_tables<PR_NC_STEP>(lds, rk, a.tables);
    const Tables T = {(lds_cptr)lds, a.egm96, (lds_cptr)rk, (gk_cptr)a.tables};
    constexpr int N = TRIM_N;
    const int64_t n = a.n;
    const int lane = threadIdx.x;
    const int64_t W = (int64_t)gridDim.x * 64, slot = (int64_t)blockIdx.x * 64 + lane;
    // The lane's workspace rows are ws[row W + slot]. Their ~170 addresses are invariant in the loop below; formed ahead of it and common to every use
    // they had nowhere to live: ~140 of them were the kernel's 1 168 B of scratch, each use a scratch_load in front of the row's own load. The lane's
    // slot is therefore opaque at every access (the INDEX, not the pointer: a laundered pointer loses its address space and the accesses turn
    // into flat_load / flat_store): an address is one 64-bit multiply-add where the row is used.
    auto wsb = [&]() -> double* { int64_t sl = slot; asm volatile("" : "+v"(sl)); return ws + sl; };
    auto Jrow = [&](int m, int i, int j) -> double& { return wsb()[(int64_t)(TRIM_WS_J + (m * N + i) * N + j) * W]; };
    auto Zrow = [&](int v, int k) -> double& { return wsb()[(int64_t)(TRIM_WS_Z + v * N + k) * W]; };   // v: 0 the given state, 1 the first descent's result, 2 the last good continuation point
    auto env_now = [&]() -> Env {
        if constexpr (PERENV) {
            auto E = [&](int k) { return wsb()[(int64_t)(TRIM_WS_ENV + k) * W]; };
            return {E(FB_ENV_T_SL), E(FB_ENV_P_SL), E(FB_ENV_WIND_N), E(FB_ENV_WIND_E), E(FB_ENV_WIND_D), E(FB_ENV_H_TERRAIN), a.env.surface, E(ENV_DEV_LN_P), E(ENV_DEV_K_RT)};
        } else return a.env;
    };
    const double lo[N] = {TRIM_LO[0], TRIM_LO[1], TRIM_LO[2], TRIM_LO[3], TRIM_LO[4], TRIM_LO[5], TRIM_LO[6]};
    const double hi[N] = {TRIM_HI[0], TRIM_HI[1], TRIM_HI[2], TRIM_HI[3], TRIM_HI[4], TRIM_HI[5], TRIM_HI[6]};
    const double fd = 1e-6;
    TrimLane S;
    int64_t i = -1;        // the lane's aircraft; with !active: finished, results not yet written
    bool active = false;
    bool fresh = false;    // a descent has just begun: its residual at the initial state is not there yet
    int maxit = TRIM_MAX_ITER;             // the descent's iteration limit (c172.jl's maxeval is never reached)
    int mode = 0;                          // 0: the descent from the given state; 1: the continuation's first descent (blend 0); 2: its later ones
    double ct = 0, cdt = 0, tn_last = 0, cost_main = 0;   // continuation: the blend reached, its step, the blend of the descent in progress, the first descent's cost
    bool more = true;      // (wave-uniform) the queue may still hold aircraft
#pragma unroll 1
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
        const int n_idle = __builtin_popcountll(idle);
        if (n_idle == 64 || (more && n_idle >= FB_TRIM_REFILL_MIN)) {
            TRIM_MARK(0);
            if (!active && i >= 0) {   // assign!(vehicle, params, state_opt) with the REQUESTED parameters
                TrimP p;
                trim_load_params(p, tp, n, i);
                trim_leave_body(a, env_now(), p, S.z, T, ts, success, cost_out, S.cost, i);
                i = -1;
            }
            if (more) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(next, (unsigned long long)n_idle);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
                more = base + (unsigned long long)n_idle < (unsigned long long)n;
                if (!active) {
                    const int64_t mine = (int64_t)base + __builtin_popcountll(idle & ((1ull << lane) - 1));
                    if (mine < n) {
                        i = mine;
                        for (int k = 0; k < FB_NTP; k++) wsb()[(int64_t)k * W] = tp[(int64_t)k * n + i];
                        if constexpr (PERENV) for (int k = 0; k < ENV_DEV_ROWS; k++) wsb()[(int64_t)(TRIM_WS_ENV + k) * W] = a.env_rows[(int64_t)k * n + i];
#pragma unroll
                        for (int k = 0; k < N; k++) { S.z[k] = ts[(int64_t)k * n + i]; Zrow(0, k) = S.z[k]; }
                        active = true; fresh = true; mode = 0; maxit = TRIM_MAX_ITER;
                    }
                }
            }
            TRIM_MARK(3);
            if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        }
        // ---- one iteration of every active lane's descent (oracle/fo_trim.hpp: trim_tr_minimize's loop body; for a lane that has just begun a descent its first residual before) ----
        bool it_on = active && (fresh || trim_tr_goes_on(S, maxit));
        int phase = 0;                                   // (wave-uniform) 0: the Jacobian, 1: the trial steps
        // phase 0, per lane: the column, and which of its points is asked for next (0: z + fd, 1: z - fd, 2 / 3: the kink probes one step further out)
        int col = fresh ? -1 : 0, sub = 0;               // (col -1: the residual at the initial state)
        uint32_t kink = 0;                               // bit j: the forward and backward slopes of column j disagree
        double rp[N], rm[N], zp = 0, zm = 0;
        // phase 1, per lane
        bool accepted = false, tdone = true, have = false;
        int attempt = 0, cand = 0;
        uint32_t side0p = 0;                             // bit j: candidate 0 ended on the forward side of kink column j
        double best_cn = 0, best_pred = 0, best_dinf = 0, best_zn[N], best_rn[N];
#pragma unroll
        for (int k = 0; k < N; k++) { rp[k] = 0; rm[k] = 0; best_zn[k] = 0; best_rn[k] = 0; }
#pragma unroll 1
        for (;;) {
            // ---- which point does the lane want the residual at?
            bool need = false;
            double zq[N], pred = 0, dinf = 0;
#pragma unroll
            for (int k = 0; k < N; k++) zq[k] = S.z[k];
            if (phase == 0) {
                need = it_on && col < N;
                if (need && col < 0) {
#pragma unroll
                    for (int k = 0; k < N; k++) { S.z[k] = fmin(fmax(S.z[k], lo[k]), hi[k]); zq[k] = S.z[k]; }   // (the descent begins inside the box)
                }
                if (need && col >= 0) {
                    if (sub == 0) { const double zj = pick7(S.z, col); zp = fmin(zj + fd, pick7(hi, col)); zm = fmax(zj - fd, pick7(lo, col)); }
                    const double tgt = sub == 0 ? zp : sub == 1 ? zm : sub == 2 ? zp + fd : zm - fd;
#pragma unroll
                    for (int k = 0; k < N; k++) zq[k] = (k == col) ? tgt : zq[k];
                }
                if (__builtin_amdgcn_ballot_w64(need) == 0) {   // every column of every lane is there: on to the trial steps
                    phase = 1;
                    tdone = !it_on;
                    continue;
                }
            } else {
                if (__builtin_amdgcn_ballot_w64(!tdone) == 0) break;
                if (!tdone) {
                    double dl[N], du[N];
#pragma unroll
                    for (int k = 0; k < N; k++) { dl[k] = fmax(lo[k] - S.z[k], -S.D); du[k] = fmin(hi[k] - S.z[k], S.D); }
                    double J[N][N], H[N][N], g[N], d[N];
                    int side[N];
#pragma unroll
                    for (int j = 0; j < N; j++) {
                        side[j] = cand == 0 ? 0 : ((kink >> j) & 1 ? ((side0p >> j) & 1 ? -1 : 1) : 0);
#pragma unroll
                        for (int r_ = 0; r_ < N; r_++) J[r_][j] = Jrow(side[j] == 0 ? 0 : side[j] > 0 ? 1 : 2, r_, j);
                    }
#pragma unroll 1
                    for (int round = 0; round < (cand == 0 ? 3 : 1); round++) {
                        double tr = 0;
#pragma unroll
                        for (int c = 0; c < N; c++) {
                            g[c] = 0;
#pragma unroll
                            for (int r_ = 0; r_ < N; r_++) g[c] += J[r_][c] * S.r[r_];
#pragma unroll
                            for (int b = 0; b < N; b++) {
                                double sum = 0;
#pragma unroll
                                for (int r_ = 0; r_ < N; r_++) sum += J[r_][c] * J[r_][b];
                                H[c][b] = sum;
                            }
                            tr += H[c][c];
                        }
                        TRIM_MARK(0);
                        trim_box_gauss_newton(H, g, dl, du, 1e-14 * tr + 1e-300, d);
                        TRIM_MARK(2);
                        if (cand != 0) break;
                        bool changed = false;
#pragma unroll
                        for (int j = 0; j < N; j++) if ((kink >> j) & 1) {
                            const int want = d[j] > 0 ? 1 : d[j] < 0 ? -1 : (side[j] != 0 ? side[j] : 1);
                            if (want != side[j]) {
                                side[j] = want; changed = true;
#pragma unroll
                                for (int r_ = 0; r_ < N; r_++) J[r_][j] = Jrow(want > 0 ? 1 : 2, r_, j);
                            }
                        }
                        if (!changed) break;
                    }
                    if (cand == 0) {
                        side0p = 0;
#pragma unroll
                        for (int j = 0; j < N; j++) side0p |= side[j] > 0 ? 1u << j : 0u;
                    }
#pragma unroll
                    for (int c = 0; c < N; c++) {
                        double Hd = 0;
#pragma unroll
                        for (int b = 0; b < N; b++) Hd += H[c][b] * d[b];
                        pred -= d[c] * (2 * g[c] + Hd);
                        dinf = fmax(dinf, fabs(d[c]));
                    }
                    if (pred > 0 && dinf != 0) {
                        need = true;
#pragma unroll
                        for (int k = 0; k < N; k++) zq[k] = fmin(fmax(S.z[k] + d[k], lo[k]), hi[k]);
                    }
                }
            }
            // ---- the residuals, every lane at its own point
            double rq[N];
#pragma unroll
            for (int k = 0; k < N; k++) rq[k] = 0;
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
                if (need) {
                    TrimP p;
                    trim_load_params(p, wsb(), W, 0);   // (the lane's parked parameters: through the opaque base, like every workspace row)
                    trim_resid_body<true>(p, zq, env_now(), T, rq);
                }
            }
            // ---- what the lane does with them
            if (phase == 0) {
                if (need) {
                    if (col < 0) {   // the descent's first residual
                        double c0 = 0;
#pragma unroll
                        for (int k = 0; k < N; k++) { S.r[k] = rq[k]; c0 += rq[k] * rq[k]; }
                        S.cost = c0; S.D = 0.05; S.it = 0;   // (0.05: the reference's initial_step, c172.jl:919)
                        fresh = false;
                        it_on = trim_tr_goes_on(S, maxit);
                        col = 0;
                    } else if (sub == 0) {
#pragma unroll
                        for (int k = 0; k < N; k++) rp[k] = rq[k];
                        sub = 1;
                    } else if (sub == 1) {
#pragma unroll
                        for (int k = 0; k < N; k++) rm[k] = rq[k];
                        const double zj = pick7(S.z, col);
                        const double ic = 1.0 / (zp - zm);
                        const double ifw = zp > zj ? 1.0 / (zp - zj) : 0.0, ibw = zj > zm ? 1.0 / (zj - zm) : 0.0;
                        double dmax = 0, cmax = 0;
#pragma unroll
                        for (int r_ = 0; r_ < N; r_++) {
                            const double jc = (rp[r_] - rm[r_]) * ic;
                            const double jf = ifw != 0 ? (rp[r_] - S.r[r_]) * ifw : jc;
                            const double jb = ibw != 0 ? (S.r[r_] - rm[r_]) * ibw : jc;
                            dmax = fmax(dmax, fabs(jf - jb));
                            cmax = fmax(cmax, fabs(jc));
                            Jrow(0, r_, col) = jc; Jrow(1, r_, col) = jf; Jrow(2, r_, col) = jb;
                        }
                        const bool kk = dmax > 1e-3 * cmax;   // smooth: |Jf - Jb| ~ fd |r''| ~ 1e-6 of the column
                        if (kk) kink |= 1u << col;
                        if (kk && zp + fd <= pick7(hi, col)) sub = 2;
                        else if (kk && zm - fd >= pick7(lo, col)) sub = 3;
                        else { col++; sub = 0; }
                    } else if (sub == 2) {
#pragma unroll
                        for (int r_ = 0; r_ < N; r_++) Jrow(1, r_, col) = (rq[r_] - rp[r_]) / (zp + fd - zp);
                        if (zm - fd >= pick7(lo, col)) sub = 3;
                        else { col++; sub = 0; }
                    } else {
#pragma unroll
                        for (int r_ = 0; r_ < N; r_++) Jrow(2, r_, col) = (rm[r_] - rq[r_]) / (zm - (zm - fd));
                        col++; sub = 0;
                    }
                }
            } else if (!tdone) {
                if (need) {
                    double cn = 0;
#pragma unroll
                    for (int k = 0; k < N; k++) cn += rq[k] * rq[k];
                    if (!have || cn < best_cn) {
                        have = true; best_cn = cn; best_pred = pred; best_dinf = dinf;
#pragma unroll
                        for (int k = 0; k < N; k++) { best_zn[k] = zq[k]; best_rn[k] = rq[k]; }
                    }
                }
                cand++;
                if (cand >= (kink != 0 ? 2 : 1)) {   // the attempt's candidates are through
                    if (have) {
                        const double rho = (S.cost - best_cn) / best_pred;
                        if (best_cn < S.cost) {
#pragma unroll
                            for (int k = 0; k < N; k++) { S.z[k] = best_zn[k]; S.r[k] = best_rn[k]; }
                            S.cost = best_cn;
                            accepted = true;
                            if (rho > 0.75 && best_dinf > 0.9 * S.D) S.D = fmin(2 * S.D, 1.0);
                            else if (rho < 0.25) S.D = fmax(0.5 * best_dinf, 1e-14);
                        } else {
                            S.D = 0.25 * fmin(S.D, best_dinf);
                        }
                    } else S.D *= 0.25;
                    attempt++; cand = 0; have = false; side0p = 0;
                    tdone = accepted || attempt >= 40 || !(S.D > 1e-13);
                }
            }
        }
        if (active) {
            bool ended = true;
            if (it_on) { S.it++; ended = !(accepted && trim_tr_goes_on(S, maxit)); }
            if (ended) {   // the descent is over: S.z, S.cost. Done, or the next descent of the continuation
                const double d0[14] = {1050.0, 50.0, 0.0, 0.0, 0.0, 0.0, 0.5, 0.5, 0.0, 75.0, 75.0, 0.0, 0.0, 50.0};   // TrimParameters(), c172.jl:806-818
                bool again = false;
                double tn = 0;
                if (mode == 0) {
                    if (S.cost > 1e-16) {   // from TrimParameters() towards the requested ones, beginning at blend 0 from the given state
                        cost_main = S.cost;
#pragma unroll
                        for (int k = 0; k < N; k++) { Zrow(1, k) = S.z[k]; S.z[k] = Zrow(0, k); }
                        ct = 0; cdt = 0.125; tn = 0; maxit = 200; mode = 1; again = true;
                    }
                } else {
                    bool good;
                    if (mode == 1) {
                        good = S.cost <= 1e-16; mode = 2;
                        if (good) {
#pragma unroll
                            for (int k = 0; k < N; k++) Zrow(2, k) = S.z[k];
                        }
                    } else if (S.cost <= 1e-16) {
                        good = true; ct = tn_last; cdt = fmin(2 * cdt, 0.25);
#pragma unroll
                        for (int k = 0; k < N; k++) Zrow(2, k) = S.z[k];
                    } else { cdt *= 0.5; good = !(cdt < 1.0 / 1024); }
                    if (good && ct < 1.0) {
                        tn = fmin(1.0, ct + cdt); maxit = 60; again = true;
#pragma unroll
                        for (int k = 0; k < N; k++) S.z[k] = Zrow(2, k);
                    } else if (good && S.cost < cost_main) {   // the requested parameters reached (the last descent, at blend 1, ended below stopval)
#pragma unroll
                        for (int k = 0; k < N; k++) S.z[k] = Zrow(2, k);
                    } else {   // no path: the first descent's result stands
#pragma unroll
                        for (int k = 0; k < N; k++) S.z[k] = Zrow(1, k);
                        S.cost = cost_main;
                    }
                }
                if (again) {
                    tn_last = tn;
                    constexpr int row[14] = {FB_TP_H_E, FB_TP_EAS, FB_TP_GAMMA_WB_N, FB_TP_PSI_WB_DOT, FB_TP_THETA_WB_DOT, FB_TP_BETA_A, FB_TP_FUEL_LOAD, FB_TP_MIXTURE,
                                             FB_TP_FLAPS, FB_TP_PAYLOAD, FB_TP_PAYLOAD + 1, FB_TP_PAYLOAD + 2, FB_TP_PAYLOAD + 3, FB_TP_PAYLOAD + 4};
#pragma unroll
                    for (int k = 0; k < 14; k++) wsb()[(int64_t)row[k] * W] = d0[k] + tn * (tp[(int64_t)row[k] * n + i] - d0[k]);
                    fresh = true;
                } else active = false;
            }
        }
    }
}
// f_init!(kinematics::ECEF / ::NED, ic) (kinematics.jl:255-280, 336-364) applied to the WA initial condition k_trim leaves
// (ψ_nw = 0 there, so q_wb = q_nb and q_ew = ltf(n_e) exactly): rows FB_X_Q_WB.. are rewritten in the mechanisation's states.
template <int KIN>
__global__ __launch_bounds__(256) void k_kin_convert(KArgs a, const double* tp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int64_t n = a.n;
    constexpr int KX = FB_X_Q_WB;
    double k[9];
#pragma unroll
    for (int j = 0; j < 9; j++) k[j] = a.x[(int64_t)(KX + j) * n + i];
    const quat q_nb = {k[0], k[1], k[2], k[3]}, q_en = {k[4], k[5], k[6], k[7]};
    const double h_e = k[8];
    const v3 n_e = {tp[(int64_t)FB_TP_N_E * n + i], tp[(int64_t)(FB_TP_N_E + 1) * n + i], tp[(int64_t)(FB_TP_N_E + 2) * n + i]};
    double o[9] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (KIN == FB_KIN_ECEF) {
        const quat q_eb = qmul(q_en, q_nb);
        o[0] = q_eb.w; o[1] = q_eb.x; o[2] = q_eb.y; o[3] = q_eb.z; o[4] = n_e.x; o[5] = n_e.y; o[6] = n_e.z; o[7] = h_e;
    } else {
        const double q1 = q_nb.w, q2 = q_nb.x, q3 = q_nb.y, q4 = q_nb.z;   // REuler(q_nb), attitude.jl:382-391
        o[0] = atan2(2 * (q1 * q4 + q2 * q3), 1 - 2 * (q3 * q3 + q4 * q4));
        o[1] = asin(fmin(fmax(2 * (q1 * q3 - q2 * q4), -1.0), 1.0));
        o[2] = atan2(2 * (q1 * q2 + q3 * q4), 1 - 2 * (q2 * q2 + q3 * q3));
        o[3] = atan2(n_e.z, sqrt(n_e.x * n_e.x + n_e.y * n_e.y));           // LatLon(n_e), geodesy.jl:103-106
        o[4] = atan2(n_e.y, n_e.x);
        o[5] = h_e;
    }
#pragma unroll
    for (int j = 0; j < 9; j++) a.x[(int64_t)(KX + j) * n + i] = o[j];
}

}  // namespace fbd
