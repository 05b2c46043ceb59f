// c172_duo_device.hpp — the right-hand side of the airborne Cessna172Sv0 evaluation (rhs_duo<KIN, ROLE>: WA, ECEF or NED kinematics),
// cut in two for k_step_duo<KIN>
// (c172_kernels.hpp): two waves serve the same 64 aircraft, each evaluating its share, so that two waves fit a SIMD's registers.
// Same arithmetic, block by block, as rhs() in c172_device_impl.inc (which stays the reference form: every other kernel uses it,
// and tools/duo_check.py + the parity tests compare the two); only the order of the blocks and who evaluates them differ:
//
//   role P  "atmosphere + power plant"                     role D  "airframe"
//   reads its state rows (position, engine, fuel)          attitude, n_e, wind-relative velocity -> put the velocity at the propeller
//   ---- publishes R                                        ---- publishes V
//   geoid height (lat / lon, EGM96) -> h_o                 airflow angles, filter rows,
//   ISA atmosphere -> T, p, log p                          table locations and lookups on the alpha / beta axes
//   put rho, h_o                                            ---- waits for R:  fuel row, kinematics derivatives (9 rows), radii of curvature,
//   ---- publishes A, waits for V                                mass properties, gravity at the CoM
//   get the velocity at the propeller; propeller            ---- waits for A:  get rho, h_o (and examines the altitude / ISA ranges),
//   put F_p, tau_p, h_rot                                        ground-effect location, aerodynamic coefficients and wrench, landing gear
//   ---- publishes W                                             (airborne shortcut), the propeller-free part of the rigid-body dynamics
//   engine (3 rows), fuel flow                              ---- waits for W:  get F_p, tau_p, h_rot
//   ---- waits for X                                        ---- publishes X
//   engine-speed row, fuel row                              the rest of the dynamics (6 rows); the step's bookkeeping; ---- publishes T
//
// Only n_e is evaluated twice (WA: ten instructions from q_ew; ECEF: it is a state; NED: the sines and cosines of latitude and
// longitude). The position rows role P reads are q_ew, h_e (WA), n_e, h_e (ECEF), latitude, longitude, h_e (NED). The two waves of a pair keep in step through two counters in LDS — no barrier
// (k_step_duo, c172_kernels.hpp) — and every wait sits where a value is needed. Who may touch which LDS row when:
//   * role P reads the state rows it needs — of role D's: the position rows — at the head of its evaluation, ahead of its point R; role D
//     rewrites the kinematics rows behind R. Role D reads nothing of role P's rows but the fuel row, behind R too (role P emits it last
//     of all, behind role D's point X, so "R reached" means the previous evaluation's fuel row is there and this one's is not);
//   * role D rewrites the filter rows at its head, the kinematics rows behind R, the angular / linear velocity rows behind X; role P its
//     engine-compensator rows behind W and the engine-speed and fuel rows behind X;
//   * exchange rows 0-5 are the angular / linear velocity rows of the evaluation panel: only role D reads them as state (at its head),
//     so it may overwrite rows 0-2 with the velocity at the propeller (point V); role P reads that behind V and then writes F_p, tau_p
//     over rows 0-5 (point W), which role D reads behind W and, behind X, overwrites with its own emit;
//   * the control and flag words of an evaluation are role D's, written before its point T, which role P waits for at the top of its loop
//     (role D has waited for W of the evaluation before, so role P has long read the previous ones).
#pragma once
// 1: role P keeps its aircraft's EGM96 cell (indices + four samples, six registers of the ~130 it has to spare) across the evaluations of a launch
// and gathers again only when the aircraft has left the cell: geoid_height_cached (c172_device_impl.inc). 0 (shipped): gather at every evaluation.
// Measured, round 5 (profiles/r05_ab_geoid_cache.txt, same box, alternating): a fleet spread over the sphere 14.21-14.24 -> 14.01-14.02 ms per
// launch — the whole cost of dispersion (1.7 %) — but the benchmark batch, which sits in ONE cell (every gather a wave-wide broadcast that
// hits the L2), 13.96 -> 14.03 ms: the compare-and-branch per evaluation costs the pair more issue slots than the four broadcast loads did.
// The headline configuration decides; a user whose fleet covers the globe builds with -DFB_DUO_GEOID_CACHE=1.
#ifndef FB_DUO_GEOID_CACHE
#define FB_DUO_GEOID_CACHE 0
#endif
#include "c172_device.hpp"

namespace fbd {

// Diagnostic builds (-DFB_STAMP -DFB_DUO_TIMELINE, tools/duo_timeline.py): when, counted from the moment the wave leaves the barrier at the
// top of an evaluation, wave 0 (role P) and wave 4 (role D) of workgroup 0 pass the marked points of their evaluation — a two-row
// Gantt chart of one SIMD (g_stamp_acc[k] role P, [16 + k] role D; no drains: a mark is one s_memtime and one lane's global add).
#if defined(FB_STAMP) && defined(FB_DUO_TIMELINE)
__device__ unsigned long long g_duo_t0[2];
__device__ __forceinline__ void duo_mark(int role, int k) {
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler may not move work across a mark: what is attributed to an interval was issued in it)
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (k == 0) g_duo_t0[role - 1] = t;
        else if (g_duo_t0[role - 1] != 0) { g_stamp_acc[(role == 1 ? 0 : 16) + k] += t - g_duo_t0[role - 1]; g_stamp_cnt[(role == 1 ? 0 : 16) + k] += 1; }
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define DUO_MARK(role, k) duo_mark(role, k)
#else
#define DUO_MARK(role, k) do { } while (0)
#endif

// the points of an evaluation the two roles publish / wait for (k_step_duo, c172_kernels.hpp: "How the two waves of a PAIR ... keep in step")
enum { DUO_PT_T = 0, DUO_PT_V = 1, DUO_PT_X = 2,    // role D: top of the loop; velocity at the propeller put; wrench and fuel row read
       DUO_PT_R = 0, DUO_PT_A = 1, DUO_PT_W = 2,    // role P: state rows read; density and altitude put; wrench put
       DUO_NPT = 3 };
constexpr int XD_FP = 0, XD_TAUP = 3, XD_HROT = 6, XD_RHO = 7, XD_HO = 8, XD_ROWS = 9;   // exchange rows (0-5 overlaid, see above)
constexpr int XD_VP = XD_FP;   // role D -> role P before barrier A: the velocity at the propeller, in the rows that carry F_p after it
// rows of KArgs::duo_tap a tapped evaluation of the Cessna172Xv2 instance writes (the DUO_TAP_* enum of c172_kernels.hpp, which checks these)
constexpr int DUO_TAP_THETA_ROW = 0, DUO_TAP_WX_ROW = 2, DUO_TAP_VD_ROW = 5, DUO_TAP_EAS_ROW = 7, DUO_TAP_ALPHA_ROW = 8, DUO_TAP_LAT_ROW = 10;

template <int KIN, int ROLE, class In, class Emit, class XV>
__device__ __forceinline__ int32_t rhs_duo(const XV& x, int stall, int eng_state, const In& in, const Env& env, const Tables& T, const Emit& emit, StepAux& aux) {
    using namespace c172;
    static_assert(ROLE == 1 || ROLE == 2, "role P or role D");
    static_assert(KIN == FB_KIN_WA || KIN == FB_KIN_ECEF || KIN == FB_KIN_NED, "kinematic mechanisation");
    int32_t st = 0;
    constexpr int KX = FB_X_Q_WB;
    constexpr bool X = Emit::x2;   // Cessna172Xv2 (k_step_duo<KIN, true>): see the kernel's header
    auto gkp = [&](int off) -> gk_cptr { return T.gk + off; };
    auto atan2m = [&](double y, double x0) -> double { return atan2_step(y, x0, T.rk + LDS_ATAN); };
    auto atan2p = [&](double y, double x0) -> double { return atan2_step<true>(y, x0, T.rk + LDS_ATAN); };   // x0 >= 0, not both zero

    // ===== kinematics head (kinematics.jl:181-223; geodesy.jl:62-69, 140-147). Role P needs the position only (n_e from q_ew, h_e); the
    // attitude products, the wind-relative velocity and from it the velocity at the propeller are role D's, which hands the latter over =====
    // (ECEF, kinematics.jl:282-320: n_e and h_e are states; NED, :366-407: latitude, longitude and h_e are, and n_e comes from their sines
    // and cosines — both roles form it, role P for the geoid, role D for the radii of curvature and gravity. The rows role P reads of the
    // kinematic block — WA q_ew, h_e; ECEF n_e, h_e; NED lat, lon, h_e — it reads ahead of its point R, like the WA ones.)
    const double h_e = x[KIN == FB_KIN_WA ? KX + 8 : (KIN == FB_KIN_ECEF ? KX + 7 : KX + 5)];
    if (!(h_e >= H_MIN)) st |= FB_ST_ALT_RANGE;
    [[maybe_unused]] quat q_ew = {1, 0, 0, 0};
    [[maybe_unused]] double dq12 = 0, dq13 = 0, dq24 = 0, dq34 = 0;
    [[maybe_unused]] double ned_cla = 1, pos0 = 0, pos1 = 0, pos2 = 0;   // (pos: the position rows as read, for role P's pin)
    v3 n_e;
    if constexpr (KIN == FB_KIN_WA) {
        q_ew = {x[KX + 4], x[KX + 5], x[KX + 6], x[KX + 7]};
        dq12 = 2 * q_ew.w * q_ew.x; dq13 = 2 * q_ew.w * q_ew.y;
        dq24 = 2 * q_ew.x * q_ew.z; dq34 = 2 * q_ew.y * q_ew.z;
        n_e = {-(dq24 + dq13), -(dq34 - dq12), -(1 - 2 * (q_ew.x * q_ew.x + q_ew.y * q_ew.y))};
    } else if constexpr (KIN == FB_KIN_ECEF) {
        pos0 = x[KX + 4]; pos1 = x[KX + 5]; pos2 = x[KX + 6];
        n_e = {pos0, pos1, pos2};   // state n-vector, normalised only in f_step! (kinematics.jl:286, 317-320)
    } else {
        pos0 = x[KX + 3]; pos1 = x[KX + 4];
        double sla, slo, clo;
        sincos_step(pos0, sla, ned_cla); sincos_step(pos1, slo, clo);
        n_e = {ned_cla * clo, ned_cla * slo, sla};   // geodesy.jl:97-101
    }
    const v3 r_p = {prop_r[0], prop_r[1], prop_r[2]};

    if constexpr (ROLE == 1) {
        // ================================= role P =================================
        lds_cptr PT = T.lds + LDS_PISTON;
        lds_cptr RPT = T.rk + LDS_PISTON;
        const double w_eng = x[FB_X_ENG_OMEGA];
        const double x_frc = x[FB_X_ENG_FRC], x_idle = x[FB_X_ENG_IDLE], x_fuel = x[FB_X_FUEL];
        {   // every state row this role reads is read HERE (pinned: the compiler may not sink a read towards its use, past the point)
            double pin_w = w_eng, pin_f = x_frc, pin_i = x_idle, pin_u = x_fuel, pin_h = h_e;
            asm volatile("" : "+v"(pin_w), "+v"(pin_f), "+v"(pin_i), "+v"(pin_u), "+v"(pin_h));
            if constexpr (KIN == FB_KIN_WA) {
                double pin_q0 = q_ew.w, pin_q1 = q_ew.x, pin_q2 = q_ew.y, pin_q3 = q_ew.z;
                asm volatile("" : "+v"(pin_q0), "+v"(pin_q1), "+v"(pin_q2), "+v"(pin_q3));
            } else {
                double pin_p0 = pos0, pin_p1 = pos1, pin_p2 = pos2;
                asm volatile("" : "+v"(pin_p0), "+v"(pin_p1), "+v"(pin_p2));
            }
        }
        emit.xpub(DUO_PT_R);   // ----- point R: state rows read (role D may rewrite the kinematics rows; this wave's previous evaluation is complete) -----
        if constexpr (X) __builtin_amdgcn_s_setprio(0);   // (Cessna172Xv2: this role ran ahead of role D from the top of its loop to here, see k_step_duo)
        double lat, lon;
        // (the Cessna172Sv0 instances: role P has the six registers; in the Cessna172Xv2 ones, where it also carries the actuators, they spill)
        double N_geoid;
        if constexpr (FB_DUO_GEOID_CACHE && !Emit::x2) N_geoid = geoid_height_cached(T, n_e, lat, lon, *emit.gcache);
        else N_geoid = geoid_height<true>(T, n_e, lat, lon);
        const double h_o = h_e - N_geoid;
        if (!(h_o >= H_MIN)) st |= FB_ST_ALT_RANGE;
        if constexpr (X) {
            if (emit.tap) {   // kinematics.y.ϕ_λ as the guidance reads it (rhs(): the states themselves in the NED mechanisation)
                const double ll[2] = {KIN == FB_KIN_NED ? pos0 : lat, KIN == FB_KIN_NED ? pos1 : lon};
                emit.template tap_rows<2>(DUO_TAP_LAT_ROW, ll);
            }
        }
        // ----- air data (atmosphere.jl:220-242) -----
        DUO_MARK(1, 1);   // geoid done
        double T_air, p_air, lnp_air;
        const double h_gp = h_o * wgs::a / (wgs::a + h_o);   // geopotential altitude
        isa_data<true>(h_gp, env.T_sl, env.p_sl, T_air, p_air, lnp_air, st);
        const double rs_T = rsqrt(T_air), i_T = rs_T * rs_T;
        const double rho = (p_air * (1 / isa::R)) * i_T;
        constexpr double sqrt_gR = 20.046795704052055;   // sqrt(1.4 * 287.05287)
        const double i_a_snd = (1 / sqrt_gR) * rs_T;
        DUO_MARK(1, 2);   // ISA done
        emit.xput(XD_RHO, rho); emit.xput(XD_HO, h_o);
        DUO_MARK(1, 11);  // at A
        emit.xpub(DUO_PT_A);    // ----- point A: density and orthometric altitude put (role D examines the altitude and ISA ranges itself) -----
        emit.xwait(DUO_PT_V);   // ----- role D's point V: the velocity at the propeller -----
        DUO_MARK(1, 3);   // past V

        // ----- propeller (propellers.jl:405-452) -----
        const double w_prop = w_eng;  // gear ratio 1
        const v3 v_p = {emit.xget(XD_VP), emit.xget(XD_VP + 1), emit.xget(XD_VP + 2)};   // v_wb_b + w_eb_b x r_p, from role D
        const double v_J = norm(v_p);
        const double J_adv = 2 * PI * v_J / (fmax(fabs(w_prop), 1.0) * prop_d);
        const double Mt = fabs(w_prop) * (prop_d / 2) * i_a_snd;
        const loc lj = range_locate(0.0, 1.5, PR_NJ, J_adv, true);
        const loc lm = range_locate(0.0, 1.5, PR_NM, Mt, true);
        const double w00 = (1 - lj.w) * (1 - lm.w), w01 = (1 - lj.w) * lm.w, w10 = lj.w * (1 - lm.w), w11 = lj.w * lm.w;
        // (no LDS to spare for this 14 KB table: its four corner records come from the blob in global memory — hot in the vector L1 — as
        // one batch of loads)
        gk_cptr g00 = T.gk + LDS_PROP + (lj.i + PR_NJ * lm.i) * PR_NC;
        auto coef = [&](int cidx) {
            return (w00 * g00[cidx] + w01 * g00[PR_NJ * PR_NC + cidx]) + (w10 * g00[PR_NC + cidx] + w11 * g00[PR_NJ * PR_NC + PR_NC + cidx]);
        };
        const double C_Fx = coef(0), C_Mx = coef(1), C_Fz_a = coef(2), C_Mz_a = coef(3);
        double a_p = 0, b_p = 0;
        if (!(v_J < 0.1)) {
            a_p = atan2m(v_p.z, v_p.x);
            b_p = atan2p(v_p.y, sqrt(v_p.x * v_p.x + v_p.z * v_p.z));
        }
        DUO_MARK(1, 4);   // propeller coefficients and angles
        const double fr = w_prop / (2 * PI), fr2 = fr * fr;
        constexpr double d4 = prop_d * prop_d * prop_d * prop_d, d5 = d4 * prop_d;
        const double kF = rho * fr2 * d4, kM = rho * fr2 * d5;
        const v3 F_p = {kF * C_Fx, kF * (C_Fz_a * b_p), kF * (C_Fz_a * a_p)};
        const v3 tau_p = {kM * C_Mx, kM * (C_Mz_a * b_p), kM * (C_Mz_a * a_p)};  // CW: sense = +1
        const v3 tau_pb = tau_p + cross(r_p, F_p);
        emit.xput(XD_FP, F_p.x); emit.xput(XD_FP + 1, F_p.y); emit.xput(XD_FP + 2, F_p.z);
        emit.xput(XD_TAUP, tau_pb.x); emit.xput(XD_TAUP + 1, tau_pb.y); emit.xput(XD_TAUP + 2, tau_pb.z);
        DUO_MARK(1, 5);   // wrench put
        emit.xput(XD_HROT, prop_Jxx * w_prop);
        emit.xpub(DUO_PT_W);    // ----- point W: propeller wrench put -----

        // ----- engine (piston.jl:314-426) -----
        double out_frc, out_idle;
        static_assert(FB_X_ENG_FRC == FB_X_ENG_IDLE + 1, "engine regulator rows are adjacent");
        const double kfrc = pi_ode(5.0, 200.0, 0.0, -1.0, 1.0, -w_eng, x_frc, out_frc);
        const double ke2[2] = {pi_ode(4.0, 2.0, 0.0, -0.5, 0.5, 1 - w_eng / w_idle, x_idle, out_idle), kfrc};
        emit_rows<2>(emit, FB_X_ENG_IDLE, ke2);
        const double mu_ratio_idle = 0.5 + out_idle;
        const double n_eng = w_eng / w_rated;
        const double rt_arg = (0.5 * 6.5e-3 * isa::R / isa::g_std) * (env.ln_p_sl + lnp_air);
        const bool tropo = h_gp < 11000.0;
        double rt_theta = env.k_rt * (T_air * rs_T);   // (T_ISA/T_std)^1/2, see rhs()
        if (__builtin_amdgcn_ballot_w64(!tropo) != 0) { const double e = exp_step(rt_arg); rt_theta = tropo ? rt_theta : e; }
        const double delta = (p_air / isa::p_std) / rt_theta;
        const double throttle = in.get_throttle(), mixture = in.get_mixture();
        const loc l_n2 = range_locate(0.667, 1.0, 2, n_eng, false);
        const double mu_wot = lerp2(PT + PT_MU_WOT_V, 2, l_n2, range_locate(0.441, 1.0, 9, delta, false));
        const double mu = mu_wot * (mu_ratio_idle + throttle * (1 - mu_ratio_idle));
DUO_MARK(1, 6);   // engine head done
        // behind W role P has the longer way to go (the rest of the engine: ~400 instructions against role D's ~110 once it has the wrench): it
        // goes ahead of D in issue priority until its evaluation ends (profiles/r03_ab_prio.txt)
#ifndef FB_DUO_P_TAIL_PRIO
#define FB_DUO_P_TAIL_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(FB_DUO_P_TAIL_PRIO);
        const double k_f = rsqrt(rho * (1 / isa::rho_std));
        const bool mix_auto = in.ui & FB_UI_MIXTURE_AUTO;
        const double f_run = mix_auto ? f_lean + mixture * (f_rich - f_lean) : k_f * (f_rich * (0.5 * (mixture + 1)));
        const loc l_n13 = grid_locate<13, true, AUX_N13>(PT + PT_PISTD_N_K, RPT + PT_PISTD_N_K, n_eng, true, true, gkp(LDS_PISTON + PT_PISTD_N_K), T.gk);
        const loc l_n5w = grid_locate<5, true>(PT + PT_PIWOT_N_K, RPT + PT_PIWOT_N_K, n_eng, true, true, gkp(LDS_PISTON + PT_PIWOT_N_K));
        const loc l_n5s = grid_locate<5, true>(PT + PT_SFC_N_K, RPT + PT_SFC_N_K, n_eng, false, false, gkp(LDS_PISTON + PT_SFC_N_K));
        const loc l_f = grid_locate<11, true, AUX_F11>(PT + PT_F_K, RPT + PT_F_K, f_run, true, true, gkp(LDS_PISTON + PT_F_K), T.gk);
        const double pi_ratio = lerp1(PT + PT_PI_RATIO_V, l_f), sfc_ratio = lerp1(PT + PT_SFC_RATIO_V, l_f);
        const double d_wot = lerp2(PT + PT_DELTA_WOT_V, 2, l_n2, range_locate(0.401, 0.936, 9, mu, false));
        const double pi_std = lerp2(PT + PT_PISTD_V, 13, l_n13, grid_locate<3, true>(PT + PT_PISTD_MU_K, RPT + PT_PISTD_MU_K, mu, true, true, gkp(LDS_PISTON + PT_PISTD_MU_K)));
        const double pi_wot = lerp2(PT + PT_PIWOT_V, 5, l_n5w, grid_locate<3, true>(PT + PT_PIWOT_D_K, RPT + PT_PIWOT_D_K, d_wot, true, false, gkp(LDS_PISTON + PT_PIWOT_D_K)));
        double pi_isa = (fabs(d_wot - 1) < 5e-3) ? pi_std : pi_std + (pi_wot - pi_std) / (d_wot - 1) * (delta - 1);
        pi_isa = fmax(pi_isa, 0.0);
        DUO_MARK(1, 8);   // engine lookups
        const double pi_pow = pi_isa * (rt_theta * (isa_sqrt_T_std * rs_T));   // pi_isa (T_ISA / T)^1/2
        const double pi_act = pi_pow * pi_ratio;
        const double P_run = P_rated * pi_act;
        const double tau_run = (w_eng > 0) ? P_run / w_eng : 0.0;
        const double SFC_run = lerp2(PT + PT_SFC_POW_V, 5, l_n5s, grid_locate<8, true>(PT + PT_SFC_PI_K, RPT + PT_SFC_PI_K, pi_act, false, false, gkp(LDS_PISTON + PT_SFC_PI_K))) * sfc_ratio;
        const bool eng_off = eng_state == 0, eng_starting = eng_state == 1, eng_running = !(eng_off || eng_starting);
        const double tau_shaft = eng_off ? out_frc * (0.01 * P_rated / w_rated) : (eng_starting ? tau_start : tau_run);
        const double mdot = eng_running ? SFC_run * P_run : 0.0;
        const double tau_load = tau_p.x;  // gear_ratio * τ_prop
        DUO_MARK(1, 9);   // engine done
        emit.xwait(DUO_PT_X);   // ----- role D's point X: it has read the fuel row (and the wrench): the rows below may be rewritten in place -----
        emit(FB_X_ENG_OMEGA, (tau_shaft + tau_load) / (J_eng + prop_Jxx));
        // ----- fuel (c172.jl:607-616) -----
        (void)x_fuel;
        emit(FB_X_FUEL, -mdot / (m_full - m_res));
        __builtin_amdgcn_s_setprio(0);
        DUO_MARK(1, 10);   // end of role P's evaluation
        return st;
    } else {
        // ================================= role D =================================
        lds_cptr A = T.lds + LDS_AERO;
        lds_cptr RA = T.rk + LDS_AERO;
        const v3 w_eb_b = {x[FB_X_OMEGA_EB_B], x[FB_X_OMEGA_EB_B + 1], x[FB_X_OMEGA_EB_B + 2]};
        const v3 v_eb_b = {x[FB_X_V_EB_B], x[FB_X_V_EB_B + 1], x[FB_X_V_EB_B + 2]};
        [[maybe_unused]] quat q_wb = {1, 0, 0, 0}, q_nw = {1, 0, 0, 0}, q_en = {1, 0, 0, 0}, q_eb_s = {1, 0, 0, 0};
        [[maybe_unused]] double ned_s2 = 0, ned_c2 = 1, ned_s3 = 0, ned_c3 = 1;
        quat q_nb;
        if constexpr (KIN == FB_KIN_WA) {
            q_wb = {x[KX], x[KX + 1], x[KX + 2], x[KX + 3]};
            double s_nw, c_nw;
            half_angle_cs(-(dq34 + dq12), dq24 - dq13, c_nw, s_nw);
            q_nw = {c_nw, 0.0, 0.0, s_nw};
            q_nb = {c_nw * q_wb.w - s_nw * q_wb.z, c_nw * q_wb.x - s_nw * q_wb.y, c_nw * q_wb.y + s_nw * q_wb.x, c_nw * q_wb.z + s_nw * q_wb.w};
        } else if constexpr (KIN == FB_KIN_ECEF) {
            q_eb_s = {x[KX], x[KX + 1], x[KX + 2], x[KX + 3]};
            q_en = ltf_quat(n_e);
            q_nb = qmul(qconj(q_en), q_eb_s);
        } else {
            double s1, c1;   // Rz(ψ) ∘ Ry(θ) ∘ Rx(φ) (attitude.jl:393-395), the products without their terms in the factors' zeros
            sincos_step(0.5 * x[KX], s1, c1); sincos_step(0.5 * x[KX + 1], ned_s2, ned_c2); sincos_step(0.5 * x[KX + 2], ned_s3, ned_c3);
            const quat zy = {c1 * ned_c2, -(s1 * ned_s2), c1 * ned_s2, ned_c2 * s1};
            q_nb = {zy.w * ned_c3 - zy.x * ned_s3, zy.w * ned_s3 + ned_c3 * zy.x, ned_c3 * zy.y + zy.z * ned_s3, ned_c3 * zy.z - zy.y * ned_s3};
            q_en = ltf_quat(n_e);
        }
        // wind-relative velocity (atmosphere.jl:269-283)
        const v3 v_ew_n = {env.wind_n, env.wind_e, env.wind_d};
        const v3 v_ew_b = qrot_inv(q_nb, v_ew_n);
        const v3 v_wb_b = v_eb_b - v_ew_b;
        {   // ... and at the propeller, for role P (exchange rows 0-2: this role's own velocity rows of the evaluation panel, read just above)
            const v3 v_p = v_wb_b + cross(w_eb_b, r_p);
            emit.xput(XD_VP, v_p.x); emit.xput(XD_VP + 1, v_p.y); emit.xput(XD_VP + 2, v_p.z);
        }
        emit.xpub(DUO_PT_V);   // ----- point V: velocity at the propeller put -----
        DUO_MARK(2, 1);   // head, velocity at the propeller put
        quat q_eb;
        if constexpr (KIN == FB_KIN_WA) q_eb = qmul(q_ew, q_wb);
        else if constexpr (KIN == FB_KIN_ECEF) q_eb = q_eb_s;
        else q_eb = qmul(q_en, q_nb);
        const double TAS = norm(v_wb_b);
        // ----- aerodynamics, the part that needs no atmosphere: airflow angles, filters, table locations (c172.jl:307-340) -----
        AeroC ac;
        [[maybe_unused]] double x2_de = 0, x2_da = 0, x2_dr = 0;
        if constexpr (!X) in.fetch_aero(ac);   // (launch constants; Cessna172Xv2: this evaluation's, behind role P's point R below)
        double alpha = 0, beta = 0, cos_al = 1, sin_al = 0;
        if (TAS > 0.1) {  // also covers get_airflow_angles' own ‖v‖ < 0.1 guard (atmosphere.jl:329-337)
            const double r2 = v_wb_b.x * v_wb_b.x + v_wb_b.z * v_wb_b.z;
            const bool r_ok = r2 > 0;
            const double ir = rsqrt(r_ok ? r2 : 1.0), r = r2 * ir;
            alpha = atan2m(v_wb_b.z, v_wb_b.x);
            beta = atan2p(v_wb_b.y, r);   // (TAS > 0.1: r and v_y are not both zero)
            if (r_ok) { cos_al = v_wb_b.x * ir; sin_al = v_wb_b.z * ir; }   // cos, sin of atan2(z, x) (atan2(0, 0) = 0)
        }
        DUO_MARK(2, 2);   // airflow angles
        if constexpr (X) { if (emit.tap) { const double ab[2] = {alpha, beta}; emit.template tap_rows<2>(DUO_TAP_ALPHA_ROW, ab); } }
        const double V = fmax(TAS, V_min);
        const double afd = 1 / tau_filt * (alpha - x[FB_X_ALPHA_FILT]);
        const double bfd = 1 / tau_filt * (beta - x[FB_X_BETA_FILT]);
        const double kf2[2] = {afd, bfd};
        emit_rows<2>(emit, FB_X_ALPHA_FILT, kf2);
        const double i2V = 1 / (2 * V);
        const double ad_nd = clampd(afd * c * i2V, -0.04, 0.04);
        const double al = clampd(alpha, -0.1, 0.36), be = clampd(beta, -0.2, 0.2);
        if constexpr (X) emit.xwait(DUO_PT_R);   // Cessna172Xv2: role P has put this evaluation's deflection-only sums into the panel ahead of its point R
        const loc l_al26 = grid_locate<26, true, AUX_AL26>(A + AT_CD_ALPHA_K, RA + AT_CD_ALPHA_K, al, true, true, gkp(LDS_AERO + AT_CD_ALPHA_K), T.gk);
        const loc l_al17 = grid_locate<17, true, AUX_AL17>(A + AT_CL_ALPHA_K, RA + AT_CL_ALPHA_K, al, true, true, gkp(LDS_AERO + AT_CL_ALPHA_K), T.gk);
        const loc l_al2 = grid_locate<2, true>(A + AT_ALPHA2_K, RA + AT_ALPHA2_K, al, true, true, gkp(LDS_AERO + AT_ALPHA2_K));
        const loc l_be3 = grid_locate<3, true>(A + AT_CY_BETA_K, RA + AT_CY_BETA_K, be, true, true, gkp(LDS_AERO + AT_CY_BETA_K));
        const loc l_bu = grid_locate<3, true>(A + AT_UNIT3_K, RA + AT_UNIT3_K, be, true, true, gkp(LDS_AERO + AT_UNIT3_K));
        DUO_MARK(2, 3);   // knot locations
        if constexpr (X) {
            // the evaluation's sums from the panel; the four that are linear in the deflections formed here (InputsAgg::sum_aero's expressions)
            lds_cptr sp = in.pld_l;
            x2_de = sp[4 * 256]; x2_da = sp[5 * 256]; x2_dr = sp[6 * 256];   // (the four linear sums are formed where the scalar derivatives are loaded, behind A)
            const uint64_t iw = __builtin_bit_cast(uint64_t, sp[9 * 256]);
            ac.cd_in = sp[0]; ac.cd_df = sp[1 * 256]; ac.cl_df = sp[2 * 256]; ac.cm_in = sp[3 * 256];
            ac.l_df4 = {(int)(uint32_t)iw, sp[7 * 256]}; ac.l_df2 = {(int)(uint32_t)(iw >> 32), sp[8 * 256]};
        }
        const loc l_stall = {0, stall ? 1.0 : 0.0};
        const loc l_df4 = ac.l_df4, l_df2 = ac.l_df2;
        // the lookups on those axes alone
        const double cd_al = lerp2(A + AT_CD_ALPHA_DF_V, 26, l_al26, l_df4) + ac.cd_df, cd_be = lerp1(A + AT_CD_BETA_V, l_bu);
        const double cy_be = lerp2(A + AT_CY_BETA_DF_V, 3, l_be3, l_df2);
        const double cy_p = lerp2(A + AT_CY_P_V, 2, l_al2, l_df2), cy_r = lerp2(A + AT_CY_R_V, 2, l_al2, l_df2);
        const double cl_al = lerp2(A + AT_CL_ALPHA_V, 17, l_al17, l_stall) + ac.cl_df;
        const double cl_r = lerp2(A + AT_CL_R_V, 2, l_al2, l_df2);
        DUO_MARK(2, 4);   // lookups done
        if constexpr (!X) emit.xwait(DUO_PT_R);   // ----- role P's point R: it has read q_ew, h_e (the kinematics rows may be rewritten) and finished its previous
                                // evaluation (the fuel row, which it emits last of all, is there; it will not emit it again before this wave's X) -----
        const double x_fuel = x[FB_X_FUEL];
        if constexpr (X) in.fetch_pld_raw(const_cast<double(&)[10]>(in.pldv));   // (the payload's sums: in flight behind the kinematics block, consumed by the mass properties)
        DUO_MARK(2, 5);   // past R

        // ----- kinematics derivatives (kinematics.jl:181-242; geodesy.jl:125-129) -----
        // ----- radii of curvature (geodesy.jl:125-129), fuel mass, mass properties, gravity at the CoM: while role P works on the atmosphere -----
        const double i_fden = rsqrt(1 - wgs::e2 * n_e.z * n_e.z);   // 1 / sqrt(1 - e^2 sin^2 lat)
        const double R_E = wgs::a * i_fden;
        const double R_N = (wgs::a * (1 - wgs::e2)) * (i_fden * i_fden * i_fden);
        const double RE_h = R_E + h_e, RN_h = R_N + h_e, i_REN = 1 / (RE_h * RN_h);   // both reciprocals from one division
        const double i_RE = RN_h * i_REN, i_RN = RE_h * i_REN;
        const v3 v_eb_n = qrot(q_nb, v_eb_b);
        const v3 w_ew_n = {v_eb_n.y * i_RE, -v_eb_n.x * i_RN, 0.0};
        v3 w_wb_b;
        if constexpr (X) {
            if (emit.tap) {
                // kinematics.y as the control laws read it (rhs(), WITH_Y; attitude.jl:382-391): theta, phi, the ground track angle, the NED velocity's
                // down component — ahead of the emits below, which rewrite the rows the NED mechanisation's angles are read from
                double tp[2];
                if constexpr (KIN == FB_KIN_NED) { tp[0] = x[KX + 1]; tp[1] = x[KX + 2]; }
                else {
                    const double q1 = q_nb.w, q2 = q_nb.x, q3 = q_nb.y, q4 = q_nb.z;
                    tp[0] = asin(fmin(fmax(2 * (q1 * q3 - q2 * q4), -1.0), 1.0));
                    tp[1] = atan2m(2 * (q1 * q2 + q3 * q4), 1 - 2 * (q2 * q2 + q3 * q3));
                }
                const bool chi_ok = KIN == FB_KIN_NED || norm(v_eb_n) > 0.1;   // the NED mechanisation has no low-speed guard (kinematics.jl:395-396)
                const double vc[2] = {v_eb_n.z, chi_ok ? atan2m(v_eb_n.y, v_eb_n.x) : 0.0};
                emit.template tap_rows<2>(DUO_TAP_THETA_ROW, tp);
                emit.template tap_rows<2>(DUO_TAP_VD_ROW, vc);
            }
        }
        if constexpr (KIN == FB_KIN_ECEF) {   // kinematics.jl:282-320
            w_wb_b = w_eb_b - qrot_inv(q_nb, w_ew_n);
            const quat a = qmul(q_eb, quat{0.0, w_eb_b.x, w_eb_b.y, w_eb_b.z});   // Attitude.dt(q_eb, ω_eb_b)
            const v3 nd = qrot(q_en, cross(w_ew_n, v3{0.0, 0.0, -1.0}));           // :309
            const double kq1[4] = {0.5 * a.w, 0.5 * a.x, 0.5 * a.y, 0.5 * a.z};
            const double kq2[5] = {nd.x, nd.y, nd.z, -v_eb_n.z, 0.0};
            emit_rows<4>(emit, KX, kq1);
            emit_rows<5>(emit, KX + 4, kq2);
        } else if constexpr (KIN == FB_KIN_NED) {   // kinematics.jl:366-425, in the forms of rhs() (sin / cos of θ, φ from the half-angle pairs)
            w_wb_b = w_eb_b - qrot_inv(q_nb, w_ew_n);
            const double tan_lat = n_e.z * rsqrt(n_e.x * n_e.x + n_e.y * n_e.y);
            const double sph = 2 * (ned_s3 * ned_c3), cph = ned_c3 * ned_c3 - ned_s3 * ned_s3;
            const double sth = 2 * (ned_s2 * ned_c2), cth = ned_c2 * ned_c2 - ned_s2 * ned_s2;
            const double sec = 1.0 / cth, tth = sth * sec, i_cla = 1.0 / ned_cla;
            const v3 w_en_n = {w_ew_n.x, w_ew_n.y, -v_eb_n.y * tan_lat / (R_E + h_e)};
            const v3 w_nb_b = w_eb_b - qrot_inv(q_nb, w_en_n);
            const double kq1[4] = {sph * sec * w_nb_b.y + cph * sec * w_nb_b.z, cph * w_nb_b.y - sph * w_nb_b.z,
                                   w_nb_b.x + sph * tth * w_nb_b.y + cph * tth * w_nb_b.z, -w_en_n.y};
            const double kq2[5] = {w_en_n.x * i_cla, -v_eb_n.z, 0.0, 0.0, 0.0};
            emit_rows<4>(emit, KX, kq1);
            emit_rows<5>(emit, KX + 4, kq2);
        } else {
        const double cpsi = q_nw.w * q_nw.w - q_nw.z * q_nw.z, spsi = 2 * (q_nw.w * q_nw.z);
        const v3 w_ew_w = {cpsi * w_ew_n.x + spsi * w_ew_n.y, cpsi * w_ew_n.y - spsi * w_ew_n.x, 0.0};
        const v3 w_ew_b = qrot_inv(q_wb, w_ew_w);
        w_wb_b = w_eb_b - w_ew_b;
            const quat a = {-(q_wb.x * w_wb_b.x + q_wb.y * w_wb_b.y + q_wb.z * w_wb_b.z),
                            q_wb.w * w_wb_b.x + (q_wb.y * w_wb_b.z - q_wb.z * w_wb_b.y),
                            q_wb.w * w_wb_b.y + (q_wb.z * w_wb_b.x - q_wb.x * w_wb_b.z),
                            q_wb.w * w_wb_b.z + (q_wb.x * w_wb_b.y - q_wb.y * w_wb_b.x)};
            const quat b2 = {-(q_ew.x * w_ew_w.x + q_ew.y * w_ew_w.y), q_ew.w * w_ew_w.x - q_ew.z * w_ew_w.y,
                             q_ew.w * w_ew_w.y + q_ew.z * w_ew_w.x, q_ew.x * w_ew_w.y - q_ew.y * w_ew_w.x};
            const double kq1[4] = {0.5 * a.w, 0.5 * a.x, 0.5 * a.y, 0.5 * a.z};
            const double kq2[5] = {0.5 * b2.w, 0.5 * b2.x, 0.5 * b2.y, 0.5 * b2.z, -v_eb_n.z};
            emit_rows<4>(emit, KX, kq1);
            emit_rows<5>(emit, KX + 4, kq2);
        }
        if constexpr (X) { if (emit.tap) { const double w3[3] = {w_wb_b.x, w_wb_b.y, w_wb_b.z}; emit.template tap_rows<3>(DUO_TAP_WX_ROW, w3); } }
        DUO_MARK(2, 6);   // kinematics rows emitted
        // ----- fuel mass, mass properties, gravity at the CoM -----
        const double m_fuel_total = m_res + x_fuel * (m_full - m_res);
        aux.m_avail = m_fuel_total - m_res;
        double M, J[6], Jc[6], iM;
        v3 r_bc;
        mass_props(m_fuel_total, in, M, J, iM, r_bc, Jc);
        const double Jxx = Jc[0], Jyy = Jc[1], Jzz = Jc[2], Jxy = Jc[3], Jxz = Jc[4], Jyz = Jc[5];
        v3 d_e;
        const v3 g_c_c = gravity_com(q_eb, r_bc, n_e, h_e, R_N, R_E, i_RN, i_RE, st, d_e);
        const v3 w_ie_b = earth_rate_b(q_eb);

        DUO_MARK(2, 7);   // mass properties, gravity
        // ----- aerodynamics, the rest (c172.jl:341-373, 226-245) -----
        emit.xwait(DUO_PT_A);   // ----- role P's point A: density, orthometric altitude -----
        const double rho = emit.xget(XD_RHO), h_o = emit.xget(XD_HO);
        // (the range checks role P's own evaluation makes on these — Altitude{Orthometric}, kinematics.jl:199; ISAData, atmosphere.jl:116-135 —
        // are made here: a status bit in this kernel is a hand-over flag, and this role keeps the book)
        if (!(h_o >= H_MIN)) st |= FB_ST_ALT_RANGE;
        if (!(h_o * wgs::a < 84852.0 * (wgs::a + h_o))) st |= FB_ST_ISA_RANGE;   // geopotential altitude h a / (a + h) beyond the last ISA layer
        const double q_dyn = 0.5 * rho * (TAS * TAS);
        if constexpr (X) { if (emit.tap) { const double e1[1] = {TAS * sqrt(rho / isa::rho_std)}; emit.template tap_rows<1>(DUO_TAP_EAS_ROW, e1); } }   // air.y.EAS (rhs(), atmosphere.jl:239)
        const double p_nd = w_wb_b.x * b * i2V, q_nd = w_wb_b.y * c * i2V, r_nd = w_wb_b.z * b * i2V;
        const double dh_nd = (h_o - env.h_trn) / b;
        const loc l_ge = grid_locate<13, true, AUX_GE>(A + AT_GE_K, RA + AT_GE_K, dh_nd, true, true, gkp(LDS_AERO + AT_GE_K), T.gk);
        auto S_ = [&](int k) -> double { return FB_SCALAR_DERIVS ? T.gk[LDS_AERO + AT_SCALARS + k] : A[AT_SCALARS + k]; };
        if constexpr (X) {   // InputsAgg::sum_aero's expressions for the sums that are linear in the deflections
            ac.cy_in = S_(AS_CY_DR) * x2_dr + S_(AS_CY_DA) * x2_da;
            ac.cl_in = S_(AS_CL_DE) * x2_de;
            ac.croll_in = S_(AS_Cl_DA) * x2_da + S_(AS_Cl_DR) * x2_dr;
            ac.cn_in = S_(AS_CN_DR) * x2_dr + S_(AS_CN_DA) * x2_da;
        }
        const double C_D = ac.cd_in + lerp1(A + AT_CD_GE_V, l_ge) * cd_al + cd_be;
        const double C_Y = ac.cy_in + cy_be + cy_p * p_nd + cy_r * r_nd;
        const double C_L = lerp1(A + AT_CL_GE_V, l_ge) * cl_al + ac.cl_in + S_(AS_CL_Q) * q_nd + S_(AS_CL_ALPHA_DOT) * ad_nd;
        const double C_l = ac.croll_in + S_(AS_Cl_BETA) * be + S_(AS_Cl_P) * p_nd + cl_r * r_nd;
        const double C_m = ac.cm_in + S_(AS_CM_ALPHA) * al + S_(AS_CM_Q) * q_nd + S_(AS_CM_ALPHA_DOT) * ad_nd;
        const double C_n = ac.cn_in + S_(AS_CN_BETA) * be + S_(AS_CN_P) * p_nd + S_(AS_CN_R) * r_nd;
        // stability -> body axes: rotation by Ry(-α) with the UNCLAMPED α (c172.jl:356-359; atmosphere.jl:353-356)
        const double qS = q_dyn * S;
        const v3 F_s = {qS * -C_D, qS * C_Y, qS * -C_L};
        const v3 F_a = {cos_al * F_s.x - sin_al * F_s.z, F_s.y, sin_al * F_s.x + cos_al * F_s.z};
        const v3 tau_a = {qS * (C_l * b), qS * (C_m * c), qS * (C_n * b)};
        aux.alpha = alpha;
        // ----- landing gear: the high-clearance shortcut of rhs() (no wheel can touch within 10 m of clearance) -----
        aux.wow = 0;
        aux.crash = 0;
        if (!(fmin(h_o - env.h_trn, h_e - H_MIN) > 10.0)) st |= FB_ST_INTERNAL_REDO;   // (within reach of the ground, or of the altitude floor: rhs(), "high-clearance shortcut")
        // ----- rigid-body dynamics at the CoM (dynamics.jl:443-525), the part that does not need the propeller: role D reaches barrier B ahead
        // of role P (tools/duo_waitprof.py), and everything behind B is on the critical path of the evaluation — so the wrench and the
        // angular momentum enter a form prepared in front of it:
        //   ω̇ = J⁻¹ (τ_a + τ_p − r_bc × (F_a + F_p) − J (ω_ie × ω) − ω_ic × (J ω_ic + h_rot))  =  J⁻¹ (T0 + τ_p − r_bc × F_p − ω_ic × h_rot)
        //   v̇_c = (F_a + F_p) / M + g − (ω + 2 ω_ie) × v_ec                                   =  A0 + F_p / M
        // (the same terms as rhs(), summed in another order: agreement with k_step_air to rounding, tests/test_gpu_duo.py)
        auto Jmul = [&](v3 v) { return v3{Jxx * v.x + Jxy * v.y + Jxz * v.z, Jxy * v.x + Jyy * v.y + Jyz * v.z, Jxz * v.x + Jyz * v.y + Jzz * v.z}; };
        const v3 v_ec_c = v_eb_b + cross(w_eb_b, r_bc);
        const v3 w_ic_c = w_ie_b + w_eb_b;
        v3 T0 = tau_a - cross(r_bc, F_a) - Jmul(cross(w_ie_b, w_eb_b)) - cross(w_ic_c, Jmul(w_ic_c));
        v3 A0 = iM * F_a + g_c_c - cross(w_eb_b + 2.0 * w_ie_b, v_ec_c);
        // symmetric 3x3 inverse by cofactors, scaled once
        const double c11 = Jyy * Jzz - Jyz * Jyz, c12 = Jyz * Jxz - Jxy * Jzz, c13 = Jxy * Jyz - Jyy * Jxz;
        const double c22 = Jxx * Jzz - Jxz * Jxz, c23 = Jxy * Jxz - Jxx * Jyz, c33 = Jxx * Jyy - Jxy * Jxy;
        const double idet = 1 / (Jxx * c11 + Jxy * c12 + Jxz * c13);
        double i11 = c11 * idet, i12 = c12 * idet, i13 = c13 * idet, i22 = c22 * idet, i23 = c23 * idet, i33 = c33 * idet;
        double wy = w_ic_c.y, wz = w_ic_c.z;
        // (pinned: without it the compiler is free to sink this arithmetic behind the barrier, next to its first use)
        asm volatile("" : "+v"(T0.x), "+v"(T0.y), "+v"(T0.z), "+v"(A0.x), "+v"(A0.y), "+v"(A0.z));
        asm volatile("" : "+v"(i11), "+v"(i12), "+v"(i13), "+v"(i22), "+v"(i23), "+v"(i33), "+v"(wy), "+v"(wz));
        DUO_MARK(2, 8);   // aerodynamics and the propeller-free part of the dynamics done
        emit.xwait(DUO_PT_W);   // ----- role P's point W: the propeller's wrench -----
        DUO_MARK(2, 9);   // past W
        const v3 F_p = {emit.xget(XD_FP), emit.xget(XD_FP + 1), emit.xget(XD_FP + 2)};
        const v3 tau_pb = {emit.xget(XD_TAUP), emit.xget(XD_TAUP + 1), emit.xget(XD_TAUP + 2)};
        const double h_rot = emit.xget(XD_HROT);   // (h_rot, 0, 0)
        {   // (pinned: the reads above are complete at the point below)
            double p0 = F_p.x, p1 = F_p.y, p2 = F_p.z, p3 = tau_pb.x, p4 = tau_pb.y, p5 = tau_pb.z, p6 = h_rot;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6));
        }
        emit.xpub(DUO_PT_X);   // ----- point X: wrench and fuel row read (role P may emit its last rows; the exchange rows are this wave's again) -----
        const v3 rw = (T0 + tau_pb) - cross(r_bc, F_p) - v3{0.0, wz * h_rot, -(wy * h_rot)};
        const v3 wd = {i11 * rw.x + i12 * rw.y + i13 * rw.z, i12 * rw.x + i22 * rw.y + i23 * rw.z, i13 * rw.x + i23 * rw.y + i33 * rw.z};
        const v3 vd_b = (A0 + iM * F_p) - cross(wd, r_bc);
        DUO_MARK(2, 10);   // dynamics
        const double kd6[6] = {wd.x, wd.y, wd.z, vd_b.x, vd_b.y, vd_b.z};
        emit_rows<6>(emit, FB_X_OMEGA_EB_B, kd6);
        DUO_MARK(2, 11);   // end of role D's evaluation
        return st;
    }
}

}  // namespace fbd
