// c172_device.hpp — device-side physics of one Cessna172Sv0 in SimpleWorld, one aircraft per lane.
//
// Written for gfx950 (MI355X): fp64 VALU work, one thread = one aircraft, all per-aircraft state in
// registers, the small lookup tables (aero / piston / propeller, 25 KB) staged in LDS per workgroup,
// the EGM96 geoid (4.2 MB float32) gathered from global memory (L2 / Infinity-Cache resident).
//
// What it computes is the reference's f_ode!(world) / f_step!(world); how it computes it is laid out
// for the GPU (file:line = reference, relative to lib/):
//   * identity frame rotations of the C172 (every t_b* is a pure translation, FlightApps/src/c172/
//     c172.jl:32,455-471,514-518,628-629; c172s.jl:30) are elided — exact, x∘1 = x in floating point;
//   * half-angle sin/cos pairs are produced by one sincos();
//   * every rotation that involves a STATE quaternion keeps the reference's composition order: the
//     states q_wb, q_ew are only renormalised in f_step! (kinematics.jl:226-229), so at RK stages they
//     are off unit norm by ~1e-8, and the reference's rotate formula is not associative for non-unit
//     quaternions (measured: re-associating the gravity rotation moves v̇ by 5e-7 m/s²);
//   * ECEF->geodetic (Fukushima) is evaluated once per point, not once per consumer
//     (dynamics.jl:476,487 call it twice on the same Oc);
//   * total mass properties are accumulated as Σm, Σm r, ΣJ with one division instead of the
//     reference's chain of pairwise `+` (dynamics.jl:262-272), which divides at every addition.
// These change results at rounding level only (tests hold GPU vs CPU oracle to 1e-9 relative on ẋ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "tables.h"
#include "../../include/flightbatch.h"

namespace fbd {

#define FBD __device__ __forceinline__
// Scheduling fence between the phases of one RHS: keeps the machine scheduler from interleaving the
// phases for ILP, which at one wave per SIMD costs more in spills than it gains.
#ifndef FB_PHASE_FENCE
#define FB_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef FB_GROUND_ATTR
#define FB_GROUND_ATTR __noinline__
#endif

constexpr double PI = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------
// small vector / quaternion algebra (FlightPhysics/src/quaternions.jl:109-115; attitude.jl:93-118)
struct v3 { double x, y, z; };
FBD v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
FBD v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
FBD v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
FBD v3 operator*(double s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
FBD double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FBD v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
FBD double norm(v3 a) { return sqrt(dot(a, a)); }

struct quat { double w, x, y, z; };
FBD quat qmul(quat a, quat b) {
    return {a.w * b.w - (a.x * b.x + a.y * b.y + a.z * b.z),
            a.w * b.x + b.w * a.x + (a.y * b.z - a.z * b.y),
            a.w * b.y + b.w * a.y + (a.z * b.x - a.x * b.z),
            a.w * b.z + b.w * a.z + (a.x * b.y - a.y * b.x)};
}
FBD quat qconj(quat q) { return {q.w, -q.x, -q.y, -q.z}; }
// v' = v + 2 q_im x (q_re v + q_im x v)   (attitude.jl:98-103)
FBD v3 qrot(quat q, v3 v) {
    const v3 qi = {q.x, q.y, q.z};
    const v3 t = q.w * v + cross(qi, v);
    return v + cross(2.0 * qi, t);
}
FBD v3 qrot_inv(quat q, v3 v) { return qrot(qconj(q), v); }

// ---------------------------------------------------------------------------------------------
// WGS-84 (FlightPhysics/src/geodesy.jl:15-35)
namespace wgs {
constexpr double GM = 3.986005e+14;
constexpr double a = 6378137.0;
constexpr double f = 1 / 298.257223563;
constexpr double w_ie = 7.292115e-05;
constexpr double b = a * (1 - f);
constexpr double e2 = 2 * f - f * f;
constexpr double a2 = a * a;
constexpr double m = w_ie * w_ie * a * a * b / GM;
constexpr double g_a = 9.7803253359;
constexpr double g_b = 9.8321849378;
constexpr double k_g = b * g_b / (a * g_a) - 1;
}  // namespace wgs
constexpr double H_MIN = -1000.0;

// LDS pointers carry their address space: a generic `double*` into LDS compiles to flat_load/flat_store, whose
// LDS aperture only reaches the first 64 KB of a workgroup's allocation on gfx950 (measured: panels above that
// read back garbage through generic pointers) and which is slower than ds_read/ds_write anyway.
typedef __attribute__((address_space(3))) const double* lds_cptr;
typedef __attribute__((address_space(3))) double* lds_ptr;
struct Tables {
    lds_cptr lds;        // [aero | piston | propeller] blob in LDS
    const float* egm96;  // 721 x 1441 float32, column-major [lat, lon], global memory
    lds_cptr rk;         // rk[j] = 1 / (lds[j+1] - lds[j]) over the aero|piston part: reciprocal knot spacings
};
constexpr int LDS_RK_DOUBLES = AT_SIZE + PT_SIZE;

// cos and sin of atan2(y, x) / 2 without evaluating the angle: half-angle identities arranged so that the
// square root is always taken of a number >= 1/2 (no cancellation). Used for the z- and y-rotation quaternions
// of the local-level frames, where only cos(angle/2), sin(angle/2) are needed (attitude.jl:288-308).
FBD void half_angle_cs(double y, double x, double& c, double& s) {
    const double r2 = x * x + y * y;
    if (!(r2 > 0)) { c = 1.0; s = 0.0; return; }   // atan2(0, 0) = 0
    const double ir = rsqrt(r2);
    const double cx = x * ir, sy = y * ir;
    if (x >= 0) {
        c = sqrt(0.5 * (1 + cx));
        s = sy / (2 * c);
    } else {
        s = copysign(sqrt(0.5 * (1 - cx)), y);
        c = sy / (2 * s);
    }
}

// EGM96 geoid height at a location given by its n-vector (geodesy.jl:103-106, 186-211).
// Bilinear on the uniform grid lat ∈ [-π/2, π/2] (721), lon ∈ [0, 2π] (1441), linear extrapolation.
FBD double geoid_height(const Tables& T, v3 n, double& lat, double& lon) {
    lat = atan2(n.z, sqrt(n.x * n.x + n.y * n.y));
    lon = atan2(n.y, n.x);
    // λ = mod(lon + 2π, 2π) (geodesy.jl:209) without calling fmod: lon ∈ [-π, π], so t = fl(lon + 2π) ∈ [π, 3π] and the
    // remainder is t itself or t - 2π, which is exact (Sterbenz). Branch-free on purpose: library fmod brings divergent
    // control flow into the stepping kernel, and with it the spill-placement bug tools/check_isa_spills.py guards against.
    const double t2 = lon + 2 * PI;
    const double lam = t2 >= 2 * PI ? t2 - 2 * PI : t2;
    const double xi = (lat + PI / 2) * (720 / PI);
    const double xj = lam * (1440 / (2 * PI));
    const int i = min(max((int)floor(xi), 0), 719);
    const int j = min(max((int)floor(xj), 0), 1439);
    const double wi = xi - i, wj = xj - j;
    const float* p = T.egm96 + i + 721 * j;
    const double a00 = p[0], a10 = p[1], a01 = p[721], a11 = p[722];
    return (1 - wi) * ((1 - wj) * a00 + wj * a01) + wi * ((1 - wj) * a10 + wj * a11);
}
FBD double geoid_height(const Tables& T, v3 n) {
    double la, lo;
    return geoid_height(T, n, la, lo);
}

// ECEF -> (n-vector, ellipsoidal altitude): Fukushima, one Halley step (geodesy.jl:367-412)
FBD void geodetic_from_ecef(v3 r, v3& n, double& h) {
    using namespace wgs;
    const double p2 = r.x * r.x + r.y * r.y;
    const double p = sqrt(p2);
    const double az = fabs(r.z);
    constexpr double c = a * e2;
    constexpr double ec2 = 1 - e2;
    const double ec = sqrt(ec2);
    const double zc = ec * az;
    const double s0 = az;
    const double c0 = ec * p;
    const double a0 = sqrt(s0 * s0 + c0 * c0);
    const double a03 = a0 * a0 * a0;
    const double b0 = 1.5 * c * s0 * c0 * ((p * s0 - zc * c0) * a0 - c * s0 * c0);
    const double s1 = (zc * a03 + c * (s0 * s0 * s0)) * a03 - b0 * s0;
    const double c1 = (p * a03 - c * (c0 * c0 * c0)) * a03 - b0 * c0;
    const double cc = ec * c1;
    const double s12 = s1 * s1, cc2 = cc * cc;
    h = (p * cc + s0 * s1 - a * sqrt(ec2 * s12 + cc2)) / sqrt(s12 + cc2);
    const double sgn = (r.z > 0) ? 1.0 : ((r.z < 0) ? -1.0 : 0.0);
    double cos_phi, sin_phi;
    if (s1 < cc) {
        const double t = s1 / cc;
        cos_phi = 1 / sqrt(1 + t * t);
        sin_phi = t * cos_phi * sgn;
    } else {
        const double t = cc / s1;
        const double as = 1 / sqrt(1 + t * t);
        cos_phi = t * as;
        sin_phi = as * sgn;
    }
    const double cl = p > 0 ? r.x / p : 1.0;
    const double sl = p > 0 ? r.y / p : 0.0;
    v3 u = {cos_phi * cl, cos_phi * sl, sin_phi};
    const double inv = 1 / norm(u);  // NVector constructor normalises (geodesy.jl:47-51)
    n = inv * u;
}
// Somigliana gravity with altitude correction (geodesy.jl:451-467)
FBD double normal_gravity(double nz, double h) {
    using namespace wgs;
    const double s2 = nz * nz;
    const double g0 = g_a * (1 + k_g * s2) / sqrt(1 - e2 * s2);
    return g0 * (1 - 2 / a * (1 + f + m - 2 * f * s2) * h + 3 / a2 * (h * h));
}

// ltf(n_e) = Rz(λ) ∘ Ry(-(ϕ + π/2)) with ψ_nw = 0 (geodesy.jl:132-135), λ = atan2(n_y, n_x), -(ϕ + π/2) = atan2(-p, -n_z),
// p = |(n_x, n_y)|: built from half-angle cos/sin pairs, no atan2 / sincos
FBD quat ltf_quat(v3 n) {
    double sl, cl, sp, cp;
    half_angle_cs(n.y, n.x, cl, sl);
    half_angle_cs(-sqrt(n.x * n.x + n.y * n.y), -n.z, cp, sp);
    return {cl * cp, -(sl * sp), cl * sp, sl * cp};
}

// ---------------------------------------------------------------------------------------------
// table lookups (Interpolations.jl semantics: Gridded(Linear) knot search = searchsortedlast clamped)
struct loc { int i; double w; };
// k: knots in LDS; rk: reciprocal spacings 1/(k[j+1]-k[j]) in LDS (computed once per workgroup), so the
// interpolation weight costs a multiply instead of an fp64 division (~12 VALU instructions each, ~20 per RHS).
template <int N>
FBD loc grid_locate(lds_cptr k, lds_cptr rk, double x, bool flat_lo, bool flat_hi) {
    x = (flat_lo && x < k[0]) ? k[0] : x;
    x = (flat_hi && x > k[N - 1]) ? k[N - 1] : x;
    int i = 0;
#pragma unroll
    for (int j = 1; j <= N - 2; j++) i += (k[j] <= x) ? 1 : 0;  // knots ascending: count = index of last knot <= x
    return {i, (x - k[i]) * rk[i]};
}
// uniform knots a + j (b-a)/(n-1): a, b, n are literals at every call site, so the reciprocal step folds at compile time
FBD loc range_locate(double a, double b, int n, double x, bool flat) {
    if (flat) x = fmin(fmax(x, a), b);
    const double xi = (x - a) * ((n - 1) / (b - a));
    const int i = min(max((int)floor(xi), 0), n - 2);
    return {i, xi - i};
}
FBD double lerp1(lds_cptr v, loc l) { return (1 - l.w) * v[l.i] + l.w * v[l.i + 1]; }
FBD double lerp2(lds_cptr v, int n1, loc l1, loc l2) {
    lds_cptr p = v + l1.i + n1 * l2.i;
    return (1 - l1.w) * ((1 - l2.w) * p[0] + l2.w * p[n1]) + l1.w * ((1 - l2.w) * p[1] + l2.w * p[n1 + 1]);
}

// ---------------------------------------------------------------------------------------------
// Atmosphere (FlightPhysics/src/atmosphere.jl:22-34, 99-135)
namespace isa {
constexpr double R = 287.05287, gamma = 1.40, beta_s = 1.458e-6, S = 110.4;
constexpr double T_std = 288.15, p_std = 101325.0, rho_std = p_std / (R * T_std), g_std = 9.80665;
}  // namespace isa
FBD void isa_data(double h, double T_sl, double p_sl, double& T, double& p, int32_t& st) {
    constexpr double beta[7] = {-6.5e-3, 0, 1e-3, 2.8e-3, 0, -2.8e-3, -2e-3};
    constexpr double hc[7] = {11000, 20000, 32000, 47000, 51000, 71000, 84852};
    double hb = 0, Tb = T_sl, pb = p_sl;
    bool done = false;
    T = Tb; p = pb;
#pragma unroll 1
    for (int i = 0; i < 7 && !done; i++) {
        const double hh = (h < hc[i]) ? h : hc[i];
        done = h < hc[i];
        const double Tn = Tb + beta[i] * (hh - hb);
        double pn;
        if (beta[i] != 0.0) pn = pb * exp(-isa::g_std / (beta[i] * isa::R) * log(1 + beta[i] / Tb * (hh - hb)));
        else pn = pb * exp(-isa::g_std / (isa::R * Tb) * (hh - hb));
        T = Tn; p = pn;
        hb = hc[i]; Tb = Tn; pb = pn;
    }
    if (!done) st |= FB_ST_ISA_RANGE;
}

// ---------------------------------------------------------------------------------------------
// continuous PI compensator with anti-windup (FlightPhysics/src/control.jl:52-81), one channel
FBD double pi_ode(double k_p, double k_i, double k_l, double lo, double hi, double input, double x_i, double& output) {
    const double out_free = k_p * input + x_i;
    output = fmin(fmax(out_free, lo), hi);
    const int sat = (out_free >= hi ? 1 : 0) - (out_free <= lo ? 1 : 0);
    const bool halted = (input * sat) > 0;  // sat_ext is never driven on this path (stays 0)
    return (halted ? 0.0 : k_i * input) - k_l * x_i;
}

// ---------------------------------------------------------------------------------------------
// Model constants of Cessna172Sv0
namespace c172 {
constexpr double D2R = PI / 180;
// aero (FlightApps/src/c172/c172.jl:247-258)
constexpr double S = 16.165, b = 10.912, c = 1.494, V_min = 1.0, tau_filt = 0.02;
constexpr double de_lo = -28 * D2R, de_hi = 23 * D2R, da_lo = -20 * D2R, da_hi = 20 * D2R;
constexpr double dr_lo = -16 * D2R, dr_hi = 16 * D2R, df_lo = 0.0, df_hi = 30 * D2R;
constexpr double alpha_stall_lo = 0.09, alpha_stall_hi = 0.36;
// landing gear (c172.jl:442-476): strut attachment points, dampers (k_s, k_d_ext, k_d_cmp)
constexpr double ldg_r[3][3] = {{-0.381, -1.092, 1.902}, {-0.381, 1.092, 1.902}, {1.27, 0.0, 1.9}};
constexpr double ldg_ks[3] = {39404, 39404, 26269};
constexpr double ldg_kd[3] = {9340, 9340, 3503};
constexpr double psi_max = PI / 6;  // DirectSteering default (FlightPhysics/src/landinggear.jl:47-49)
// contact friction regulator (landinggear.jl:401-409)
constexpr double frc_kp = 5.0, frc_ki = 400.0, frc_kl = 0.2;
// power plant (FlightApps/src/c172/c172s/c172s.jl:16-34; FlightPhysics/src/piston.jl:25-35)
constexpr double P_rated = 735.49875 * 200, w_rated = 2700 * PI / 30, w_stall = 300 * PI / 30, w_idle = 600 * PI / 30;
constexpr double tau_start = 40, J_eng = 0.05;
constexpr double f_lean = 0.0625, f_rich = 0.0950;
constexpr double prop_d = 2.0, prop_Jxx = 0.3, prop_r[3] = {2.055, 0.0, 0.833};
// airframe mass (c172.jl:26-35): RigidBodyDistribution(767, diag(820,1164,1702)) at r = (0.056, 0, 0.582)
constexpr double afm_m = 767.0, afm_J[3] = {820.0, 1164.0, 1702.0}, afm_r[3] = {0.056, 0.0, 0.582};
// payload slots (c172.jl:513-519) and fuel tanks (c172.jl:628-629)
constexpr double pld_r[5][3] = {{0.183, -0.356, 0.899}, {0.183, 0.356, 0.899}, {-0.681, -0.356, 0.899}, {-0.681, 0.356, 0.899}, {-1.316, 0.0, 0.899}};
constexpr double fuel_r[2][3] = {{0.325, -2.845, 0.0}, {0.325, 2.845, 0.0}};
constexpr double m_full = 114.4, m_res = 1.0;
}  // namespace c172

FBD double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
struct Env {  // fb_params subset, wave-uniform (SGPRs)
    double T_sl, p_sl, wind_n, wind_e, wind_d, h_trn;
    int surface;
};
// Per-aircraft inputs, constant during a launch. Only what the airborne path needs every RHS lives in
// registers (deflections are pre-scaled once); the ground-only inputs (steering, brakes) are fetched from
// global memory inside the rare contact branch.
struct Inputs {
    double de, da, dr, df;     // surface deflections [rad] (c172s.jl:92-120 sign conventions + c172.jl:337-340 scaling)
    double throttle, mixture;  // engine (Ranged [0,1])
    double m_pld[5];           // payload masses (Ranged [0,100])
    int ui;                    // FB_UI_* bits
    const double* u_glob;      // &u[0*n + i] for ground-only inputs; may be null (then they read as 0)
    int64_t n;
    FBD double get_de() const { return de; }
    FBD double get_da() const { return da; }
    FBD double get_dr() const { return dr; }
    FBD double get_df() const { return df; }
    FBD double get_throttle() const { return throttle; }
    FBD double get_mixture() const { return mixture; }
    FBD double get_m_pld(int k) const { return m_pld[k]; }
    FBD double get_steering() const {
        return u_glob ? clampd(clampd(u_glob[FB_U_RUDDER * n], -1, 1) + clampd(u_glob[FB_U_RUDDER_OFFSET * n], -1, 1), -1, 1) : 0.0;
    }
    FBD double get_brake(int g) const { return u_glob ? clampd(u_glob[(g == 0 ? FB_U_BRAKE_LEFT : FB_U_BRAKE_RIGHT) * n], 0, 1) : 0.0; }
};
// The same eleven values parked in an LDS panel [11][STRIDE] (row = quantity, column = lane) and read at the
// point of use: the stepping kernel cannot afford 22 VGPRs for values that are touched once per RHS.
template <int STRIDE>
struct InputsLds {
    lds_cptr p;                // &panel[lane]
    int ui;
    const double* u_glob;
    int64_t n;
    FBD double get_de() const { return p[0 * STRIDE]; }
    FBD double get_da() const { return p[1 * STRIDE]; }
    FBD double get_dr() const { return p[2 * STRIDE]; }
    FBD double get_df() const { return p[3 * STRIDE]; }
    FBD double get_throttle() const { return p[4 * STRIDE]; }
    FBD double get_mixture() const { return p[5 * STRIDE]; }
    FBD double get_m_pld(int k) const { return p[(6 + k) * STRIDE]; }
    FBD double get_steering() const {
        return u_glob ? clampd(clampd(u_glob[FB_U_RUDDER * n], -1, 1) + clampd(u_glob[FB_U_RUDDER_OFFSET * n], -1, 1), -1, 1) : 0.0;
    }
    FBD double get_brake(int g) const { return u_glob ? clampd(u_glob[(g == 0 ? FB_U_BRAKE_LEFT : FB_U_BRAKE_RIGHT) * n], 0, 1) : 0.0; }
    FBD void store(lds_ptr q, const Inputs& in) {
        q[0 * STRIDE] = in.de; q[1 * STRIDE] = in.da; q[2 * STRIDE] = in.dr; q[3 * STRIDE] = in.df;
        q[4 * STRIDE] = in.throttle; q[5 * STRIDE] = in.mixture;
#pragma unroll
        for (int k = 0; k < 5; k++) q[(6 + k) * STRIDE] = in.m_pld[k];
        p = q; ui = in.ui; u_glob = in.u_glob; n = in.n;
    }
};
constexpr int INPUT_PANEL_ROWS = 11;
// Cessna172X (fly-by-wire): surfaces, throttle, nose-wheel steering and brakes follow the actuator POSITIONS, which are
// states (device rows 27..33: throttle, aileron, elevator, rudder, flaps, brake_left, brake_right), saturated to their
// Ranged type on the way out (Actuator1.f_ode!: pos = R(x.p), c172x.jl:45; assign!, c172x.jl:127-143: aero.u.e = -elevator.pos,
// aero.u.a = aileron.pos, aero.u.r = -rudder.pos, aero.u.f = flaps.pos, throttle, steering = rudder.pos, brakes).
// Mixture and payload are launch constants read from global memory at the point of use (L1/L2 resident).
constexpr int X2_ACT = FB_NX;  // device row of the first actuator position
struct InputsX {
    const double* xa;      // the seven actuator positions of the state being evaluated (caller's registers)
    const double* u_glob;  // &u[0 * n + i]
    int64_t n;
    int ui;
    FBD double pos(int k) const { return clampd(xa[k], (k == FB_ACT_THROTTLE || k >= FB_ACT_FLAPS) ? 0.0 : -1.0, 1.0); }
    FBD double get_de() const { return c172::de_lo + (c172::de_hi - c172::de_lo) / 2 * (clampd(-pos(FB_ACT_ELEVATOR), -1, 1) + 1); }
    FBD double get_da() const { return c172::da_lo + (c172::da_hi - c172::da_lo) / 2 * (clampd(pos(FB_ACT_AILERON), -1, 1) + 1); }
    FBD double get_dr() const { return c172::dr_lo + (c172::dr_hi - c172::dr_lo) / 2 * (clampd(-pos(FB_ACT_RUDDER), -1, 1) + 1); }
    FBD double get_df() const { return c172::df_lo + (c172::df_hi - c172::df_lo) / 1 * (clampd(pos(FB_ACT_FLAPS), 0, 1) - 0); }
    FBD double get_throttle() const { return pos(FB_ACT_THROTTLE); }
    FBD double get_mixture() const { return clampd(u_glob[FB_U_MIXTURE * n], 0, 1); }
    FBD double get_m_pld(int k) const { return clampd(u_glob[(FB_U_M_PILOT + k) * n], 0, 100); }
    FBD double get_steering() const { return pos(FB_ACT_RUDDER); }
    FBD double get_brake(int g) const { return pos(g == 0 ? FB_ACT_BRAKE_LEFT : FB_ACT_BRAKE_RIGHT); }
};
constexpr double ACT_TAU = 1.0 / 20;  // Actuator1 time constant, c172x.jl:21
// mechanical actuation + Ranged saturation + linear_scaling, from the raw FB_U_* inputs
FBD void make_inputs(Inputs& in, const double* u, int64_t stride, int ui) {
    using namespace c172;
    const double ail = clampd(u[FB_U_AILERON * stride], -1, 1) + clampd(u[FB_U_AILERON_OFFSET * stride], -1, 1);
    const double elv = clampd(u[FB_U_ELEVATOR * stride], -1, 1) + clampd(u[FB_U_ELEVATOR_OFFSET * stride], -1, 1);
    const double rud = clampd(u[FB_U_RUDDER * stride], -1, 1) + clampd(u[FB_U_RUDDER_OFFSET * stride], -1, 1);
    const double aero_e = clampd(-elv, -1, 1), aero_a = clampd(ail, -1, 1), aero_r = clampd(-rud, -1, 1);
    const double aero_f = clampd(u[FB_U_FLAPS * stride], 0, 1);
    in.de = de_lo + (de_hi - de_lo) / 2 * (aero_e + 1);
    in.da = da_lo + (da_hi - da_lo) / 2 * (aero_a + 1);
    in.dr = dr_lo + (dr_hi - dr_lo) / 2 * (aero_r + 1);
    in.df = df_lo + (df_hi - df_lo) / 1 * (aero_f - 0);
    in.throttle = clampd(u[FB_U_THROTTLE * stride], 0, 1);
    in.mixture = clampd(u[FB_U_MIXTURE * stride], 0, 1);
#pragma unroll
    for (int k = 0; k < 5; k++) in.m_pld[k] = clampd(u[(FB_U_M_PILOT + k) * stride], 0, 100);
    in.ui = ui;
}
// quantities f_step! reads from the y of the last f_ode! (aircraftbase.jl:172-181; c172.jl:715-724)
struct StepAux {
    double alpha;       // aero.y.α (c172.jl:375-384)
    double m_avail;     // fuel.y.m_avail (c172.jl:641)
    int wow;            // bit g: strut g weight-on-wheel (landinggear.jl:479-483)
    int crash;          // any strut: α_ts > 60° or -ξ_dot > 10 (landinggear.jl:331-347)
};

// accumulate point mass into (m, m r, J)  (dynamics.jl:211-214)
FBD void add_point(double m, const double* r, double& M, v3& Mr, double (&J)[6]) {
    M += m;
    Mr = Mr + m * v3{r[0], r[1], r[2]};
    J[0] += m * (r[1] * r[1] + r[2] * r[2]);
    J[1] += m * (r[0] * r[0] + r[2] * r[2]);
    J[2] += m * (r[0] * r[0] + r[1] * r[1]);
    J[3] -= m * (r[0] * r[1]);  // xy
    J[4] -= m * (r[0] * r[2]);  // xz
    J[5] -= m * (r[1] * r[2]);  // yz
}

// Ground-contact branch of one landing-gear unit (landinggear.jl:260-328, 426-476). Rare and divergent:
// kept out of line so that the airborne path does not pay its registers.
struct GroundIn {
    quat q_eb, q_nb, q_en;
    v3 r_eb_e, r_bs_e, ks_e, w_eb_b, v_eb_b, loc_Ot;
    double he_Ot, frc_out0, frc_out1, steer_in, brake_in;
    int g, steer_engaged, surface;
};
struct GroundOut {
    v3 F_b, tau_b;
    double v_xy0, v_xy1, xi, xi_dot, F_dmp, alpha_ts;
    quat q_sc;   // strut -> contact frame rotation, kept between the two phases
    v3 r_bc_b;   // contact point in body frame
    int st;
};
__device__ FB_GROUND_ATTR void gear_ground_kinematics(const GroundIn& in, GroundOut& o);
__device__ FB_GROUND_ATTR void gear_ground_force(const GroundIn& in, GroundOut& o);

// ---------------------------------------------------------------------------------------------
// One RHS evaluation. x[27] -> 27 derivatives through emit(); fills aux (for f_step!) and, if Y != nullptr, the output record.
// Call order of the reference: world.jl:26-32 -> aircraftbase.jl:221-230,142-170 -> c172.jl:697-713.
// Every derivative component is handed to `emit(index, value)` the moment it is known, so that the
// caller can consume it at once (the stepping kernel folds it into the RK stage sums in LDS) instead of
// keeping a 27-double array alive across the whole evaluation.
// Output sinks: where the components of the output record y go.
//   NoSink    — nowhere (the stepping kernel): everything that only feeds y is dead code;
//   PanelSink — the full record into a global [FB_NY x n] panel (f_ode!, logging);
//   a partial sink (enabled, !full) picks a few components — the control laws' inputs — and still allows the
//   shortcuts that a full record forbids.
struct NoSink {
    static constexpr bool enabled = false, full = false;
    FBD void put(int, double) const {}
};
struct PanelSink {
    static constexpr bool enabled = true, full = true;
    double* Y;   // &y[0 * n + i]
    int64_t n;
    FBD void put(int k, double v) const { Y[(int64_t)k * n] = v; }
};
// GROUND = false compiles the landing gear's ground-contact branch (and the three per-wheel geodetic conversions) out: such an
// instance can only evaluate states in the high-clearance regime and answers FB_ST_INTERNAL_REDO anywhere else, so that the
// caller repeats the work with the full instance (k_step: the airborne pass, then a pass over the lanes that asked for it).
constexpr int32_t FB_ST_INTERNAL_REDO = 1 << 30;
template <int KIN, bool GROUND = true, class Sink, class Emit, class In, int NXT>
FBD int32_t rhs(const double (&x)[NXT], int stall, int eng_state, const In& in, const Env& env, const Tables& T,
                Emit&& emit, StepAux& aux, Sink&& sink) {
    using namespace c172;
    using SinkT = typename std::remove_reference<Sink>::type;
    constexpr bool WITH_Y = SinkT::enabled;
    constexpr int NC = SinkT::full ? PR_NC : PR_NC_STEP;   // propeller table stride in LDS (see stage_tables)
    int32_t st = 0;
    auto YP = [&](int k, double v) { if (WITH_Y) sink.put(k, v); };
    auto YP3 = [&](int k, v3 v) { if (WITH_Y) { sink.put(k, v.x); sink.put(k + 1, v.y); sink.put(k + 2, v.z); } };
    auto YP4 = [&](int k, quat q) { if (WITH_Y) { sink.put(k, q.w); sink.put(k + 1, q.x); sink.put(k + 2, q.y); sink.put(k + 3, q.z); } };

    // ===== kinematics (FlightPhysics/src/kinematics.jl): WA :181-223, ECEF :282-320, NED :366-407 =====
    // Rows FB_X_Q_WB.. hold the mechanisation's own states: WA q_wb[4] q_ew[4] h_e; ECEF q_eb[4] n_e[3] h_e (one row unused);
    // NED ψ θ φ ϕ λ h_e (three rows unused; unused rows carry a zero derivative).
    constexpr int KX = FB_X_Q_WB;
    const v3 w_eb_b = {x[FB_X_OMEGA_EB_B], x[FB_X_OMEGA_EB_B + 1], x[FB_X_OMEGA_EB_B + 2]};
    const v3 v_eb_b = {x[FB_X_V_EB_B], x[FB_X_V_EB_B + 1], x[FB_X_V_EB_B + 2]};
    const double h_e = x[KIN == FB_KIN_WA ? KX + 8 : (KIN == FB_KIN_ECEF ? KX + 7 : KX + 5)];
    if (!(h_e >= H_MIN)) st |= FB_ST_ALT_RANGE;
    quat q_nb, q_eb, q_en = {1, 0, 0, 0}, q_nw = {1, 0, 0, 0};
    v3 n_e;
    if constexpr (KIN == FB_KIN_WA) {
        const quat q_wb = {x[KX], x[KX + 1], x[KX + 2], x[KX + 3]};
        const quat q_ew = {x[KX + 4], x[KX + 5], x[KX + 6], x[KX + 7]};
        // ψ_nw and n_e straight from q_ew (geodesy.jl:62-69, 140-147)
        const double dq12 = 2 * q_ew.w * q_ew.x, dq13 = 2 * q_ew.w * q_ew.y;
        const double dq24 = 2 * q_ew.x * q_ew.z, dq34 = 2 * q_ew.y * q_ew.z;
        n_e = {-(dq24 + dq13), -(dq34 - dq12), -(1 - 2 * (q_ew.x * q_ew.x + q_ew.y * q_ew.y))};
        // q_nw = Rz(ψ_nw), ψ_nw = atan2(-(dq34+dq12), dq24-dq13): only cos, sin of ψ_nw/2 are needed
        double s_nw, c_nw;
        half_angle_cs(-(dq34 + dq12), dq24 - dq13, c_nw, s_nw);
        q_nw = {c_nw, 0.0, 0.0, s_nw};
        q_nb = qmul(q_nw, q_wb);
        q_eb = qmul(q_ew, q_wb);
    } else if constexpr (KIN == FB_KIN_ECEF) {
        q_eb = {x[KX], x[KX + 1], x[KX + 2], x[KX + 3]};
        n_e = {x[KX + 4], x[KX + 5], x[KX + 6]};   // state n-vector, normalised only in f_step! (kinematics.jl:286, 317-320)
        q_en = ltf_quat(n_e);
        q_nb = qmul(qconj(q_en), q_eb);
    } else {
        double s1, c1, s2, c2, s3, c3, sla, cla, slo, clo;
        sincos(0.5 * x[KX], &s1, &c1); sincos(0.5 * x[KX + 1], &s2, &c2); sincos(0.5 * x[KX + 2], &s3, &c3);
        q_nb = qmul(qmul(quat{c1, 0, 0, s1}, quat{c2, 0, s2, 0}), quat{c3, s3, 0, 0});   // Rz(ψ) ∘ Ry(θ) ∘ Rx(φ), attitude.jl:393-395
        sincos(x[KX + 3], &sla, &cla); sincos(x[KX + 4], &slo, &clo);
        n_e = {cla * clo, cla * slo, sla};                                               // geodesy.jl:97-101
        q_en = ltf_quat(n_e);
        q_eb = qmul(q_en, q_nb);
    }

    double lat, lon;
    const double N_geoid = geoid_height(T, n_e, lat, lon);
    const double h_o = h_e - N_geoid;
    if (!(h_o >= H_MIN)) st |= FB_ST_ALT_RANGE;

    const v3 v_eb_n = qrot(q_nb, v_eb_b);
    // radii of curvature and ECEF position (geodesy.jl:125-129, 418-428)
    const double f_den = sqrt(1 - wgs::e2 * n_e.z * n_e.z);
    const double R_E = wgs::a / f_den;
    const double R_N = wgs::a * (1 - wgs::e2) / (f_den * f_den * f_den);
    const v3 r_eb_e = {(R_E + h_e) * n_e.x, (R_E + h_e) * n_e.y, (R_E * (1 - wgs::e2) + h_e) * n_e.z};
    // transport rate (kinematics.jl:232-242)
    const v3 w_ew_n = {v_eb_n.y / (R_E + h_e), -v_eb_n.x / (R_N + h_e), 0.0};
    v3 w_wb_b;
    if constexpr (KIN == FB_KIN_WA) {
        const quat q_wb = {x[KX], x[KX + 1], x[KX + 2], x[KX + 3]};
        const quat q_ew = {x[KX + 4], x[KX + 5], x[KX + 6], x[KX + 7]};
        const v3 w_ew_w = qrot_inv(q_nw, w_ew_n);
        const v3 w_ew_b = qrot_inv(q_wb, w_ew_w);
        w_wb_b = w_eb_b - w_ew_b;
        const quat a = qmul(q_wb, quat{0.0, w_wb_b.x, w_wb_b.y, w_wb_b.z});
        const quat b2 = qmul(q_ew, quat{0.0, w_ew_w.x, w_ew_w.y, w_ew_w.z});
        emit(KX, 0.5 * a.w); emit(KX + 1, 0.5 * a.x); emit(KX + 2, 0.5 * a.y); emit(KX + 3, 0.5 * a.z);
        emit(KX + 4, 0.5 * b2.w); emit(KX + 5, 0.5 * b2.x); emit(KX + 6, 0.5 * b2.y); emit(KX + 7, 0.5 * b2.z);
        emit(KX + 8, -v_eb_n.z);
    } else if constexpr (KIN == FB_KIN_ECEF) {
        w_wb_b = w_eb_b - qrot_inv(q_nb, w_ew_n);
        const quat a = qmul(q_eb, quat{0.0, w_eb_b.x, w_eb_b.y, w_eb_b.z});   // Attitude.dt(q_eb, ω_eb_b)
        const v3 nd = qrot(q_en, cross(w_ew_n, v3{0.0, 0.0, -1.0}));           // kinematics.jl:309
        emit(KX, 0.5 * a.w); emit(KX + 1, 0.5 * a.x); emit(KX + 2, 0.5 * a.y); emit(KX + 3, 0.5 * a.z);
        emit(KX + 4, nd.x); emit(KX + 5, nd.y); emit(KX + 6, nd.z);
        emit(KX + 7, -v_eb_n.z);
        emit(KX + 8, 0.0);
    } else {
        w_wb_b = w_eb_b - qrot_inv(q_nb, w_ew_n);
        // ω_en_n (kinematics.jl:413-425) with ϕ = LatLon(Ob).ϕ re-derived from n_e, like the reference
        const v3 w_en_n = {w_ew_n.x, w_ew_n.y, -v_eb_n.y * tan(lat) / (R_E + h_e)};
        const v3 w_nb_b = w_eb_b - qrot_inv(q_nb, w_en_n);
        double sph, cph;
        sincos(x[KX + 2], &sph, &cph);
        const double tth = tan(x[KX + 1]), sec = 1.0 / cos(x[KX + 1]);
        emit(KX, sph * sec * w_nb_b.y + cph * sec * w_nb_b.z);                 // Attitude.dt(e_nb, ω_nb_b), attitude.jl:436-449
        emit(KX + 1, cph * w_nb_b.y - sph * w_nb_b.z);
        emit(KX + 2, w_nb_b.x + sph * tth * w_nb_b.y + cph * tth * w_nb_b.z);
        emit(KX + 3, -w_en_n.y);                                               // Geodesy.dt(ϕ_λ, ω_en_n), geodesy.jl:112-118
        emit(KX + 4, w_en_n.x / cos(x[KX + 3]));
        emit(KX + 5, -v_eb_n.z);
        emit(KX + 6, 0.0); emit(KX + 7, 0.0); emit(KX + 8, 0.0);
    }
    if (WITH_Y) {
        if constexpr (KIN == FB_KIN_WA) q_en = qmul(q_eb, qconj(q_nb));
        if constexpr (KIN == FB_KIN_NED) {   // e_nb, ϕ_λ are the states themselves
            YP(FB_Y_KIN + 0, x[KX]); YP(FB_Y_KIN + 1, x[KX + 1]); YP(FB_Y_KIN + 2, x[KX + 2]);
            lat = x[KX + 3]; lon = x[KX + 4];
        } else {
            // Euler angles (attitude.jl:382-391)
            const double q1 = q_nb.w, q2 = q_nb.x, q3 = q_nb.y, q4 = q_nb.z;
            YP(FB_Y_KIN + 0, atan2(2 * (q1 * q4 + q2 * q3), 1 - 2 * (q3 * q3 + q4 * q4)));
            YP(FB_Y_KIN + 1, asin(fmin(fmax(2 * (q1 * q3 - q2 * q4), -1.0), 1.0)));
            YP(FB_Y_KIN + 2, atan2(2 * (q1 * q2 + q3 * q4), 1 - 2 * (q2 * q2 + q3 * q3)));
        }
        YP4(FB_Y_KIN + 3, q_nb); YP4(FB_Y_KIN + 7, q_eb); YP4(FB_Y_KIN + 11, q_en);
        YP(FB_Y_KIN + 15, lat); YP(FB_Y_KIN + 16, lon); YP3(FB_Y_KIN + 17, n_e);
        YP(FB_Y_KIN + 20, h_e); YP(FB_Y_KIN + 21, h_o); YP3(FB_Y_KIN + 22, r_eb_e);
        YP3(FB_Y_KIN + 25, w_wb_b); YP3(FB_Y_KIN + 28, w_eb_b); YP3(FB_Y_KIN + 31, v_eb_b); YP3(FB_Y_KIN + 34, v_eb_n);
        const double v_gnd = norm(v_eb_n);
        YP(FB_Y_KIN + 37, v_gnd);
        const bool chi_ok = KIN == FB_KIN_NED || v_gnd > 0.1;   // the NED mechanisation has no low-speed guard (kinematics.jl:395-396)
        YP(FB_Y_KIN + 38, chi_ok ? atan2(v_eb_n.y, v_eb_n.x) : 0.0);
        YP(FB_Y_KIN + 39, chi_ok ? atan2(-v_eb_n.z, sqrt(v_eb_n.x * v_eb_n.x + v_eb_n.y * v_eb_n.y)) : 0.0);
    }

    FB_PHASE_FENCE();
    // ===== air data (atmosphere.jl:269-283, 220-242) =====
    double T_air, p_air;
    isa_data(h_o * wgs::a / (wgs::a + h_o), env.T_sl, env.p_sl, T_air, p_air, st);
    const double rho = p_air / (isa::R * T_air);
    const double a_snd = sqrt(isa::gamma * isa::R * T_air);
    const v3 v_ew_n = {env.wind_n, env.wind_e, env.wind_d};
    const v3 v_ew_b = qrot_inv(q_nb, v_ew_n);
    const v3 v_wb_b = v_eb_b - v_ew_b;
    const double TAS = norm(v_wb_b);
    const double q_dyn = 0.5 * rho * (TAS * TAS);
    if (WITH_Y) {
        YP3(FB_Y_AIR, v_ew_n); YP3(FB_Y_AIR + 3, v_ew_b); YP3(FB_Y_AIR + 6, v_wb_b);
        const double M = TAS / a_snd;
        const double Tt = T_air * (1 + (isa::gamma - 1) / 2 * (M * M));
        const double pt = p_air * pow(Tt / T_air, isa::gamma / (isa::gamma - 1));
        const double dp = pt - p_air;
        YP(FB_Y_AIR + 9, T_air); YP(FB_Y_AIR + 10, p_air); YP(FB_Y_AIR + 11, rho); YP(FB_Y_AIR + 12, a_snd);
        YP(FB_Y_AIR + 13, (isa::beta_s * pow(T_air, 1.5)) / (T_air + isa::S));
        YP(FB_Y_AIR + 14, M); YP(FB_Y_AIR + 15, Tt); YP(FB_Y_AIR + 16, pt); YP(FB_Y_AIR + 17, dp); YP(FB_Y_AIR + 18, q_dyn);
        YP(FB_Y_AIR + 19, TAS); YP(FB_Y_AIR + 20, TAS * sqrt(rho / isa::rho_std));
        YP(FB_Y_AIR + 21, sqrt(2 * isa::gamma / (isa::gamma - 1) * isa::p_std / isa::rho_std *
                               (pow(1 + dp / isa::p_std, (isa::gamma - 1) / isa::gamma) - 1)));
    }

    v3 F_b = {0, 0, 0}, tau_b = {0, 0, 0};  // total external wrench at Ob, body axes

    FB_PHASE_FENCE();
    // ===== aerodynamics (c172.jl:307-373, 226-245) =====
    {
        lds_cptr A = T.lds + LDS_AERO;
        lds_cptr RA = T.rk + LDS_AERO;
        double alpha = 0, beta = 0;
        if (TAS > 0.1) {  // also covers get_airflow_angles' own ‖v‖ < 0.1 guard (atmosphere.jl:329-337)
            alpha = atan2(v_wb_b.z, v_wb_b.x);
            beta = atan2(v_wb_b.y, sqrt(v_wb_b.x * v_wb_b.x + v_wb_b.z * v_wb_b.z));
        }
        const double V = fmax(TAS, V_min);
        const double afd = 1 / tau_filt * (alpha - x[FB_X_ALPHA_FILT]);
        const double bfd = 1 / tau_filt * (beta - x[FB_X_BETA_FILT]);
        emit(FB_X_ALPHA_FILT, afd);
        emit(FB_X_BETA_FILT, bfd);
        const double i2V = 1 / (2 * V);
        const double p_nd = w_wb_b.x * b * i2V, q_nd = w_wb_b.y * c * i2V, r_nd = w_wb_b.z * b * i2V;
        const double ad_nd = clampd(afd * c * i2V, -0.04, 0.04);
        const double de = in.get_de(), da = in.get_da(), dr = in.get_dr(), df = in.get_df();
        const double dh_nd = (h_o - env.h_trn) / b;
        const double al = clampd(alpha, -0.1, 0.36), be = clampd(beta, -0.2, 0.2);

        const loc l_ge = grid_locate<13>(A + AT_GE_K, RA + AT_GE_K, dh_nd, true, true);
        const loc l_df4 = grid_locate<4>(A + AT_DF4_K, RA + AT_DF4_K, df, true, true);
        const loc l_df2 = grid_locate<2>(A + AT_DF2_K, RA + AT_DF2_K, df, true, true);
        const loc l_al26 = grid_locate<26>(A + AT_CD_ALPHA_K, RA + AT_CD_ALPHA_K, al, true, true);
        const loc l_al17 = grid_locate<17>(A + AT_CL_ALPHA_K, RA + AT_CL_ALPHA_K, al, true, true);
        const loc l_al2 = grid_locate<2>(A + AT_ALPHA2_K, RA + AT_ALPHA2_K, al, true, true);
        const loc l_be3 = grid_locate<3>(A + AT_CY_BETA_K, RA + AT_CY_BETA_K, be, true, true);
        const loc l_de = grid_locate<3>(A + AT_UNIT3_K, RA + AT_UNIT3_K, de, true, true);
        const loc l_bu = grid_locate<3>(A + AT_UNIT3_K, RA + AT_UNIT3_K, be, true, true);
        const loc l_stall = {0, stall ? 1.0 : 0.0};
        lds_cptr S_ = A + AT_SCALARS;

        const double C_D = S_[AS_CD_ZERO] + lerp1(A + AT_CD_GE_V, l_ge) * (lerp2(A + AT_CD_ALPHA_DF_V, 26, l_al26, l_df4) + lerp1(A + AT_CD_DF_V, l_df4)) +
                           lerp1(A + AT_CD_DE_V, l_de) + lerp1(A + AT_CD_BETA_V, l_bu);
        const double C_Y = S_[AS_CY_DR] * dr + S_[AS_CY_DA] * da + lerp2(A + AT_CY_BETA_DF_V, 3, l_be3, l_df2) +
                           lerp2(A + AT_CY_P_V, 2, l_al2, l_df2) * p_nd + lerp2(A + AT_CY_R_V, 2, l_al2, l_df2) * r_nd;
        const double C_L = lerp1(A + AT_CL_GE_V, l_ge) * (lerp2(A + AT_CL_ALPHA_V, 17, l_al17, l_stall) + lerp1(A + AT_CL_DF_V, l_df4)) +
                           S_[AS_CL_DE] * de + S_[AS_CL_Q] * q_nd + S_[AS_CL_ALPHA_DOT] * ad_nd;
        const double C_l = S_[AS_Cl_DA] * da + S_[AS_Cl_DR] * dr + S_[AS_Cl_BETA] * be + S_[AS_Cl_P] * p_nd +
                           lerp2(A + AT_CL_R_V, 2, l_al2, l_df2) * r_nd;
        const double C_m = S_[AS_CM_ZERO] + S_[AS_CM_DE] * de + lerp1(A + AT_CM_DF_V, l_df4) + S_[AS_CM_ALPHA] * al + S_[AS_CM_Q] * q_nd +
                           S_[AS_CM_ALPHA_DOT] * ad_nd;
        const double C_n = S_[AS_CN_DR] * dr + S_[AS_CN_DA] * da + S_[AS_CN_BETA] * be + S_[AS_CN_P] * p_nd + S_[AS_CN_R] * r_nd;

        // stability -> body axes: rotation by Ry(-α) with the UNCLAMPED α (c172.jl:356-359; atmosphere.jl:353-356)
        double sa, ca;
        sincos(0.5 * (-alpha), &sa, &ca);
        const double qS = q_dyn * S;
        const v3 F_s = {qS * -C_D, qS * C_Y, qS * -C_L};
        const v3 F_a = qrot(quat{ca, 0.0, sa, 0.0}, F_s);
        const v3 tau_a = {qS * (C_l * b), qS * (C_m * c), qS * (C_n * b)};
        F_b = F_b + F_a;
        tau_b = tau_b + tau_a;
        aux.alpha = alpha;
        if (WITH_Y) {
            YP(FB_Y_AERO, alpha); YP(FB_Y_AERO + 1, beta); YP(FB_Y_AERO + 2, afd); YP(FB_Y_AERO + 3, bfd);
            YP(FB_Y_AERO + 4, C_D); YP(FB_Y_AERO + 5, C_Y); YP(FB_Y_AERO + 6, C_L); YP(FB_Y_AERO + 7, C_l); YP(FB_Y_AERO + 8, C_m); YP(FB_Y_AERO + 9, C_n);
            YP3(FB_Y_AERO + 10, F_a); YP3(FB_Y_AERO + 13, tau_a);
        }
    }

    FB_PHASE_FENCE();
    // ===== landing gear: left, right, nose (landinggear.jl:524-537, 228-328, 411-476) =====
    aux.wow = 0;
    aux.crash = 0;
    // High-clearance shortcut (exact for the state): every strut attachment is within 2.3 m of Ob and the geoid
    // moves by < 1 mm over that distance, so when Ob is more than 10 m above the terrain no wheel can touch it:
    // wow = false, zero wrench, regulator input 0 (landinggear.jl:255-258, 418-424) without evaluating the three
    // ECEF->geodetic conversions and geoid gathers. Not taken when the FULL output record (which logs Δh) is requested.
    static_assert(GROUND || !SinkT::full, "the full output record needs the ground-capable instance");
    const bool clear10 = !SinkT::full && (h_o - env.h_trn > 10.0);
    if (!GROUND && !clear10) st |= FB_ST_INTERNAL_REDO;
    const bool high_clearance = GROUND ? clear10 : true;
#pragma unroll
    for (int g = 0; g < 3; g++) {
        FB_PHASE_FENCE();
        v3 r_bs_e = {0, 0, 0}, loc_Ot = {1, 0, 0};
        double he_Ot = 0, dh = 0;
        bool wow = false;
        if (!high_clearance) {
            const v3 r_bs_b = {ldg_r[g][0], ldg_r[g][1], ldg_r[g][2]};
            r_bs_e = qrot(q_eb, r_bs_b);
            const v3 r_ew0_e = r_eb_e + r_bs_e;  // l_0 = 0
            double he_Ow0;
            geodetic_from_ecef(r_ew0_e, loc_Ot, he_Ow0);
            if (!(he_Ow0 >= H_MIN)) st |= FB_ST_ALT_RANGE;
            he_Ot = env.h_trn + geoid_height(T, loc_Ot);
            dh = he_Ow0 - he_Ot;
            wow = dh <= 0;
        }
        const double x0 = x[FB_X_LDG_FRC + 2 * g], x1 = x[FB_X_LDG_FRC + 2 * g + 1];
        double v_xy0 = 0, v_xy1 = 0;
        GroundOut go;
        go.F_b = {0, 0, 0}; go.tau_b = {0, 0, 0}; go.xi = 0; go.xi_dot = 0; go.F_dmp = 0; go.alpha_ts = 0; go.st = 0;
        GroundIn gi;
        if (wow) {
            gi.q_eb = q_eb; gi.q_nb = q_nb; gi.q_en = qmul(q_eb, qconj(q_nb));
            gi.r_eb_e = r_eb_e; gi.r_bs_e = r_bs_e; gi.ks_e = qrot(q_eb, v3{0, 0, 1});
            gi.w_eb_b = w_eb_b; gi.v_eb_b = v_eb_b; gi.loc_Ot = loc_Ot; gi.he_Ot = he_Ot;
            gi.g = g; gi.surface = env.surface;
            gi.steer_engaged = (in.ui & FB_UI_STEERING_ENGAGED) ? 1 : 0;
            // ground-only inputs, fetched on demand (c172s.jl:107-110; c172x.jl:139-141)
            gi.steer_in = (g == 2) ? in.get_steering() : 0.0;
            gi.brake_in = (g == 2) ? 0.0 : in.get_brake(g);
            gear_ground_kinematics(gi, go);
            v_xy0 = go.v_xy0; v_xy1 = go.v_xy1;
            st |= go.st;
        }
        // friction regulator: input = -v_ec_xy (zero when airborne) (landinggear.jl:418-424)
        double out0, out1;
        emit(FB_X_LDG_FRC + 2 * g, pi_ode(frc_kp, frc_ki, frc_kl, -1.0, 1.0, -v_xy0, x0, out0));
        emit(FB_X_LDG_FRC + 2 * g + 1, pi_ode(frc_kp, frc_ki, frc_kl, -1.0, 1.0, -v_xy1, x1, out1));
        if (wow) {
            gi.frc_out0 = out0; gi.frc_out1 = out1;
            gear_ground_force(gi, go);
            F_b = F_b + go.F_b;
            tau_b = tau_b + go.tau_b;
            aux.wow |= (1 << g);
            if (go.alpha_ts * (180 / PI) > 60) aux.crash = 1;
        }
        if (-go.xi_dot > 10) aux.crash = 1;
        if (WITH_Y) {
            const int k = FB_Y_LDG + 11 * g;
            YP(k, dh); YP(k + 1, wow ? 1.0 : 0.0); YP(k + 2, go.xi); YP(k + 3, go.xi_dot); YP(k + 4, go.F_dmp);
            YP3(k + 5, go.F_b); YP3(k + 8, go.tau_b);
        }
    }

    FB_PHASE_FENCE();
    // ===== power plant: propeller then engine (piston.jl:575-595; propellers.jl:405-452; piston.jl:314-426) =====
    v3 h_rot;
    double mdot;
    {
        lds_cptr PT = T.lds + LDS_PISTON;
        lds_cptr RPT = T.rk + LDS_PISTON;
        lds_cptr PR = T.lds + LDS_PROP;
        const double w_eng = x[FB_X_ENG_OMEGA];
        const double w_prop = w_eng;  // gear ratio 1
        const v3 r_p = {prop_r[0], prop_r[1], prop_r[2]};
        const v3 v_p = v_wb_b + cross(w_eb_b, r_p);
        const double v_J = norm(v_p);
        const double J_adv = 2 * PI * v_J / (fmax(fabs(w_prop), 1.0) * prop_d);
        const double Mt = fabs(w_prop) * (prop_d / 2) / a_snd;
        const loc lj = range_locate(0.0, 1.5, PR_NJ, J_adv, true);
        const loc lm = range_locate(0.0, 1.5, PR_NM, Mt, true);
        lds_cptr c00 = PR + (lj.i + PR_NJ * lm.i) * NC;
        lds_cptr c10 = c00 + NC;
        lds_cptr c01 = c00 + PR_NJ * NC;
        lds_cptr c11 = c01 + NC;
        const double w00 = (1 - lj.w) * (1 - lm.w), w01 = (1 - lj.w) * lm.w, w10 = lj.w * (1 - lm.w), w11 = lj.w * lm.w;
        auto coef = [&](int cidx) { return (w00 * c00[cidx] + w01 * c01[cidx]) + (w10 * c10[cidx] + w11 * c11[cidx]); };
        const double C_Fx = coef(0), C_Mx = coef(1), C_Fz_a = coef(2), C_Mz_a = coef(3);
        double a_p = 0, b_p = 0;
        if (!(v_J < 0.1)) {
            a_p = atan2(v_p.z, v_p.x);
            b_p = atan2(v_p.y, sqrt(v_p.x * v_p.x + v_p.z * v_p.z));
        }
        const double fr = w_prop / (2 * PI), fr2 = fr * fr;
        constexpr double d4 = prop_d * prop_d * prop_d * prop_d, d5 = d4 * prop_d;
        const double kF = rho * fr2 * d4, kM = rho * fr2 * d5;
        const v3 F_p = {kF * C_Fx, kF * (C_Fz_a * b_p), kF * (C_Fz_a * a_p)};
        const v3 tau_p = {kM * C_Mx, kM * (C_Mz_a * b_p), kM * (C_Mz_a * a_p)};  // CW: sense = +1
        const v3 tau_pb = tau_p + cross(r_p, F_p);
        F_b = F_b + F_p;
        tau_b = tau_b + tau_pb;
        h_rot = {prop_Jxx * w_prop, 0.0, 0.0};

        // ---- engine ----
        double out_frc, out_idle;
        emit(FB_X_ENG_FRC, pi_ode(5.0, 200.0, 0.0, -1.0, 1.0, -w_eng, x[FB_X_ENG_FRC], out_frc));
        emit(FB_X_ENG_IDLE, pi_ode(4.0, 2.0, 0.0, -0.5, 0.5, 1 - w_eng / w_idle, x[FB_X_ENG_IDLE], out_idle));
        const double mu_ratio_idle = 0.5 + out_idle;
        const double n_eng = w_eng / w_rated;
        // T_ISA(p) = T_std (p/p_std)^(-βR/g), δ = (p/p_std) (T_ISA/T_std)^-1/2 (piston.jl:38-41)
        const double T_ISA = isa::T_std * exp((6.5e-3 * isa::R / isa::g_std) * log(p_air * (1 / isa::p_std)));
        const double delta = (p_air / isa::p_std) / sqrt(T_ISA / isa::T_std);
        const double throttle = in.get_throttle(), mixture = in.get_mixture();
        const loc l_n2 = range_locate(0.667, 1.0, 2, n_eng, false);
        const double mu_wot = lerp2(PT + PT_MU_WOT_V, 2, l_n2, range_locate(0.441, 1.0, 9, delta, false));
        const double mu = mu_wot * (mu_ratio_idle + throttle * (1 - mu_ratio_idle));
        const double k_f = 1 / sqrt(rho / isa::rho_std);
        // Branch-free on purpose: the running-engine chain (five dependent table lookups) is evaluated for every
        // lane and the off / starting cases are selected at the end, so that the whole power-plant phase is one
        // basic block the scheduler can interleave (at one wave per SIMD, LDS latency is hidden only by ILP).
        // Every lookup clamps or extrapolates its index, so evaluating it for a stopped engine is harmless.
        const double mixture_pos = (in.ui & FB_UI_MIXTURE_AUTO) ? (f_lean + mixture * (f_rich - f_lean)) / (k_f * f_rich) : 0.5 * (mixture + 1);
        const double f_run = k_f * (f_rich * mixture_pos);
        // compute_π_ISA_pow (piston.jl:457-477)
        const loc l_n13 = grid_locate<13>(PT + PT_PISTD_N_K, RPT + PT_PISTD_N_K, n_eng, true, true);
        const loc l_n5w = grid_locate<5>(PT + PT_PIWOT_N_K, RPT + PT_PIWOT_N_K, n_eng, true, true);
        const loc l_n5s = grid_locate<5>(PT + PT_SFC_N_K, RPT + PT_SFC_N_K, n_eng, false, false);
        const loc l_f = grid_locate<11>(PT + PT_F_K, RPT + PT_F_K, f_run, true, true);
        const double pi_ratio = lerp1(PT + PT_PI_RATIO_V, l_f), sfc_ratio = lerp1(PT + PT_SFC_RATIO_V, l_f);
        const double d_wot = lerp2(PT + PT_DELTA_WOT_V, 2, l_n2, range_locate(0.401, 0.936, 9, mu, false));
        const double pi_std = lerp2(PT + PT_PISTD_V, 13, l_n13, grid_locate<3>(PT + PT_PISTD_MU_K, RPT + PT_PISTD_MU_K, mu, true, true));
        const double pi_wot = lerp2(PT + PT_PIWOT_V, 5, l_n5w, grid_locate<3>(PT + PT_PIWOT_D_K, RPT + PT_PIWOT_D_K, d_wot, true, false));
        double pi_isa = (fabs(d_wot - 1) < 5e-3) ? pi_std : pi_std + (pi_wot - pi_std) / (d_wot - 1) * (delta - 1);
        pi_isa = fmax(pi_isa, 0.0);
        const double pi_pow = pi_isa * sqrt(T_ISA / T_air);
        const double pi_act = pi_pow * pi_ratio;
        const double P_run = P_rated * pi_act;
        const double tau_run = (w_eng > 0) ? P_run / w_eng : 0.0;
        const double SFC_run = lerp2(PT + PT_SFC_POW_V, 5, l_n5s, grid_locate<8>(PT + PT_SFC_PI_K, RPT + PT_SFC_PI_K, pi_act, false, false)) * sfc_ratio;
        const bool eng_off = eng_state == 0, eng_starting = eng_state == 1, eng_running = !(eng_off || eng_starting);
        const double MAP = eng_off ? p_air : mu * isa::p_std;
        const double f_ar = eng_running ? f_run : 0.0;
        const double tau_shaft = eng_off ? out_frc * (0.01 * P_rated / w_rated) : (eng_starting ? tau_start : tau_run);
        const double P_shaft = eng_off ? 0.0 : (eng_starting ? tau_start * w_eng : P_run);
        const double SFC = eng_running ? SFC_run : 0.0;
        mdot = eng_running ? SFC_run * P_run : 0.0;
        const double tau_load = tau_p.x;  // gear_ratio * τ_prop
        emit(FB_X_ENG_OMEGA, (tau_shaft + tau_load) / (J_eng + prop_Jxx));
        if (WITH_Y) {
            const int k = FB_Y_PWP;
            YP(k, MAP); YP(k + 1, f_ar); YP(k + 2, mdot); YP(k + 3, w_eng); YP(k + 4, tau_shaft); YP(k + 5, P_shaft); YP(k + 6, SFC);
            YP(k + 7, out_idle); YP(k + 8, out_frc);
            YP(k + 9, J_adv); YP(k + 10, Mt); YP3(k + 11, F_p); YP3(k + 14, tau_pb); YP3(k + 17, h_rot);
            if constexpr (SinkT::full) { YP(k + 20, rho * fabs(fr * fr2) * d5 * coef(4)); YP(k + 21, coef(5)); }
        }
    }

    FB_PHASE_FENCE();
    // ===== fuel (c172.jl:607-616) =====
    const double m_fuel_total = m_res + x[FB_X_FUEL] * (m_full - m_res);
    emit(FB_X_FUEL, -mdot / (m_full - m_res));
    aux.m_avail = m_fuel_total - m_res;
    YP(FB_Y_FUEL, m_fuel_total);

    FB_PHASE_FENCE();
    // ===== total mass properties at Ob (dynamics.jl:328-399; c172.jl:26-44, 542-554, 618-636) =====
    double M = afm_m;
    v3 Mr = afm_m * v3{afm_r[0], afm_r[1], afm_r[2]};
    double J[6] = {afm_J[0] + afm_m * (afm_r[1] * afm_r[1] + afm_r[2] * afm_r[2]),
                   afm_J[1] + afm_m * (afm_r[0] * afm_r[0] + afm_r[2] * afm_r[2]),
                   afm_J[2] + afm_m * (afm_r[0] * afm_r[0] + afm_r[1] * afm_r[1]),
                   -afm_m * (afm_r[0] * afm_r[1]), -afm_m * (afm_r[0] * afm_r[2]), -afm_m * (afm_r[1] * afm_r[2])};
    {
        const double m_half = 0.5 * fmax(0.0, m_fuel_total);
        add_point(m_half, fuel_r[0], M, Mr, J);
        add_point(m_half, fuel_r[1], M, Mr, J);
#pragma unroll
        for (int k = 0; k < 5; k++) add_point(in.get_m_pld(k), pld_r[k], M, Mr, J);
    }
    const double iM = 1 / M;
    const v3 r_bc = iM * Mr;  // CoM position in body frame

    FB_PHASE_FENCE();
    // ===== rigid-body dynamics at the CoM (dynamics.jl:443-525) =====
    {
        const v3 w_ie_b = qrot_inv(q_eb, v3{0, 0, wgs::w_ie});
        // inertia about the CoM: J_c = J_b + m skew(r)^2
        const double rr = dot(r_bc, r_bc);
        const double Jxx = J[0] - M * (rr - r_bc.x * r_bc.x), Jyy = J[1] - M * (rr - r_bc.y * r_bc.y), Jzz = J[2] - M * (rr - r_bc.z * r_bc.z);
        const double Jxy = J[3] + M * (r_bc.x * r_bc.y), Jxz = J[4] + M * (r_bc.x * r_bc.z), Jyz = J[5] + M * (r_bc.y * r_bc.z);
        auto Jmul = [&](v3 v) { return v3{Jxx * v.x + Jxy * v.y + Jxz * v.z, Jxy * v.x + Jyy * v.y + Jyz * v.z, Jxz * v.x + Jyz * v.y + Jzz * v.z}; };
        const v3 F_c = F_b;
        const v3 tau_c = tau_b - cross(r_bc, F_b);
        const v3 v_ec_c = v_eb_b + cross(w_eb_b, r_bc);
        const v3 w_ic_c = w_ie_b + w_eb_b;
        // gravity at the CoM, along -n_e(Oc)
        const v3 r_ec_e = r_eb_e + qrot(q_eb, r_bc);
        v3 n_c;
        double h_c;
        geodetic_from_ecef(r_ec_e, n_c, h_c);
        if (!(h_c >= H_MIN)) st |= FB_ST_ALT_RANGE;
        const double g = normal_gravity(n_c.z, h_c);
        // q_el = ltf(Oc) = Rz(λ) ∘ Ry(-(ϕ + π/2)) (geodesy.jl:132-135), q_cl = q_eb' ∘ q_el, g_c = q_cl(0,0,g).
        // The composition order must be the reference's: at RK stages q_eb is not exactly unit, and
        // v + 2 q_im x (q_re v + q_im x v) with a non-unit q does not commute with re-association.
        // λ = atan2(n_y, n_x); θ = -(ϕ + π/2) = atan2(-p, -n_z) with p = |(n_x, n_y)|: half-angle forms, no atan2/sincos
        const quat q_el = ltf_quat(n_c);
        const quat q_cl = qmul(qconj(q_eb), q_el);
        const v3 g_c_c = qrot(q_cl, v3{0.0, 0.0, g});

        const v3 hc = Jmul(w_ic_c) + h_rot;
        const v3 rhs_w = tau_c - Jmul(cross(w_ie_b, w_eb_b)) - cross(w_ic_c, hc);
        // symmetric 3x3 solve by cofactors
        const double c11 = Jyy * Jzz - Jyz * Jyz, c12 = Jyz * Jxz - Jxy * Jzz, c13 = Jxy * Jyz - Jyy * Jxz;
        const double c22 = Jxx * Jzz - Jxz * Jxz, c23 = Jxy * Jxz - Jxx * Jyz, c33 = Jxx * Jyy - Jxy * Jxy;
        const double idet = 1 / (Jxx * c11 + Jxy * c12 + Jxz * c13);
        const v3 wd = {(c11 * rhs_w.x + c12 * rhs_w.y + c13 * rhs_w.z) * idet, (c12 * rhs_w.x + c22 * rhs_w.y + c23 * rhs_w.z) * idet,
                       (c13 * rhs_w.x + c23 * rhs_w.y + c33 * rhs_w.z) * idet};
        const v3 vd_c = iM * F_c + g_c_c - cross(w_eb_b + 2.0 * w_ie_b, v_ec_c);
        const v3 vd_b = vd_c - cross(wd, r_bc);
        emit(FB_X_OMEGA_EB_B, wd.x); emit(FB_X_OMEGA_EB_B + 1, wd.y); emit(FB_X_OMEGA_EB_B + 2, wd.z);
        emit(FB_X_V_EB_B, vd_b.x); emit(FB_X_V_EB_B + 1, vd_b.y); emit(FB_X_V_EB_B + 2, vd_b.z);
        if (WITH_Y) {
            const int k = FB_Y_DYN;
            YP(k, M); YP3(k + 1, r_bc);
            YP(k + 4, J[0]); YP(k + 5, J[3]); YP(k + 6, J[4]); YP(k + 7, J[3]); YP(k + 8, J[1]); YP(k + 9, J[5]); YP(k + 10, J[4]); YP(k + 11, J[5]); YP(k + 12, J[2]);
            YP3(k + 13, F_b); YP3(k + 16, tau_b); YP3(k + 19, wd); YP3(k + 22, vd_b);
            const v3 r_eb_b = qrot_inv(q_eb, r_eb_e), r_ec_c = qrot_inv(q_eb, r_ec_e);
            YP3(k + 25, vd_b + cross(w_eb_b, v_eb_b));
            YP3(k + 28, vd_b + cross(w_eb_b + 2.0 * w_ie_b, v_eb_b) + cross(w_ie_b, cross(w_ie_b, r_eb_b)));
            const v3 a_ic = vd_c + cross(w_eb_b + 2.0 * w_ie_b, v_ec_c) + cross(w_ie_b, cross(w_ie_b, r_ec_c));
            const v3 gam = g_c_c + cross(w_ie_b, cross(w_ie_b, r_ec_c));
            YP3(k + 31, a_ic - gam); YP3(k + 34, wd - cross(w_eb_b, w_ie_b)); YP3(k + 37, g_c_c);
        }
    }
    return st;
}

// f_step!(world): kinematics renormalisation, stall hysteresis, contact-regulator reset, crash checks,
// engine state machine (aircraftbase.jl:172-181; kinematics.jl:226-229,114-118; c172.jl:375-384,715-724;
// landinggear.jl:331-347,479-483; piston.jl:428-453). Returns true when x or s changed.
template <int KIN, class In, int NXT>
FBD bool f_step(double (&x)[NXT], int& stall, int& eng_state, const In& in, const StepAux& aux, int32_t& st) {
    bool mod = false;
    // normalize_block!(x, ε = 1e-8) (kinematics.jl:114-118) on q_wb, q_ew (WA :226-229) or q_eb, n_e (ECEF :317-320); NED: nothing
    auto renorm = [&](int k0, int len) {
        double n2 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < len) n2 += x[k0 + k] * x[k0 + k];
        const double nr = sqrt(n2);
        if (fabs(nr - 1.0) > 1e-8) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (k < len) x[k0 + k] /= nr;
            mod = true;
        }
    };
    if constexpr (KIN == FB_KIN_WA) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_EW, 4); }
    else if constexpr (KIN == FB_KIN_ECEF) { renorm(FB_X_Q_WB, 4); renorm(FB_X_Q_WB + 4, 3); }
    const int stall0 = stall, eng0 = eng_state;
    if (aux.alpha > c172::alpha_stall_hi) stall = 1;
    else if (aux.alpha < c172::alpha_stall_lo) stall = 0;
    if (aux.crash) st |= FB_ST_GROUND_CRASH;
#pragma unroll
    for (int g = 0; g < 3; g++) {
        if (!(aux.wow & (1 << g))) {
            if (x[FB_X_LDG_FRC + 2 * g] != 0.0 || x[FB_X_LDG_FRC + 2 * g + 1] != 0.0) mod = true;
            x[FB_X_LDG_FRC + 2 * g] = 0.0;
            x[FB_X_LDG_FRC + 2 * g + 1] = 0.0;
        }
    }
    const double w = x[FB_X_ENG_OMEGA];
    const bool fuel = aux.m_avail > 0;
    const bool start = in.ui & FB_UI_ENG_START, stop = in.ui & FB_UI_ENG_STOP;
    if (eng_state == 0) {
        if (start) eng_state = 1;
    } else if (eng_state == 1) {
        if (!start) eng_state = 0;
        if (w > c172::w_idle && fuel) eng_state = 2;
    } else {
        if (stop || w < c172::w_stall || !fuel) eng_state = 0;
    }
    return mod || stall != stall0 || eng_state != eng0;
}

}  // namespace fbd
