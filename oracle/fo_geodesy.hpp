// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's WGS-84 geodesy.
// Follows lib/FlightPhysics/src/geodesy.jl:15-35,62-69,97-106,125-147,186-211,218-246,367-428,451-489
#pragma once
#include "fo_math.hpp"
#include <vector>

namespace fo {

// geodesy.jl:15-35
namespace wgs {
constexpr double GM = 3.986005e+14;
constexpr double a = 6378137.0;
constexpr double f = 1 / 298.257223563;
constexpr double w_ie = 7.292115e-05;
constexpr double b = a * (1 - f);
constexpr double e2 = 2 * f - f * f;
constexpr double a2 = a * a;
constexpr double b2 = b * b;
constexpr double m = w_ie * w_ie * a * a * b / GM;
constexpr double g_a = 9.7803253359;
constexpr double g_b = 9.8321849378;
constexpr double k_g = b * g_b / (a * g_a) - 1;
}  // namespace wgs

constexpr double H_MIN = -1000.0;  // geodesy.jl:158

// status bits raised where the reference would throw
enum StatusBits : int32_t {
    ST_OK = 0,
    ST_ALT_RANGE = 1,     // Altitude{D}(h) with h < -1000 (geodesy.jl:218-221)
    ST_ISA_RANGE = 2,     // ISAData above 84852 m (atmosphere.jl:133)
    ST_GROUND_CRASH = 4,  // GroundCrash (landinggear.jl:331-347)
    ST_NAN = 8,
    ST_CONTACT_ASSERT = 16,  // landinggear.jl:321 assertion
};

// The reference THROWS at these sites and the simulation stops there (sim.jl:561-570). The steppers run the restatement in throwing
// mode (ThrowScope) and catch Termination where step!(sim) would be left; the single-call entry points (f_ode, f_step: tests that
// look at outputs next to the status word) keep collecting bits.
struct Termination { int32_t bit; };
inline thread_local bool tl_throw_mode = false;
struct ThrowScope {
    bool prev;
    ThrowScope() : prev(tl_throw_mode) { tl_throw_mode = true; }
    ~ThrowScope() { tl_throw_mode = prev; }
};
inline void raise_status(int32_t& st, int32_t bit) {
    if (tl_throw_mode) throw Termination{bit};
    st |= bit;
}

struct LatLon { double phi = 0, lam = 0; };

// geodesy.jl:62-69 : n-vector straight from q_ew
inline V3 nvector_from_qew(Quat q) {
    const double dq12 = 2 * q.w * q.x, dq13 = 2 * q.w * q.y;
    const double dq24 = 2 * q.x * q.z, dq34 = 2 * q.y * q.z;
    return -V3{dq24 + dq13, dq34 - dq12, 1 - 2 * (q.x * q.x + q.y * q.y)};
}
// geodesy.jl:140-147
inline double psi_nw_from_qew(Quat q) {
    const double dq12 = 2 * q.w * q.x, dq13 = 2 * q.w * q.y;
    const double dq24 = 2 * q.x * q.z, dq34 = 2 * q.y * q.z;
    return std::atan2(-(dq34 + dq12), dq24 - dq13);
}
// geodesy.jl:97-101
inline V3 nvector_from_latlon(LatLon ll) {
    const double c = std::cos(ll.phi);
    return {c * std::cos(ll.lam), c * std::sin(ll.lam), std::sin(ll.phi)};
}
// geodesy.jl:103-106
inline LatLon latlon_from_nvector(V3 n) {
    return {std::atan2(n.z, std::sqrt(n.x * n.x + n.y * n.y)), std::atan2(n.y, n.x)};
}
// geodesy.jl:125-129 : (M, N) = (R_N, R_E)
struct Radii { double M, N; };
inline Radii radii(V3 n_e) {
    const double f_den = std::sqrt(1 - wgs::e2 * n_e.z * n_e.z);
    return {wgs::a * (1 - wgs::e2) / (f_den * f_den * f_den), wgs::a / f_den};
}
// geodesy.jl:132-135
inline Quat ltf(V3 n_e, double psi_nw = 0.0) {
    const LatLon ll = latlon_from_nvector(n_e);
    return compose(compose(Rz(ll.lam), Ry(-(ll.phi + 0.5 * PI))), Rz(psi_nw));
}

// EGM96 geoid grid: 721 x 1441 float32, column-major [phi, lam] (geodesy.jl:186-198)
struct Geoid {
    static constexpr int NPHI = 721, NLAM = 1441;
    const float* data = nullptr;
};
inline Geoid& geoid_table() {
    static Geoid g;
    return g;
}
// geodesy.jl:204-211 + scaled-BSpline linear interpolation with Line() extrapolation
inline double geoid_height(V3 n_e) {
    const Geoid& g = geoid_table();
    const LatLon ll = latlon_from_nvector(n_e);
    double lam = std::fmod(ll.lam + 2 * PI, 2 * PI);
    if (lam < 0) lam += 2 * PI;  // Julia mod: result has the sign of the divisor
    const GridLoc lp = range_locate(-PI / 2, PI / 2, Geoid::NPHI, ll.phi, LINE, LINE);
    const GridLoc ll2 = range_locate(0.0, 2 * PI, Geoid::NLAM, lam, LINE, LINE);
    const float* A = g.data;
    const int n1 = Geoid::NPHI;
    const double a00 = A[lp.i + n1 * ll2.i], a10 = A[lp.i + 1 + n1 * ll2.i];
    const double a01 = A[lp.i + n1 * (ll2.i + 1)], a11 = A[lp.i + 1 + n1 * (ll2.i + 1)];
    return (1 - lp.w) * ((1 - ll2.w) * a00 + ll2.w * a01) + lp.w * ((1 - ll2.w) * a10 + ll2.w * a11);
}
// geodesy.jl:232-237
inline double h_orth_from_ellip(double h_e, V3 n_e) { return h_e - geoid_height(n_e); }
inline double h_ellip_from_orth(double h_o, V3 n_e) { return h_o + geoid_height(n_e); }
inline double h_geop_from_orth(double h_o) { return h_o * wgs::a / (wgs::a + h_o); }
inline double h_orth_from_geop(double h_g) { return h_g * wgs::a / (wgs::a - h_g); }

// geodesy.jl:418-428 : Geographic{NVector,Ellipsoidal} -> ECEF Cartesian
inline V3 cartesian_from_geographic(V3 n_e, double h) {
    const double N = radii(n_e).N;
    return {(N + h) * n_e.x, (N + h) * n_e.y, (N * (1 - wgs::e2) + h) * n_e.z};
}
struct GeoNE { V3 n_e; double h_e; };
// geodesy.jl:367-412 : ECEF -> n-vector + ellipsoidal altitude (Fukushima, one Halley step)
inline GeoNE geographic_from_cartesian(V3 r) {
    using namespace wgs;
    const double x = r.x, y = r.y, z = r.z;
    const double p = std::sqrt(x * x + y * y);
    const double c = a * e2;
    const double ec2 = 1 - e2;
    const double ec = std::sqrt(ec2);
    const double zc = ec * std::fabs(z);
    const double s0 = std::fabs(z);
    const double c0 = ec * p;
    const double a0 = std::sqrt(s0 * s0 + c0 * c0);
    const double a03 = a0 * a0 * a0;
    const double b0 = 1.5 * c * s0 * c0 * ((p * s0 - zc * c0) * a0 - c * s0 * c0);
    const double s1 = (zc * a03 + c * (s0 * s0 * s0)) * a03 - b0 * s0;
    const double c1 = (p * a03 - c * (c0 * c0 * c0)) * a03 - b0 * c0;
    const double cc = ec * c1;
    const double s12 = s1 * s1;
    const double cc2 = cc * cc;
    const double h = (p * cc + s0 * s1 - a * std::sqrt(ec2 * s12 + cc2)) / std::sqrt(s12 + cc2);
    const double sgn = (z > 0) ? 1.0 : ((z < 0) ? -1.0 : 0.0);
    double cos_phi, sin_phi;
    if (s1 < cc) {
        const double abs_tan = s1 / cc;
        cos_phi = 1 / std::sqrt(1 + abs_tan * abs_tan);
        const double abs_sin = abs_tan * cos_phi;
        sin_phi = abs_sin * sgn;
    } else {
        const double abs_cot = cc / s1;
        const double abs_sin = 1 / std::sqrt(1 + abs_cot * abs_cot);
        cos_phi = abs_cot * abs_sin;
        sin_phi = abs_sin * sgn;
    }
    const double cos_lam = p > 0 ? (x / p) : 1;
    const double sin_lam = p > 0 ? (y / p) : 0;
    // NVector(...) constructor normalises (geodesy.jl:47-51, default normalization = true)
    const V3 n = normalize(V3{cos_phi * cos_lam, cos_phi * sin_lam, sin_phi});
    return {n, h};
}
// geodesy.jl:451-467 : Somigliana normal gravity with 2nd-order altitude correction
inline double gravity(V3 n_e, double h) {
    using namespace wgs;
    const double sin2 = n_e.z * n_e.z;
    const double g_0 = g_a * (1 + k_g * sin2) / std::sqrt(1 - e2 * sin2);
    return g_0 * (1 - 2 / a * (1 + f + m - 2 * f * sin2) * h + 3 / a2 * (h * h));
}
// geodesy.jl:481-489
inline V3 G_n(V3 n_e, double h) {
    const Quat q_en = ltf(n_e);
    const V3 w_ie_e = {0, 0, wgs::w_ie};
    const V3 r = cartesian_from_geographic(n_e, h);
    return V3{0, 0, gravity(n_e, h)} + rotate(inv(q_en), cross(w_ie_e, cross(w_ie_e, r)));
}

}  // namespace fo
