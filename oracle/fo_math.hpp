// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's quaternion / attitude primitives.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
//
// Follows (operation order included):
//   lib/FlightPhysics/src/quaternions.jl:65-78,109-115
//   lib/FlightPhysics/src/attitude.jl:93-103,118,129,175-233,288-308,382-395,436-478
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>

namespace fo {

constexpr double PI = 3.14159265358979323846;
constexpr double HALF_PI = PI / 2;
inline constexpr double deg2rad(double d) { return d * (PI / 180); }  // Base.deg2rad: z * (π/180)
inline constexpr double rad2deg(double r) { return r * (180 / PI); }

struct V3 {
    double x = 0, y = 0, z = 0;
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalize(V3 a) { return a / norm(a); }

// 3x3 matrix, row-major m[r][c]
struct M3 {
    double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
};
inline M3 operator+(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}
inline M3 operator-(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] - b.m[i][j];
    return r;
}
inline M3 operator*(double s, const M3& a) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = s * a.m[i][j];
    return r;
}
inline M3 operator*(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
inline V3 operator*(const M3& a, V3 v) {
    return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
            a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline M3 transpose(const M3& a) {
    M3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
    return r;
}
inline M3 diag3(double a, double b, double c) {
    M3 r;
    r.m[0][0] = a; r.m[1][1] = b; r.m[2][2] = c;
    return r;
}
// attitude.jl:43-51
inline M3 v2skew(V3 v) {
    M3 r;
    r.m[0][1] = -v.z; r.m[0][2] = v.y;
    r.m[1][0] = v.z;  r.m[1][2] = -v.x;
    r.m[2][0] = -v.y; r.m[2][1] = v.x;
    return r;
}
// solve A x = b for 3x3 (StaticArrays' closed-form 3x3 solve; dynamics.jl:492 `J \ v`)
inline V3 solve3(const M3& A, V3 b) {
    const double a11 = A.m[0][0], a12 = A.m[0][1], a13 = A.m[0][2];
    const double a21 = A.m[1][0], a22 = A.m[1][1], a23 = A.m[1][2];
    const double a31 = A.m[2][0], a32 = A.m[2][1], a33 = A.m[2][2];
    const double c11 = a22 * a33 - a23 * a32;
    const double c12 = a23 * a31 - a21 * a33;
    const double c13 = a21 * a32 - a22 * a31;
    const double det = a11 * c11 + a12 * c12 + a13 * c13;
    const double idet = 1.0 / det;
    V3 x;
    x.x = (c11 * b.x + (a13 * a32 - a12 * a33) * b.y + (a12 * a23 - a13 * a22) * b.z) * idet;
    x.y = (c12 * b.x + (a11 * a33 - a13 * a31) * b.y + (a13 * a21 - a11 * a23) * b.z) * idet;
    x.z = (c13 * b.x + (a12 * a31 - a11 * a32) * b.y + (a11 * a22 - a12 * a21) * b.z) * idet;
    return x;
}

// ---------------------------------------------------------------------------------------------
// Quaternion (quaternions.jl). q = (re, im)
struct Quat {
    double w = 1, x = 0, y = 0, z = 0;
    V3 im() const { return {x, y, z}; }
    double operator[](int i) const { return i == 0 ? w : (i == 1 ? x : (i == 2 ? y : z)); }
};
inline Quat make_quat(double re, V3 im) { return {re, im.x, im.y, im.z}; }
// quaternions.jl:109-115 (Hamilton product; no normalisation)
inline Quat qmul(Quat q1, Quat q2) {
    const double p_re = q1.w * q2.w - dot(q1.im(), q2.im());
    const V3 p_im = q1.w * q2.im() + q2.w * q1.im() + cross(q1.im(), q2.im());
    return make_quat(p_re, p_im);
}
inline Quat qconj(Quat q) { return {q.w, -q.x, -q.y, -q.z}; }                       // :74-78
inline double qnorm(Quat q) { return std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); }  // :65
inline Quat qnormalize(Quat q) {
    const double n = qnorm(q);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
inline Quat qneg(Quat q) { return {-q.w, -q.x, -q.y, -q.z}; }
inline bool qeq(Quat a, Quat b) { return a.w == b.w && a.x == b.x && a.y == b.y && a.z == b.z; }

// RQuat operations (attitude.jl)
inline Quat compose(Quat a, Quat b) { return qmul(a, b); }           // :93
inline Quat inv(Quat q) { return qconj(q); }                          // :95
// strict equality with double cover, attitude.jl:89
inline bool rq_equal(Quat a, Quat b) { return qeq(a, b) || qeq(a, qneg(b)); }
// attitude.jl:98-103:  v_a = v_b + 2q_im × (q_re * v_b + q_im × v_b)
inline V3 rotate(Quat q, V3 v_b) {
    const V3 q_im = q.im();
    return v_b + cross(2.0 * q_im, q.w * v_b + cross(q_im, v_b));
}
// attitude.jl:118: dt(r_ab, ω_ab_b) = 0.5 * (r_ab * FreeQuat(imag = ω))
inline Quat qdot(Quat q, V3 w) {
    const Quat p = qmul(q, make_quat(0.0, w));
    return {0.5 * p.w, 0.5 * p.x, 0.5 * p.y, 0.5 * p.z};
}
// attitude.jl:288-290, 304-308: axis-angle to quaternion
inline Quat Rx(double a) { return {std::cos(0.5 * a), 1.0 * std::sin(0.5 * a), 0.0 * std::sin(0.5 * a), 0.0 * std::sin(0.5 * a)}; }
inline Quat Ry(double a) { return {std::cos(0.5 * a), 0.0 * std::sin(0.5 * a), 1.0 * std::sin(0.5 * a), 0.0 * std::sin(0.5 * a)}; }
inline Quat Rz(double a) { return {std::cos(0.5 * a), 0.0 * std::sin(0.5 * a), 0.0 * std::sin(0.5 * a), 1.0 * std::sin(0.5 * a)}; }

struct Euler { double psi = 0, theta = 0, phi = 0; };
// attitude.jl:382-391
inline Euler euler_from_quat(Quat q) {
    const double q1 = q.w, q2 = q.x, q3 = q.y, q4 = q.z;
    const double s1 = q1 * q1, s2 = q2 * q2, s3 = q3 * q3, s4 = q4 * q4;
    (void)s1;
    Euler e;
    e.psi = std::atan2(2 * (q1 * q4 + q2 * q3), 1 - 2 * (s3 + s4));
    e.theta = std::asin(std::clamp(2 * (q1 * q3 - q2 * q4), -1.0, 1.0));
    e.phi = std::atan2(2 * (q1 * q2 + q3 * q4), 1 - 2 * (s2 + s3));
    return e;
}
// attitude.jl:393-395
inline Quat quat_from_euler(Euler e) { return compose(compose(Rz(e.psi), Ry(e.theta)), Rx(e.phi)); }

// attitude.jl:175-190 (normalises first)
inline M3 rmatrix_from_quat(Quat r) {
    const Quat q = qnormalize(r);
    const double q1 = q.w, q2 = q.x, q3 = q.y, q4 = q.z;
    const double s2 = q2 * q2, s3 = q3 * q3, s4 = q4 * q4;
    const double dq12 = 2 * q1 * q2, dq13 = 2 * q1 * q3, dq14 = 2 * q1 * q4;
    const double dq23 = 2 * q2 * q3, dq24 = 2 * q2 * q4, dq34 = 2 * q3 * q4;
    M3 M;
    M.m[0][0] = 1 - 2 * (s3 + s4); M.m[0][1] = dq23 - dq14;       M.m[0][2] = dq24 + dq13;
    M.m[1][0] = dq23 + dq14;       M.m[1][1] = 1 - 2 * (s2 + s4); M.m[1][2] = dq34 - dq12;
    M.m[2][0] = dq24 - dq13;       M.m[2][1] = dq34 + dq12;       M.m[2][2] = 1 - 2 * (s2 + s3);
    return M;
}
// attitude.jl:192-233
inline Quat quat_from_rmatrix(const M3& R) {
    const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
    const double cand[4] = {tr, R.m[0][0], R.m[1][1], R.m[2][2]};
    int imax = 0;
    for (int i = 1; i < 4; i++) if (cand[i] > cand[imax]) imax = i;  // findmax: first maximum
    Quat v;
    if (imax == 0) {
        v = {1 + tr, R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]};
    } else if (imax == 1) {
        v = {R.m[2][1] - R.m[1][2], 1 + 2 * R.m[0][0] - tr, R.m[0][1] + R.m[1][0], R.m[2][0] + R.m[0][2]};
    } else if (imax == 2) {
        v = {R.m[0][2] - R.m[2][0], R.m[0][1] + R.m[1][0], 1 + 2 * R.m[1][1] - tr, R.m[1][2] + R.m[2][1]};
    } else {
        v = {R.m[1][0] - R.m[0][1], R.m[2][0] + R.m[0][2], R.m[1][2] + R.m[2][1], 1 + 2 * R.m[2][2] - tr};
    }
    return qnormalize(v);
}

// attitude.jl:436-449 : Euler angle rates from body rates
inline V3 euler_dot(Euler e, V3 w) {
    const double sphi = std::sin(e.phi), cphi = std::cos(e.phi);
    const double tth = std::tan(e.theta), sec = 1.0 / std::cos(e.theta);
    return {sphi * sec * w.y + cphi * sec * w.z, cphi * w.y - sphi * w.z, w.x + sphi * tth * w.y + cphi * tth * w.z};
}
// attitude.jl:460-474 : body rates from Euler angle rates (ė = ψ̇, θ̇, φ̇)
inline V3 omega_from_euler_dot(Euler e, V3 ed) {
    const double sth = std::sin(e.theta), cth = std::cos(e.theta);
    const double sphi = std::sin(e.phi), cphi = std::cos(e.phi);
    return {-sth * ed.x + ed.z, cth * sphi * ed.x + cphi * ed.y, cth * cphi * ed.x - sphi * ed.y};
}
inline double azimuth(V3 v) { return std::atan2(v.y, v.x); }                                    // :476
inline double inclination(V3 v) { return std::atan2(-v.z, std::sqrt(v.x * v.x + v.y * v.y)); }  // :477
inline double wrap_to_pi(double x) { return x + 2 * PI * std::floor((PI - x) / (2 * PI)); }     // :478

// ---------------------------------------------------------------------------------------------
// Interpolations.jl equivalents (third-party; restated from its documented algorithm, v0.16).
// Gridded(Linear()): locate interval with searchsortedlast clamped to [1, n-1]; w = (x-k_i)/(k_{i+1}-k_i).
enum Extrap { FLAT = 0, LINE = 1 };

struct GridLoc { int i; double w; };
inline GridLoc grid_locate(const double* k, int n, double x, Extrap lo, Extrap hi) {
    if (x < k[0] && lo == FLAT) x = k[0];
    if (x > k[n - 1] && hi == FLAT) x = k[n - 1];
    int i = 0;  // 0-based index of lower knot: last knot <= x, clamped to [0, n-2]
    while (i < n - 2 && k[i + 1] <= x) i++;
    return {i, (x - k[i]) / (k[i + 1] - k[i])};
}
// uniform range knots (scale(interpolate(A, BSpline(Linear())), range)): index coordinate
inline GridLoc range_locate(double a, double b, int n, double x, Extrap lo, Extrap hi) {
    if (x < a && lo == FLAT) x = a;
    if (x > b && hi == FLAT) x = b;
    const double step = (b - a) / (n - 1);
    const double xi = (x - a) / step;  // 0-based continuous index
    int i = (int)std::floor(xi);
    i = std::clamp(i, 0, n - 2);
    return {i, xi - i};
}
inline double lerp1(const double* A, GridLoc l) { return (1 - l.w) * A[l.i] + l.w * A[l.i + 1]; }
// A is column-major [n1 x n2] as in Julia: A[i + n1*j]
inline double lerp2(const double* A, int n1, GridLoc l1, GridLoc l2) {
    const double a00 = A[l1.i + n1 * l2.i], a10 = A[l1.i + 1 + n1 * l2.i];
    const double a01 = A[l1.i + n1 * (l2.i + 1)], a11 = A[l1.i + 1 + n1 * (l2.i + 1)];
    return (1 - l1.w) * ((1 - l2.w) * a00 + l2.w * a01) + l1.w * ((1 - l2.w) * a10 + l2.w * a11);
}

}  // namespace fo
