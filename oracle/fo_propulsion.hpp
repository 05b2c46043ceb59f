// TEST INFRASTRUCTURE — CPU oracle, not product code.
// Scalar fp64 restatement of Flight.jl's propeller (blade-element table generator + run-time
// wrench) and Lycoming IO-360 piston engine model.
// Follows lib/FlightPhysics/src/propellers.jl:50-107,131-208,235-291,405-452
//         lib/FlightPhysics/src/piston.jl:25-41,70-195,290-312,314-477,575-603
// Third-party pieces restated from their documented behaviour:
//   Roots.find_zero(f, x0) (Roots.jl 3) -> secant iteration from x0 with bracketing fallback, to
//   machine tolerance; Trapz.trapz (2.0.3) -> composite trapezoid rule.
#pragma once
#include "fo_dynamics.hpp"
#include <functional>
#include <stdexcept>
#include <vector>

namespace fo {

// =============================================================================================
// Propeller blade + airfoil (propellers.jl:24-107)
struct Blade {
    double zeta_h = 0.2;     // hub/blade diameter ratio
    double c_a = 0.075;      // EllipticDistribution(0.075): chord/diameter = a*sqrt(1-ζ²)
    double p_a = 0.8;        // ConstantDistribution(0.8): pitch/diameter
    double chord(double z) const { return c_a * std::sqrt(1 - z * z); }
    double pitch(double) const { return p_a; }
};
inline double airfoil_alpha0() { return deg2rad(-2.1); }  // :48
// propellers.jl:50-58
inline double airfoil_cL(double al, double M) {
    if (M <= 0.8) return (al < 0.25 ? 2 * PI * al : PI / 2 * std::cos(al) / std::cos(0.25)) / std::sqrt(1 - M * M);
    if (M >= 1.2) return (al < 0.25 ? 4 * al : std::cos(al) / std::cos(0.25)) / std::sqrt(M * M - 1);
    return airfoil_cL(al, 0.8) + (airfoil_cL(al, 1.2) - airfoil_cL(al, 0.8)) / 0.4 * (M - 0.8);
}
// propellers.jl:60-68
inline double airfoil_cL_alpha(double al, double M) {
    if (M <= 0.8) return (al < 0.25 ? 2 * PI : -PI / 2 * std::sin(al) / std::cos(0.25)) / std::sqrt(1 - M * M);
    if (M >= 1.2) return (al < 0.25 ? 4.0 : -std::sin(al) / std::cos(0.25)) / std::sqrt(M * M - 1);
    return airfoil_cL_alpha(al, 0.8) + (airfoil_cL_alpha(al, 1.2) - airfoil_cL_alpha(al, 0.8)) / 0.4 * (M - 0.8);
}
// propellers.jl:70-94
inline double airfoil_cD(double al, double M) {
    double cD_inc;
    if (al < 0.25) cD_inc = 0.006 + 0.224 * (al * al);
    else if (al < 0.3) cD_inc = -1.0234 + 16.6944 * (al * al);
    else cD_inc = PI / 2 * std::sin(al) / std::cos(0.25);
    double k_dd;
    if (M <= 0.8) k_dd = 1.0;
    else if (M <= 0.95) k_dd = 1.0 + 160000 * std::pow(M - 0.8, 4) / 27;
    else if (M <= 1.0) k_dd = 6.0 - 800 * ((1 - M) * (1 - M));
    else k_dd = 6 - 5 * (M - 1);
    return k_dd * cD_inc;
}
inline double blade_beta_c(const Blade& b, double z, double dbeta) { return std::atan(b.pitch(z) / (PI * z)) + dbeta; }  // :104
inline double blade_beta_a(const Blade& b, double z, double dbeta) { return blade_beta_c(b, z, dbeta) - airfoil_alpha0(); }  // :107

// propellers.jl:198-202
inline double M_section(double J, double Mt, double z, double eps_i) {
    const double pi2 = PI * PI;
    return Mt * std::sqrt((pi2 * (z * z) + J * J) / (pi2 + J * J)) * std::cos(eps_i);
}
// propellers.jl:204-208
inline double induced_angle_eq(int n_blades, double c, double beta_a_t, double J, double Mt, double beta_a,
                               double eps_inf, double z, double eps_i) {
    const double al = beta_a - eps_inf - eps_i;
    const double M = M_section(J, Mt, z, eps_i);
    return n_blades * c / (8 * z) * airfoil_cL(al, M) -
           std::acos(std::exp(-n_blades * (1 - z) / (2 * std::sin(beta_a_t)))) * std::tan(eps_i) * std::sin(eps_inf + eps_i);
}

// Root finder standing in for Roots.find_zero(f, x0) (Order0: secant steps, switch to bracketing
// once a sign change is seen). Converges to the root nearest the secant path from x0.
inline double find_zero(const std::function<double(double)>& f, double x0) {
    double a = x0, fa = f(a);
    if (fa == 0.0) return a;
    double h = std::fabs(x0) * 1e-4 + 1e-6;  // second secant point
    double b = x0 + h, fb = f(b);
    if (fb == 0.0) return b;
    bool bracket = (fa < 0) != (fb < 0);
    double lo = a, flo = fa, hi = b, fhi = fb;
    for (int it = 0; it < 200 && !bracket; it++) {
        // secant step
        double c = b - fb * (b - a) / (fb - fa);
        if (!std::isfinite(c)) c = b + (b - a);
        // keep the step bounded (tan() poles at ±π/2)
        const double maxstep = 0.5;
        if (c - b > maxstep) c = b + maxstep;
        if (b - c > maxstep) c = b - maxstep;
        const double fc = f(c);
        if (fc == 0.0) return c;
        if ((fc < 0) != (fb < 0)) { lo = b; flo = fb; hi = c; fhi = fc; bracket = true; break; }
        if (std::fabs(c - b) <= 4e-16 * std::fabs(c)) return c;
        a = b; fa = fb; b = c; fb = fc;
    }
    if (!bracket) throw std::runtime_error("find_zero: no convergence");
    // bracketed: secant/bisection safeguarded (regula falsi w/ Illinois + bisection), to machine tolerance
    if (lo > hi) { std::swap(lo, hi); std::swap(flo, fhi); }
    for (int it = 0; it < 200; it++) {
        double c = hi - fhi * (hi - lo) / (fhi - flo);
        if (!(c > lo && c < hi) || (it % 3 == 2)) c = 0.5 * (lo + hi);
        const double fc = f(c);
        if (fc == 0.0) return c;
        if ((fc < 0) == (flo < 0)) { lo = c; flo = fc; } else { hi = c; fhi = fc; }
        if (hi - lo <= 2e-16 * std::max(std::fabs(lo), std::fabs(hi))) break;
    }
    return std::fabs(flo) < std::fabs(fhi) ? lo : hi;
}

struct PropCoeffs { double C_Fx = 0, C_Mx = 0, C_Fz_a = 0, C_Mz_a = 0, C_P = 0, eta_p = 0; };

inline double trapz(const std::vector<double>& x, const std::vector<double>& y) {
    double s = 0;
    for (size_t i = 1; i < x.size(); i++) s += (x[i] - x[i - 1]) * (y[i] + y[i - 1]);
    return 0.5 * s;
}
// propellers.jl:131-196
inline PropCoeffs prop_coefficients(int n_blades, const Blade& blade, double J, double Mt, double dbeta, int n_z = 101) {
    const double pi2 = PI * PI;
    std::vector<double> zs(n_z), dFx(n_z), dMx(n_z), dFz(n_z), dMz(n_z);
    const double beta_a_t = blade_beta_a(blade, 1.0, dbeta);
    double eps_i = 1;
    for (int i = 0; i < n_z; i++) {
        // range(ζ_h, 1, length = n_ζ): first and last points exact
        const double z = (i == n_z - 1) ? 1.0 : blade.zeta_h + i * ((1.0 - blade.zeta_h) / (n_z - 1));
        zs[i] = z;
        const double eps_inf = std::atan(J / (PI * z));
        const double beta_a = blade_beta_a(blade, z, dbeta);
        const double c = blade.chord(z);
        auto f = [&](double e) { return induced_angle_eq(n_blades, c, beta_a_t, J, Mt, beta_a, eps_inf, z, e); };
        eps_i = find_zero(f, eps_i);
        const double eps = eps_inf + eps_i;
        const double al = beta_a - eps;
        const double M = M_section(J, Mt, z, eps_i);
        const double kc = n_blades * c;
        const double z2 = z * z, z3 = z * z * z;
        const double cos_e = std::cos(eps), sin_e = std::sin(eps);
        const double c2ei = std::cos(eps_i) * std::cos(eps_i), c2einf = std::cos(eps_inf) * std::cos(eps_inf);
        const double t_inf = std::tan(eps_inf), t2_inf = t_inf * t_inf;
        const double cL = airfoil_cL(al, M), cD = airfoil_cD(al, M), cLa = airfoil_cL_alpha(al, M);
        dFx[i] = pi2 / 4 * z2 * kc * c2ei / c2einf * (cL * cos_e - cD * sin_e);
        dMx[i] = -pi2 / 8 * z3 * kc * c2ei / c2einf * (cD * cos_e + cL * sin_e);
        dFz[i] = -pi2 / 8 * z2 * kc * c2ei * (2 * t_inf * (cD * cos_e + cL * sin_e) - t2_inf * (cL * cos_e - (cLa + cD) * sin_e));
        dMz[i] = -pi2 / 16 * z3 * kc * c2ei * (2 * t_inf * (cL * cos_e - cD * sin_e) + t2_inf * ((cLa + cD) * cos_e + cL * sin_e));
    }
    PropCoeffs r;
    r.C_Fx = trapz(zs, dFx);
    r.C_Mx = trapz(zs, dMx);
    r.C_Fz_a = trapz(zs, dFz);
    r.C_Mz_a = trapz(zs, dMz);
    r.C_P = 2 * PI * r.C_Mx;
    r.eta_p = r.C_Fx > 0 ? -J * r.C_Fx / r.C_P : 0.0;
    return r;
}

// propellers.jl:235-291 : fixed-pitch lookup, J,Mt ∈ range(0,1.5,21), Flat extrapolation
struct PropLookup {
    static constexpr int NJ = 21, NM = 21;
    double J_lo = 0, J_hi = 1.5, Mt_lo = 0, Mt_hi = 1.5;
    std::vector<double> data[6];  // C_Fx, C_Mx, C_Fz_α, C_Mz_α, C_P, η_p ; column-major [J, Mt]
    void build(int n_blades = 2, const Blade& blade = Blade{}) {
        for (auto& d : data) d.assign(NJ * NM, 0.0);
        for (int j = 0; j < NM; j++)
            for (int i = 0; i < NJ; i++) {
                const double J = (i == NJ - 1) ? J_hi : J_lo + i * ((J_hi - J_lo) / (NJ - 1));
                const double Mt = (j == NM - 1) ? Mt_hi : Mt_lo + j * ((Mt_hi - Mt_lo) / (NM - 1));
                const PropCoeffs c = prop_coefficients(n_blades, blade, J, Mt, 0.0);
                const int k = i + NJ * j;
                data[0][k] = c.C_Fx; data[1][k] = c.C_Mx; data[2][k] = c.C_Fz_a;
                data[3][k] = c.C_Mz_a; data[4][k] = c.C_P; data[5][k] = c.eta_p;
            }
    }
    PropCoeffs eval(double J, double Mt) const {
        const GridLoc lj = range_locate(J_lo, J_hi, NJ, J, FLAT, FLAT);
        const GridLoc lm = range_locate(Mt_lo, Mt_hi, NM, Mt, FLAT, FLAT);
        PropCoeffs c;
        c.C_Fx = lerp2(data[0].data(), NJ, lj, lm);
        c.C_Mx = lerp2(data[1].data(), NJ, lj, lm);
        c.C_Fz_a = lerp2(data[2].data(), NJ, lj, lm);
        c.C_Mz_a = lerp2(data[3].data(), NJ, lj, lm);
        c.C_P = lerp2(data[4].data(), NJ, lj, lm);
        c.eta_p = lerp2(data[5].data(), NJ, lj, lm);
        return c;
    }
};

struct PropParams {        // c172s.jl:28-30 defaults for the C172S power plant
    int sense = 1;         // CW = 1, CCW = -1
    double d = 2.0;
    double J_xx = 0.3;
    FrameTransform t_bp;
};
struct PropY {
    V3 v_wOp_p;
    double omega = 0, J = 0, Mt = 0;
    Wrench wr_p, wr_b;
    V3 hr_p, hr_b;
    double P = 0, eta_p = 0;
};
// propellers.jl:405-452
inline void propeller_f_ode(const PropParams& prm, const PropLookup& lookup, const KinData& kin,
                            const AirData& air, double omega, PropY& y) {
    const V3 v_wOp_b = air.v_wb_b + cross(kin.w_eb_b, prm.t_bp.r);
    const V3 v_wOp_p = rotate(inv(prm.t_bp.q), v_wOp_b);
    const double abs_w_min = 1.0;
    const double v_J = norm(v_wOp_p);
    const double w_J = std::max(std::fabs(omega), abs_w_min);
    const double J = 2 * PI * v_J / (w_J * prm.d);
    const double Mt = std::fabs(omega) * (prm.d / 2) / air.a;
    const PropCoeffs c = lookup.eval(J, Mt);
    const double C_Fy_b = c.C_Fz_a, C_My_b = c.C_Mz_a;
    double a_p, b_p;
    airflow_angles(v_wOp_p, a_p, b_p);
    const V3 C_F = {c.C_Fx, C_Fy_b * b_p, c.C_Fz_a * a_p};
    const V3 C_M = (double)prm.sense * V3{c.C_Mx, C_My_b * b_p, c.C_Mz_a * a_p};
    const double rho = air.rho;
    const double f = omega / (2 * PI), f2 = f * f, f3 = f * f2;
    const double d4 = prm.d * prm.d * prm.d * prm.d, d5 = prm.d * d4;
    y.wr_p.F = (rho * f2 * d4) * C_F;
    y.wr_p.tau = (rho * f2 * d5) * C_M;
    y.P = rho * std::fabs(f3) * d5 * c.C_P;
    y.wr_b = translate(prm.t_bp, y.wr_p);
    y.hr_p = {prm.J_xx * omega, 0, 0};
    y.hr_b = rotate(prm.t_bp.q, y.hr_p);
    y.v_wOp_p = v_wOp_p; y.omega = omega; y.J = J; y.Mt = Mt; y.eta_p = c.eta_p;
}

// =============================================================================================
// Piston engine (piston.jl)
namespace pst {
constexpr double beta = -6.5e-3;            // ISA_layers[1].β
constexpr double f_cutoff = 0.0580, f_lean = 0.0625, f_rich = 0.0950;
inline double inHg2Pa(double p) { return 3386.389 * p; }
inline double ft2m(double h) { return 0.3048 * h; }
inline double hp2W(double P) { return 735.49875 * P; }
inline double RPM2radpersec(double w) { return w * PI / 30; }
inline double T_ISA(double p) { return isa::T_std * std::pow(p / isa::p_std, -beta * isa::R / isa::g_std); }  // :38
inline double p2delta(double p) { return (p / isa::p_std) * std::pow(T_ISA(p) / isa::T_std, -0.5); }          // :41
inline double h2delta(double h) {                                                                               // :44-47
    int32_t st = 0;
    const ISAData d = isa_data(h, ISAData{}, st);
    return d.p / isa::p_std / std::sqrt(d.T / isa::T_std);
}
}  // namespace pst

// 1-D gridded table with per-side extrapolation
struct Table1 {
    std::vector<double> k, v;
    Extrap lo = FLAT, hi = FLAT;
    double operator()(double x) const { return lerp1(v.data(), grid_locate(k.data(), (int)k.size(), x, lo, hi)); }
};
// 2-D gridded table, column-major [n1 x n2]
struct Table2 {
    std::vector<double> k1, k2, v;
    Extrap lo1 = FLAT, hi1 = FLAT, lo2 = FLAT, hi2 = FLAT;
    double operator()(double x1, double x2) const {
        const GridLoc l1 = grid_locate(k1.data(), (int)k1.size(), x1, lo1, hi1);
        const GridLoc l2 = grid_locate(k2.data(), (int)k2.size(), x2, lo2, hi2);
        return lerp2(v.data(), (int)k1.size(), l1, l2);
    }
};
// 2-D uniform (scaled B-spline) table, column-major [n1 x n2], Line extrapolation both axes
struct RangeTable2 {
    double a1, b1, a2, b2;
    int n1, n2;
    std::vector<double> v;
    double operator()(double x1, double x2) const {
        const GridLoc l1 = range_locate(a1, b1, n1, x1, LINE, LINE);
        const GridLoc l2 = range_locate(a2, b2, n2, x2, LINE, LINE);
        return lerp2(v.data(), n1, l1, l2);
    }
};
inline std::vector<double> linrange(double a, double b, int n) {
    std::vector<double> r(n);
    for (int i = 0; i < n; i++) r[i] = (i == n - 1) ? b : a + i * ((b - a) / (n - 1));
    return r;
}

// piston.jl:70-195
struct PistonLookup {
    RangeTable2 delta_wot, mu_wot;
    Table2 pi_std, pi_wot, sfc_pow;
    Table1 pi_ratio, sfc_ratio;

    void build(double n_stall, double n_max) {
        // δ_wot(n, μ): n ∈ range(0.667,1,2), μ ∈ range(0.401,0.936,9); Line
        const double d_data[2][9] = {{0.455, 0.523, 0.587, 0.652, 0.718, 0.781, 0.844, 0.906, 0.965},
                                     {0.464, 0.530, 0.596, 0.662, 0.727, 0.792, 0.855, 0.921, 0.981}};
        delta_wot = {0.667, 1.0, 0.401, 0.936, 2, 9, std::vector<double>(18)};
        for (int i = 0; i < 2; i++) for (int j = 0; j < 9; j++) delta_wot.v[i + 2 * j] = d_data[i][j];

        // μ_wot(n, δ): inverse interpolation of δ_wot per n row, resampled on δ ∈ range(0.441,1,9); Line
        {
            const std::vector<double> n_range = linrange(0.667, 1.0, 2);
            const std::vector<double> d_range = linrange(0.441, 1.0, 9);
            const std::vector<double> mu_knots = linrange(0.401, 0.936, 9);
            mu_wot = {0.667, 1.0, 0.441, 1.0, 2, 9, std::vector<double>(18)};
            for (int i = 0; i < 2; i++) {
                Table1 mu_1D;
                mu_1D.lo = LINE; mu_1D.hi = LINE;
                mu_1D.k.resize(9);
                for (int j = 0; j < 9; j++) mu_1D.k[j] = delta_wot(n_range[i], mu_knots[j]);
                mu_1D.v = mu_knots;
                for (int j = 0; j < 9; j++) mu_wot.v[i + 2 * j] = mu_1D(d_range[j]);
            }
        }
        // π_std(n, μ): 13 x 3, Flat
        {
            const std::vector<double> n_data = {n_stall, 0.667, 0.704, 0.741, 0.778, 0.815, 0.852, 0.889, 0.926, 0.963, 1.000, 1.074, n_max};
            const std::vector<double> mu_data = {0, 0.568, 1.0};
            const double mu_kn[3][13] = {
                {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                {0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568, 0.568},
                {1.000, 0.836, 0.854, 0.874, 0.898, 0.912, 0.939, 0.961, 0.959, 0.958, 0.956, 0.953, 1.000}};
            const double pi_kn[3][13] = {
                {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                {0, 0.270, 0.305, 0.335, 0.360, 0.380, 0.405, 0.428, 0.450, 0.476, 0.498, 0.498, 0},
                {0, 0.489, 0.548, 0.609, 0.680, 0.729, 0.810, 0.880, 0.920, 0.965, 1.000, 0.950, 0}};
            pi_std.k1 = n_data; pi_std.k2 = mu_data; pi_std.v.assign(13 * 3, 0.0);
            for (int i = 0; i < 13; i++) {
                Table1 t;
                t.lo = LINE; t.hi = LINE;
                t.k = {mu_kn[0][i], mu_kn[1][i], mu_kn[2][i]};
                t.v = {pi_kn[0][i], pi_kn[1][i], pi_kn[2][i]};
                for (int j = 0; j < 3; j++) pi_std.v[i + 13 * j] = t(mu_data[j]);
            }
        }
        // π_wot(n, δ): 5 x 3; n Flat/Flat, δ Flat below / Line above
        {
            const std::vector<double> n_data = {n_stall, 0.667, 1.000, 1.074, n_max};
            pi_wot.k1 = n_data; pi_wot.k2 = {0, 0.441, 1};
            pi_wot.lo2 = FLAT; pi_wot.hi2 = LINE;
            pi_wot.v.assign(15, 0.0);
            const double col2[5] = {0, 0.23, 0.409, 0.409, 0};
            for (int i = 0; i < 5; i++) {
                pi_wot.v[i + 5 * 0] = 0;
                pi_wot.v[i + 5 * 1] = col2[i];
                pi_wot.v[i + 5 * 2] = pi_std(n_data[i], mu_wot(n_data[i], 1.0));
            }
        }
        // π_ratio(f), sfc_ratio(f): 11 knots, Flat
        {
            std::vector<double> f_data = {pst::f_cutoff};
            for (double v : linrange(pst::f_lean, pst::f_rich, 10)) f_data.push_back(v);
            pi_ratio.k = f_data;
            pi_ratio.v = {0.000, 0.8600, 0.9492, 0.9776, 0.9933, 1.000, 0.9983, 0.9910, 0.9798, 0.9657, 0.9500};
            sfc_ratio.k = f_data;
            sfc_ratio.v = {5, 0.8700, 0.8524, 0.8818, 0.9261, 0.9839, 1.0510, 1.1279, 1.2135, 1.3163, 1.4280};
        }
        // sfc_pow(n, π): 5 x 8, π knots log-spaced, Line
        {
            sfc_pow.k1 = {2000.0 / 2700, 2200.0 / 2700, 2400.0 / 2700, 2600.0 / 2700, 2700.0 / 2700};
            const std::vector<double> ex = linrange(-1, 0, 8);
            sfc_pow.k2.resize(8);
            for (int j = 0; j < 8; j++) sfc_pow.k2[j] = std::pow(10.0, ex[j]);
            sfc_pow.lo1 = sfc_pow.hi1 = sfc_pow.lo2 = sfc_pow.hi2 = LINE;
            const double s[5][8] = {{1.7671, 1.43728, 1.19992, 1.02909, 0.906153, 0.817674, 0.753997, 0.708169},
                                    {1.83791, 1.49664, 1.25103, 1.07427, 0.947056, 0.855503, 0.789613, 0.742193},
                                    {1.98614, 1.60588, 1.3322, 1.13524, 0.993496, 0.891482, 0.818064, 0.765226},
                                    {2.11663, 1.70062, 1.40123, 1.18576, 1.03069, 0.919083, 0.838765, 0.780961},
                                    {2.33484, 1.85418, 1.50825, 1.2593, 1.08012, 0.951177, 0.858376, 0.791588}};
            sfc_pow.v.assign(40, 0.0);
            for (int i = 0; i < 5; i++) for (int j = 0; j < 8; j++) sfc_pow.v[i + 5 * j] = 1e-7 * s[i][j];
        }
    }
};

// piston.jl:457-477
inline double compute_pi_ISA_pow(const PistonLookup& L, double n, double mu, double delta) {
    const double d_wot = L.delta_wot(n, mu);
    const double p_std = L.pi_std(n, mu);
    const double p_wot = L.pi_wot(n, d_wot);
    double r;
    if (std::fabs(d_wot - 1) < 5e-3) r = p_std;
    else r = p_std + (p_wot - p_std) / (d_wot - 1) * (delta - 1);
    return std::max(r, 0.0);
}

enum EngineState : int { ENG_OFF = 0, ENG_STARTING = 1, ENG_RUNNING = 2 };
enum MixtureControl : int { MIX_MANUAL = 0, MIX_AUTO = 1 };

struct EngineParams {  // piston.jl:220-250 ; c172s.jl:18-26
    double P_rated = pst::hp2W(200);
    double w_rated = pst::RPM2radpersec(2700);
    double w_stall = pst::RPM2radpersec(300);
    double w_max = pst::RPM2radpersec(3100);
    double w_idle = pst::RPM2radpersec(600);
    double tau_start = 40;
    double J = 0.05;
    PIParams idle = {4.0, 2.0, 0.0, 1.0, -0.5, 0.5};  // piston.jl:299-312
    PIParams frc = {5.0, 200.0, 0.0, 1.0, -1.0, 1.0};
};
struct EngineU {
    bool start = false, stop = false;
    double throttle = 0;     // Ranged [0,1]
    int mixture_ctl = MIX_AUTO;
    double mixture = 0.5;    // Ranged [0,1]
    double tau_load = 0, J_load = 0;
};
struct EngineY {
    int state = ENG_OFF;
    double throttle = 0, MAP = 0, mixture = 0, mixture_pos = 0, f = 0, mdot = 0, omega = 0, n = 0;
    double tau_shaft = 0, P_shaft = 0, SFC = 0;
    PIOut idle, frc;
};
// piston.jl:314-426. x_eng = [ω, idle, frc]
inline void engine_f_ode(const EngineParams& prm, const PistonLookup& L, const EngineU& u, int state,
                         const double* x_eng, const AirData& air, double* xdot_eng, EngineY& y) {
    const double throttle = std::clamp(u.throttle, 0.0, 1.0);
    const double mixture = std::clamp(u.mixture, 0.0, 1.0);
    const double w = x_eng[0];

    xdot_eng[2] = pi_f_ode(prm.frc, -w, 0, x_eng[2], y.frc);
    xdot_eng[1] = pi_f_ode(prm.idle, 1 - w / prm.w_idle, 0, x_eng[1], y.idle);

    const double mu_ratio_idle = 0.5 + y.idle.output;
    const double n = w / prm.w_rated;
    const double delta = pst::p2delta(air.p);
    const double mu_wot = L.mu_wot(n, delta);
    const double mu = mu_wot * (mu_ratio_idle + throttle * (1 - mu_ratio_idle));
    const double k_f = 1 / std::sqrt(air.rho / isa::rho_std);

    double mixture_pos;
    if (u.mixture_ctl == MIX_MANUAL) {
        mixture_pos = 0.5 * (mixture + 1);
    } else {
        const double f_target = pst::f_lean + mixture * (pst::f_rich - pst::f_lean);
        mixture_pos = f_target / (k_f * pst::f_rich);
    }

    double MAP, f, tau_shaft, P_shaft, SFC, mdot;
    if (state == ENG_OFF) {
        const double tau_fr_max = 0.01 * prm.P_rated / prm.w_rated;
        const double tau_fr = y.frc.output * tau_fr_max;
        MAP = air.p; f = 0; tau_shaft = tau_fr; P_shaft = 0.0; SFC = 0.0; mdot = 0.0;
    } else if (state == ENG_STARTING) {
        MAP = mu * isa::p_std; f = 0; tau_shaft = prm.tau_start; P_shaft = tau_shaft * w; SFC = 0.0; mdot = 0.0;
    } else {
        const double f_sl = pst::f_rich * mixture_pos;
        f = k_f * f_sl;
        const double pi_ISA_pow = compute_pi_ISA_pow(L, n, mu, delta);
        const double pi_pow = pi_ISA_pow * std::sqrt(pst::T_ISA(air.p) / air.T);
        const double pi_actual = pi_pow * L.pi_ratio(f);
        MAP = mu * isa::p_std;
        P_shaft = prm.P_rated * pi_actual;
        tau_shaft = (w > 0 ? P_shaft / w : 0.0);
        SFC = L.sfc_pow(n, pi_actual) * L.sfc_ratio(f);
        mdot = SFC * P_shaft;
    }
    const double S_tau = tau_shaft + u.tau_load;
    const double S_J = prm.J + u.J_load;
    xdot_eng[0] = S_tau / S_J;

    y.state = state; y.throttle = throttle; y.MAP = MAP; y.mixture = mixture; y.mixture_pos = mixture_pos;
    y.f = f; y.mdot = mdot; y.omega = w; y.n = n; y.tau_shaft = tau_shaft; y.P_shaft = P_shaft; y.SFC = SFC;
}
// piston.jl:428-453
inline int engine_f_step(const EngineParams& prm, const EngineU& u, int state, double w, bool fuel_available) {
    if (state == ENG_OFF) {
        if (u.start) state = ENG_STARTING;
    } else if (state == ENG_STARTING) {
        if (!u.start) state = ENG_OFF;
        if (w > prm.w_idle && fuel_available) state = ENG_RUNNING;
    } else {
        if (u.stop || w < prm.w_stall || !fuel_available) state = ENG_OFF;
    }
    return state;
}

struct ThrusterY { EngineY engine; PropY propeller; };
// piston.jl:575-595
inline void thruster_f_ode(const EngineParams& eprm, const PistonLookup& L, const PropParams& pprm,
                           const PropLookup& PL, double gear_ratio, EngineU& eu, int state, const double* x_eng,
                           const AirData& air, const KinData& kin, double* xdot_eng, ThrusterY& y) {
    const double w_eng = x_eng[0];
    const double w_prop = gear_ratio * w_eng;
    propeller_f_ode(pprm, PL, kin, air, w_prop, y.propeller);
    const double tau_prop = y.propeller.wr_p.tau.x;
    eu.tau_load = gear_ratio * tau_prop;
    eu.J_load = gear_ratio * gear_ratio * pprm.J_xx;
    engine_f_ode(eprm, L, eu, state, x_eng, air, xdot_eng, y.engine);
}

}  // namespace fo
