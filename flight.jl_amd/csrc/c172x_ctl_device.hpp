// c172x_ctl_device.hpp — the Cessna172X gain-scheduled control laws, one aircraft per lane, run every Δt.
//
// Reference (relative to lib/): FlightApps/src/c172/c172x/control/c172x_ctl.jl:203-447 (ControlLawsLon), :463-519 (its
// f_init!), :814-979 (ControlLawsLat), :1000-1032 (its f_init!), :46-77, :735-741 (which compensators a mode enables);
// FlightPhysics/src/control.jl:161-183 (Integrator), :431-471 (PID), :708-743 (LQR), :950-994 (gain lookups: linear in
// (EAS, h_e), Flat extrapolation); c172x2.jl:27-50 (Avionics: guidance in `direct` mode is a no-op).
// Per-aircraft data lives in global memory, structure-of-arrays: inputs cu [FB_NCU x n], record cs [FB_NCS x n]
// (include/flightbatch.h). The update runs ~2 k instructions once per Δt, against ~9 k per RK4 step. Two callers: x2_periodic
// (c172_kernels.hpp), out of line inside the stepping kernels so that their register allocation is untouched, with the record
// cached in registers; and k_x2_ctl (c172x_kernels.hpp), the fb_f_periodic verb as a kernel of its own. Lanes of one wave may sit
// in different modes — the branches below are the price, paid once per Δt.
#pragma once
#include "c172_device.hpp"

namespace fbd {
#ifndef FB_X2_STAMP
#ifdef FB_STAMP
#define FB_X2_STAMP(k) fb_stamp(k)
#else
#define FB_X2_STAMP(k) do { } while (0)
#endif
#endif

// the ten lookups (te2te tv2te vh2te q2e c2θ v2t ar2ar φβ2ar p2φ χ2φ) inside the FB_TABLE_CTL_GAINS blob: offsets in doubles, and the
// pointer type the control laws read them through — global memory (init kernel) or the workgroup's LDS copy (periodic kernel:
// 46 KB staged once per workgroup; the ~330 corner loads per aircraft then cost LDS, not L2, latency)
// same_grid: all ten lookups sit on ONE (EAS, h) grid (fb_set_table compares the headers; true of the reference's data files:
// 7 x 4 nodes over 25-55 m/s x 50-3050 m). The cell and the weights are then located once per update, from the first lookup's
// header, instead of once per lookup (with its own header fetch: an exposed round trip ahead of every gather).
struct CtlOffsets { int off[10]; int total; int same_grid; };
typedef __attribute__((address_space(3))) const double* ldsd_cptr;
typedef __attribute__((address_space(1))) const double* ctlg_cptr;   // the gains blob through a global-memory pointer (x2_periodic)
typedef __attribute__((address_space(4))) const double* ctlk_cptr;   // ... and its wave-uniform headers through scalar loads
FBD ctlk_cptr ctl_hdr(ctlg_cptr p) { return (ctlk_cptr)(uintptr_t)p; }
template <class P> FBD P ctl_hdr(P p) { return p; }
struct CtlCell { int r00, r10, r01, r11; double wE, wH; };   // the four corner records of the cell and the weights along EAS and h
// linear in (EAS, h_e), Flat extrapolation (FP/control.jl:950-994); H: the lookup's six-double header
template <class H>
FBD CtlCell ctl_cell(const H& lk, double EAS, double h) {
    const int nE = (int)lk[0], nH = (int)lk[1];
    int i0 = 0, j0 = 0, i1 = 0, j1 = 0;
    double wE = 0, wH = 0;
    if (nE > 1) {
        const double xi = (fmin(fmax(EAS, lk[2]), lk[3]) - lk[2]) / ((lk[3] - lk[2]) / (nE - 1));
        i0 = min(max((int)floor(xi), 0), nE - 2); i1 = i0 + 1; wE = xi - i0;
    }
    if (nH > 1) {
        const double xj = (fmin(fmax(h, lk[4]), lk[5]) - lk[4]) / ((lk[5] - lk[4]) / (nH - 1));
        j0 = min(max((int)floor(xj), 0), nH - 2); j1 = j0 + 1; wH = xj - j0;
    }
    return {i0 + nE * j0, i1 + nE * j0, i0 + nE * j1, i1 + nE * j1, wE, wH};
}
template <int REC, class P>
FBD void ctl_interp(P d, const CtlCell& c, double (&out)[REC]) {
    const P a00 = d + c.r00 * REC, a10 = d + c.r10 * REC, a01 = d + c.r01 * REC, a11 = d + c.r11 * REC;
    const double wE = c.wE, wH = c.wH;
#pragma unroll
    for (int k = 0; k < REC; k++) out[k] = (1 - wE) * ((1 - wH) * a00[k] + wH * a01[k]) + wE * ((1 - wH) * a10[k] + wH * a11[k]);
}
template <class P>
struct CtlTabT {
    P base;
    CtlOffsets o;
    CtlCell cell;      // of this update's (EAS, h_e), when o.same_grid (ctl_tab)
    FBD P lk(int k) const { return base + o.off[k]; }
    // the record of lookup `p` (a per-lane choice between lookups of one record size is fine) at (EAS, h)
    template <int REC>
    FBD void lookup(P p, double EAS, double h, double (&out)[REC]) const {
        CtlCell c = cell;
        if (!o.same_grid) c = ctl_cell(p, EAS, h);
        ctl_interp<REC>(p + FB_CTL_GRID_HDR, c, out);
    }
};
template <class P>
FBD CtlTabT<P> ctl_tab(P base, const CtlOffsets& o, double EAS, double h) {
    CtlTabT<P> T = {base, o, {0, 0, 0, 0, 0.0, 0.0}};
    if (o.same_grid) T.cell = ctl_cell(ctl_hdr(base + o.off[0]), EAS, h);
    return T;
}
template <class P>
struct CtlMemT {
    P cu;              // &cu[0 * n + i]  (guidance rewrites references and mode requests)
    P cs;              // &cs[0 * n + i]
    int64_t n;
    FBD auto& U(int k) const { return cu[(int64_t)k * n]; }
    FBD auto& S(int k) const { return cs[(int64_t)k * n]; }
};
typedef CtlMemT<double*> CtlMem;
// The same interface with the whole record (cu and cs) cached in registers: the stepping kernels' in-line update (x2_periodic)
// fetches all 94 rows in ONE burst before guidance and the laws run. At one wave per SIMD every batch of reads from memory at its
// point of use is an exposed HBM round trip (the record of 524 288 aircraft is 394 MB: no cache holds it between updates), and there
// were four of them in a row (guidance inputs, longitudinal inputs, lateral inputs, compensator states): 15.32 -> 15.17 ms per launch
// of 25 updates — a modest gain, because the burst reads all 94 rows where the laws touched ~60 (writing everything back in one burst
// at the end as well costs another 1.0 ms: HBM bandwidth). Reads come from the cache, writes go to both (write-through). The
// per-phase cycle profile of one update (tools/stamp_x2.py, -DFB_STAMP build; profiles/r02_x2_update_stamps.txt): record burst
// 2.3 k, guidance 1.2 k, longitudinal channel 20 k, lateral 7.6 k, call and return 5.4 k cycles.
// Every index the laws pass is a compile-time constant after inlining, so the cache is 94 register pairs, not an array in scratch.
template <class P>
struct CtlMemCachedT {
    P cu, cs;
    int64_t n;
    double* lu;        // the caller's local copy of the cu rows
    double* ls;        // ... and of the cs rows
    struct Ref {
        P g; double* l;
        FBD operator double() const { return *l; }
        FBD const Ref& operator=(double v) const { *l = v; *g = v; return *this; }
        FBD const Ref& operator=(const Ref& o) const { return *this = (double)o; }
    };
    FBD Ref U(int k) const { return {cu + (int64_t)k * n, lu + k}; }
    FBD Ref S(int k) const { return {cs + (int64_t)k * n, ls + k}; }
};
// The same cache for the two HALVES of an update that the wave-specialised stepper runs side by side (x2_periodic_lon on role P's wave,
// x2_periodic_lat on role D's, c172_kernels.hpp): the longitudinal and the lateral laws (c172x_ctl.jl:286-446 / :880-983) share nothing but
// the tapped vehicle outputs and the guidance's references, so each half runs the guidance for itself — a few hundred instructions, in
// parallel — and keeps the output it needs (h_ref + the EAS_alt request / chi_ref + the chi_beta request) in its own cache; to MEMORY each
// row is written by exactly one half: the guidance's record rows and its longitudinal outputs by the longitudinal half, its lateral
// outputs by the lateral one. Row indices are compile-time constants after inlining, so `wr` folds.
enum { CTL_HALF_LON = 1, CTL_HALF_LAT = 2 };
// Registers: a half runs on a wave that shares its SIMD with its partner, i.e. in 256 registers — half of what x2_periodic has, whose cache of
// all 94 rows is 188 of them. Only the INPUT rows (cu: a dozen per half, read-only but for the guidance's two outputs) are cached here; the
// record's rows (cs) are read and written in memory where the laws use them. (With both cached the halves spilled ~50 values each to scratch
// memory and every reload was an exposed ~1 k-cycle round trip: 75 k cycles for the longitudinal half, profiles/r04_x2_half_stamps.txt.)
template <class P, int HALF>
struct CtlMemHalfT {
    P cu, cs;
    int64_t n;
    double* lu;        // the caller's local copy of the cu rows
    static constexpr bool writes_u(int k) { return HALF == CTL_HALF_LON ? !(k == FB_CU_CHI_REF || k == FB_CU_LAT_MODE_REQ) : !(k == FB_CU_H_REF || k == FB_CU_LON_MODE_REQ); }
    static constexpr bool writes_s(int k) { return HALF == CTL_HALF_LON ? true : k < FB_CS_GDC_MODE; }
    struct RefU {
        P g; double* l; bool wr;
        FBD operator double() const { return *l; }
        FBD const RefU& operator=(double v) const { *l = v; if (wr) *g = v; return *this; }
        FBD const RefU& operator=(const RefU& o) const { return *this = (double)o; }
    };
    struct RefS {   // (a row this half does not write to memory is one it never reads back: the guidance's record rows in the lateral half)
        P g; bool wr;
        FBD operator double() const { return *g; }
        FBD const RefS& operator=(double v) const { if (wr) *g = v; return *this; }
        FBD const RefS& operator=(const RefS& o) const { return *this = (double)o; }
    };
    FBD RefU U(int k) const { return {cu + (int64_t)k * n, lu + k, writes_u(k)}; }
    FBD RefS S(int k) const { return {cs + (int64_t)k * n, writes_s(k)}; }
};
static_assert(FB_CS_GDC_MODE > FB_CS_CHI2PHI_PID + 2 && FB_CS_SEG_S_2B == FB_NCS - 1, "the guidance's record rows are the last rows of cs");
// what the control laws read from vehicle.y (XLonRed/XLonFull/XLatRed, Zte/Ztv/Zvh/Zφβ/Zar: c172x_ctl.jl:84-199, 745-810)
struct CtlIn {
    double EAS, h_e, theta, phi, clm, chi, lat, lon;
    v3 w_wb_b, w_eb_b;
    double alpha, beta, alpha_filt, beta_filt, n_eng;
    double pos[4];  // throttle, aileron, elevator, rudder positions (Ranged)
    double cmd[4];  // idem, the commands the last f_ode! saw
    bool on_gnd;
};
// the few output-record components the control laws need, tapped from rhs()
struct CtlSink {
    static constexpr bool enabled = true, full = false;
    double theta, phi, wx, wy, wz, vd, chi, EAS, alpha, beta, lat, lon;
    FBD void put(int k, double v) {
        if (k == FB_Y_KIN + 1) theta = v;
        else if (k == FB_Y_KIN + 2) phi = v;
        else if (k == FB_Y_KIN + 15) lat = v;
        else if (k == FB_Y_KIN + 16) lon = v;
        else if (k == FB_Y_KIN + 25) wx = v;
        else if (k == FB_Y_KIN + 26) wy = v;
        else if (k == FB_Y_KIN + 27) wz = v;
        else if (k == FB_Y_KIN + 36) vd = v;
        else if (k == FB_Y_KIN + 38) chi = v;
        else if (k == FB_Y_AIR + 20) EAS = v;
        else if (k == FB_Y_AERO) alpha = v;
        else if (k == FB_Y_AERO + 1) beta = v;
    }
};

// The same sink with a run-time switch (wave-uniform): the one-wave stepping kernels evaluate every f_ode! through ONE instance of rhs() and switch the tap
// on for the step's last evaluation. Up to round 5 that evaluation ran a SECOND inlined instance of rhs() (partial sink) next to the plain one: the
// ground-capable Cessna172Xv2 kernel was ~90 KB of code, more than the 64 KB instruction cache two CUs share — every workgroup of a short launch fetched
// it from L2 again (~70 us per workgroup), and the stepping loop thrashed it (profiles/r06_ab_x2_ground_inline.txt, second part).
struct CtlSinkOpt : CtlSink {
    typedef void dynamic_tag;
    bool on;
};

// ... and with its twelve values in an LDS panel [12][B] instead of twelve doubles that are live — and, at 512 registers, spilled — across every evaluation of
// a launch (the ground-capable Cessna172Xv2 kernels: at 256 lanes their LDS has 27 KB to spare; k_step_air)
template <int B>
struct CtlSinkLds {
    static constexpr bool enabled = true, full = false;
    typedef void dynamic_tag;
    bool on;
    __attribute__((address_space(3))) double* base;   // &panel[lane]
    FBD void put(int k, double v) const {
        int r = -1;
        if (k == FB_Y_KIN + 1) r = 0; else if (k == FB_Y_KIN + 2) r = 1; else if (k == FB_Y_KIN + 15) r = 2; else if (k == FB_Y_KIN + 16) r = 3;
        else if (k == FB_Y_KIN + 25) r = 4; else if (k == FB_Y_KIN + 26) r = 5; else if (k == FB_Y_KIN + 27) r = 6; else if (k == FB_Y_KIN + 36) r = 7;
        else if (k == FB_Y_KIN + 38) r = 8; else if (k == FB_Y_AIR + 20) r = 9; else if (k == FB_Y_AERO) r = 10; else if (k == FB_Y_AERO + 1) r = 11;
        if (r >= 0) base[r * B] = v;
    }
    FBD CtlSink values() const {
        CtlSink t;
        t.theta = base[0 * B]; t.phi = base[1 * B]; t.lat = base[2 * B]; t.lon = base[3 * B]; t.wx = base[4 * B]; t.wy = base[5 * B]; t.wz = base[6 * B];
        t.vd = base[7 * B]; t.chi = base[8 * B]; t.EAS = base[9 * B]; t.alpha = base[10 * B]; t.beta = base[11 * B];
        return t;
    }
};
FBD CtlSink tap_values(const CtlSinkOpt& t) { return t; }
template <int B> FBD CtlSink tap_values(const CtlSinkLds<B>& t) { return t.values(); }

FBD double sgnd(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }
FBD double wrap_to_pi(double x) { return x + 2 * PI * floor((PI - x) / (2 * PI)); }   // FP/attitude.jl:478
constexpr double CTL_INF = __builtin_huge_val();

// ---- compensators; their states are rows of the cs record ------------------------------------------------------
struct PidGains { double k_p, k_i, k_d, tau_f; };
template <class TAB, class P>
FBD PidGains pid_gains(const TAB& T, P lk, double EAS, double h) {
    double g[FB_CTL_PID_REC];
    T.template lookup<FB_CTL_PID_REC>(lk, EAS, h, g);
    return {g[0], g[1], g[2], g[3]};
}
// PID f_periodic! (β_p = β_d = 1); rows s0 .. s0+2 = x_i0, x_d0, sat_out_0
template <class MEM>
FBD double pid_run(const MEM& M, int s0, const PidGains& P, double lo, double hi, double dT, double input, double sat_ext) {
    const double x_i0 = M.S(s0), x_d0 = M.S(s0 + 1), sat0 = M.S(s0 + 2);
    const double a = 1 / (P.tau_f + dT);
    const bool halted = (sgnd(input * sat0) > 0) || (sgnd(input * sat_ext) > 0);
    const double x_i = x_i0 + dT * P.k_i * input * (halted ? 0.0 : 1.0);
    const double x_d = a * P.tau_f * x_d0 + dT * a * P.k_d * input;
    const double out_free = P.k_p * input + x_i + a * (-x_d0 + P.k_d * input);
    M.S(s0) = x_i; M.S(s0 + 1) = x_d;
    M.S(s0 + 2) = (out_free >= hi ? 1.0 : 0.0) - (out_free <= lo ? 1.0 : 0.0);
    return fmin(fmax(out_free, lo), hi);
}
template <class MEM>
FBD void pid_init(const MEM& M, int s0, const PidGains& P, double lo, double hi, double dT) {
    M.S(s0) = 0; M.S(s0 + 1) = 0; M.S(s0 + 2) = 0;
    pid_run(M, s0, P, lo, hi, dT, 0.0, 0.0);
}
// Integrator f_periodic!, unbounded; rows s0, s0+1 = x0, sat_out_0
template <class MEM>
FBD double integ_run(const MEM& M, int s0, double dT, double input, double sat_ext) {
    const bool halted = (sgnd(input * M.S(s0 + 1)) > 0) || (sgnd(input * sat_ext) > 0);
    const double x1 = M.S(s0) + dT * input * (halted ? 0.0 : 1.0);
    M.S(s0) = x1;
    M.S(s0 + 1) = (x1 >= CTL_INF ? 1.0 : 0.0) - (x1 <= -CTL_INF ? 1.0 : 0.0);
    return x1;
}
template <class MEM>
FBD void integ_init(const MEM& M, int s0, double dT) { M.S(s0) = 0; M.S(s0 + 1) = 0; integ_run(M, s0, dT, 0.0, 0.0); }
// LQR{NX,2,2} f_periodic!; g = [K_fbk 2xNX column-major | K_fwd 2x2 | K_int 2x2 | x_trim | u_trim | z_trim];
// rows s0 .. s0+3 = int_out_0[2], out_sat_0[2]; sat_ext is never set by the control laws
template <int NX, class MEM>
FBD void lqr_run(const MEM& M, int s0, const double* g, const double (&lo)[2], const double (&hi)[2], double dT, const double (&x)[NX],
                 const double (&z)[2], const double (&z_ref)[2], double (&out)[2]) {
    const double* K_fbk = g; const double* K_fwd = g + 2 * NX; const double* K_int = K_fwd + 4;
    const double* x_trim = K_int + 4; const double* u_trim = x_trim + NX; const double* z_trim = u_trim + 2;
    const double dz0 = z_ref[0] - z[0], dz1 = z_ref[1] - z[1], dt0 = z_ref[0] - z_trim[0], dt1 = z_ref[1] - z_trim[1];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double int_in = K_int[i] * dz0 + K_int[i + 2] * dz1;
        const bool halted = sgnd(int_in * M.S(s0 + 2 + i)) > 0;
        const double int_out = M.S(s0 + i) + dT * int_in * (halted ? 0.0 : 1.0);
        const double fwd = K_fwd[i] * dt0 + K_fwd[i + 2] * dt1;
        double fbk = K_fbk[i] * (x[0] - x_trim[0]);
#pragma unroll
        for (int k = 1; k < NX; k++) fbk += K_fbk[i + 2 * k] * (x[k] - x_trim[k]);
        const double out_free = u_trim[i] + int_out + fwd - fbk;
        M.S(s0 + i) = int_out;
        M.S(s0 + 2 + i) = (out_free >= hi[i] ? 1.0 : 0.0) - (out_free <= lo[i] ? 1.0 : 0.0);
        out[i] = fmin(fmax(out_free, lo[i]), hi[i]);
    }
}
template <int NX, class MEM>
FBD void lqr_init(const MEM& M, int s0, const double* g, const double (&lo)[2], const double (&hi)[2], double dT) {
    double x[NX], out[2];
#pragma unroll
    for (int k = 0; k < NX; k++) x[k] = 0;
    const double z[2] = {0, 0};
    M.S(s0) = 0; M.S(s0 + 1) = 0; M.S(s0 + 2) = 0; M.S(s0 + 3) = 0;
    lqr_run<NX>(M, s0, g, lo, hi, dT, x, z, z, out);
}

// The same LQR with its gains taken where they are used, through an accessor g(k) (k: the record's index) — for the halves of an update that
// run in 256 registers (x2_periodic_half): interpolating the whole 36-double record first (144 corner loads, 36 results held for the run)
// does not fit there, and what the allocator made of it was a load / wait / spill per gain (profiles/r04_x2_half_stamps.txt). Here every
// gain is consumed as it arrives, in four groups whose corner loads are in flight together (compiler barriers between the groups keep the
// scheduler from hoisting all 144): trims (x_trim, z_trim), then per output i: K_int, K_fwd, K_fbk, u_trim. Same operations, same order of
// summation as lqr_run (x - x_trim is formed once per state instead of once per output: the same value).
template <int REC, class P>
struct LqrGainSrc {
    P a00, a10, a01, a11;
    double wE, wH;
    // the gains idx[0..N): their 4 N corner values are loaded FIRST, all in flight together (the scheduling barrier keeps the interpolation
    // arithmetic from being interleaved load by load, which is what the scheduler otherwise does under register pressure), then interpolated
    // — element by element ctl_interp's expression
    template <int N>
    FBD void fetch(const int (&idx)[N], double (&v)[N]) const {
        double r00[N], r01[N], r10[N], r11[N];
#pragma unroll
        for (int e = 0; e < N; e++) { r00[e] = a00[idx[e]]; r01[e] = a01[idx[e]]; r10[e] = a10[idx[e]]; r11[e] = a11[idx[e]]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < N; e++) v[e] = (1 - wE) * ((1 - wH) * r00[e] + wH * r01[e]) + wE * ((1 - wH) * r10[e] + wH * r11[e]);
    }
};
template <int REC, class TAB, class P>
FBD LqrGainSrc<REC, P> lqr_gain_src(const TAB& T, P lk, double EAS, double h) {
    CtlCell c = T.cell;
    if (!T.o.same_grid) c = ctl_cell(ctl_hdr(lk), EAS, h);
    const P d = lk + FB_CTL_GRID_HDR;
    return {d + c.r00 * REC, d + c.r10 * REC, d + c.r01 * REC, d + c.r11 * REC, c.wE, c.wH};
}
template <int NX, class MEM, class G>
FBD void lqr_run_g(const MEM& M, int s0, const G& g, const double (&lo)[2], const double (&hi)[2], double dT, const double (&x)[NX],
                   const double (&z)[2], const double (&z_ref)[2], double (&out)[2]) {
    constexpr int KF = 0, KW = 2 * NX, KI = KW + 4, XT = KI + 4, UT = XT + NX, ZT = UT + 2;
    const double st[4] = {M.S(s0), M.S(s0 + 1), M.S(s0 + 2), M.S(s0 + 3)};   // int_out_0[2], out_sat_0[2] (read before either output writes them)
    const double dz0 = z_ref[0] - z[0], dz1 = z_ref[1] - z[1];
    int i1[NX + 4];
    double t1[NX + 4];
#pragma unroll
    for (int k = 0; k < NX + 4; k++) i1[k] = XT + k;   // x_trim[NX], u_trim[2], z_trim[2]: contiguous in the record
    static_assert(UT == XT + NX && ZT == UT + 2, "record layout");
    g.template fetch<NX + 4>(i1, t1);
    double dx[NX];
#pragma unroll
    for (int k = 0; k < NX; k++) dx[k] = x[k] - t1[k];
    const double dt0 = z_ref[0] - t1[NX + 2], dt1 = z_ref[1] - t1[NX + 3];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int i2[NX + 4];
        double t2[NX + 4];
        i2[0] = KI + i; i2[1] = KI + i + 2; i2[2] = KW + i; i2[3] = KW + i + 2;
#pragma unroll
        for (int k = 0; k < NX; k++) i2[4 + k] = KF + i + 2 * k;
        g.template fetch<NX + 4>(i2, t2);
        const double int_in = t2[0] * dz0 + t2[1] * dz1;
        const bool halted = sgnd(int_in * st[2 + i]) > 0;
        const double int_out = st[i] + dT * int_in * (halted ? 0.0 : 1.0);
        const double fwd = t2[2] * dt0 + t2[3] * dt1;
        double fbk = t2[4] * dx[0];
#pragma unroll
        for (int k = 1; k < NX; k++) fbk += t2[4 + k] * dx[k];
        const double out_free = t1[NX + i] + int_out + fwd - fbk;
        M.S(s0 + i) = int_out;
        M.S(s0 + 2 + i) = (out_free >= hi[i] ? 1.0 : 0.0) - (out_free <= lo[i] ? 1.0 : 0.0);
        out[i] = fmin(fmax(out_free, lo[i]), hi[i]);
    }
}
// The same run for the longitudinal half of an update whose pitch-axis outer loops run on the PARTNER wave (ctl_lon_pitch_half, below): everything
// that does not depend on z_ref — the trims, x - x_trim, both outputs' feedback sums, the K_int / K_fwd gains — is formed first, `late()` then
// waits for the partner and returns z_ref, and the run ends with lqr_run_g's own expressions for int_in, fwd and the outputs. Same operations on
// the same operands as lqr_run_g; fourteen values ride across the wait.
template <int NX, class MEM, class G, class Late>
FBD void lqr_run_g_late(const MEM& M, int s0, const G& g, const double (&lo)[2], const double (&hi)[2], double dT, const double (&x)[NX],
                        const double (&z)[2], double z_ref0, const Late& late, double (&out)[2], double& z_ref1_out) {
    constexpr int KF = 0, KW = 2 * NX, KI = KW + 4, XT = KI + 4, UT = XT + NX, ZT = UT + 2;
    const double st0 = M.S(s0), st1 = M.S(s0 + 1), st2 = M.S(s0 + 2), st3 = M.S(s0 + 3);
    int i1[NX + 4];
    double t1[NX + 4];
#pragma unroll
    for (int k = 0; k < NX + 4; k++) i1[k] = XT + k;
    static_assert(UT == XT + NX && ZT == UT + 2, "record layout");
    g.template fetch<NX + 4>(i1, t1);
    double dx[NX];
#pragma unroll
    for (int k = 0; k < NX; k++) dx[k] = x[k] - t1[k];
    const double ut0 = t1[NX], ut1 = t1[NX + 1], zt0 = t1[NX + 2], zt1 = t1[NX + 3];
    asm volatile("" : "+v"(dx[0]) : : "memory");
    // (groups sized for the registers of a wave that shares its SIMD: the two feedback rows one after the other, 4 NX corner loads in flight each,
    // then K_int and K_fwd of both outputs in one group — 32 loads — whose eight results are what rides across the wait with the two sums)
    double fbk0, fbk1;
    {
        int i2[NX];
        double t2[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) i2[k] = KF + 2 * k;
        g.template fetch<NX>(i2, t2);
        double f = t2[0] * dx[0];
#pragma unroll
        for (int k = 1; k < NX; k++) f += t2[k] * dx[k];
        fbk0 = f;
    }
    // (the next group's corner loads stay behind this one's arithmetic — in lqr_run_g the record's stores between the outputs see to that; without a
    // fence the load vectoriser merges all 144 corner loads of the record into one burst of dwordx4 and the allocator spills sixty values around it)
    asm volatile("" : "+v"(fbk0) : : "memory");
    {
        int i2[NX];
        double t2[NX];
#pragma unroll
        for (int k = 0; k < NX; k++) i2[k] = KF + 1 + 2 * k;
        g.template fetch<NX>(i2, t2);
        double f = t2[0] * dx[0];
#pragma unroll
        for (int k = 1; k < NX; k++) f += t2[k] * dx[k];
        fbk1 = f;
    }
    asm volatile("" : "+v"(fbk1) : : "memory");
    const int i3[8] = {KI, KI + 2, KW, KW + 2, KI + 1, KI + 3, KW + 1, KW + 3};
    double t3[8];
    g.template fetch<8>(i3, t3);
    asm volatile("" : "+v"(t3[0]), "+v"(t3[1]), "+v"(t3[2]), "+v"(t3[3]), "+v"(t3[4]), "+v"(t3[5]), "+v"(t3[6]), "+v"(t3[7]) : : "memory");   // (interpolated AHEAD of the wait)
    const double z_ref1 = late();   // ----- the partner's outer loops have put their reference -----
    z_ref1_out = z_ref1;
    M.S(s0 + 4) = z_ref0; M.S(s0 + 5) = z_ref1;
    const double dz0 = z_ref0 - z[0], dz1 = z_ref1 - z[1];
    const double dt0 = z_ref0 - zt0, dt1 = z_ref1 - zt1;
    {
        const double int_in = t3[0] * dz0 + t3[1] * dz1;
        const bool halted = sgnd(int_in * st2) > 0;
        const double int_out = st0 + dT * int_in * (halted ? 0.0 : 1.0);
        const double fwd = t3[2] * dt0 + t3[3] * dt1;
        const double out_free = ut0 + int_out + fwd - fbk0;
        M.S(s0) = int_out;
        M.S(s0 + 2) = (out_free >= hi[0] ? 1.0 : 0.0) - (out_free <= lo[0] ? 1.0 : 0.0);
        out[0] = fmin(fmax(out_free, lo[0]), hi[0]);
    }
    {
        const double int_in = t3[4] * dz0 + t3[5] * dz1;
        const bool halted = sgnd(int_in * st3) > 0;
        const double int_out = st1 + dT * int_in * (halted ? 0.0 : 1.0);
        const double fwd = t3[6] * dt0 + t3[7] * dt1;
        const double out_free = ut1 + int_out + fwd - fbk1;
        M.S(s0 + 1) = int_out;
        M.S(s0 + 3) = (out_free >= hi[1] ? 1.0 : 0.0) - (out_free <= lo[1] ? 1.0 : 0.0);
        out[1] = fmin(fmax(out_free, lo[1]), hi[1]);
    }
}
template <int NX, class MEM, class G>
FBD void lqr_init_g(const MEM& M, int s0, const G& g, const double (&lo)[2], const double (&hi)[2], double dT) {
    double x[NX], out[2];
#pragma unroll
    for (int k = 0; k < NX; k++) x[k] = 0;
    const double z[2] = {0, 0};
    M.S(s0) = 0; M.S(s0 + 1) = 0; M.S(s0 + 2) = 0; M.S(s0 + 3) = 0;
    lqr_run_g<NX>(M, s0, g, lo, hi, dT, x, z, z, out);
}

// ---- longitudinal channel ------------------------------------------------------------------------------------------
// which compensators a longitudinal mode runs (c172x_ctl.jl:46-77)
struct LonLoops { bool te2te, q2e, th2q, v2t; };
FBD LonLoops lon_loops(int mode) {
    LonLoops L;
    L.te2te = mode == FB_LON_SAS || mode == FB_LON_THR_Q || mode == FB_LON_THR_THETA || mode == FB_LON_EAS_Q || mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    L.q2e = L.te2te && mode != FB_LON_SAS;
    L.th2q = mode == FB_LON_THR_THETA || mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    L.v2t = mode == FB_LON_EAS_Q || mode == FB_LON_EAS_THETA || mode == FB_LON_EAS_CLM;
    return L;
}
// The pitch-axis outer loops of a mode that runs q2e (c172x_ctl.jl:355-398): climb rate -> θ_ref (c2θ PID), θ_ref -> q_ref, q_ref -> the elevator
// reference of the te2te LQR (q2e integrator + PID). One body, used by ctl_lon (everything on one wave) and by ctl_lon_pitch_half (the partner wave
// of the wave-pair stepper): same operations in the same order either way.
template <class MEM>
FBD double lon_pitch_loops(const MEM& M, double dT, const CtlIn& v, int mode, bool changed, bool th2q, const PidGains& P, const PidGains& Pc,
                           double sat_ele, double clm_ref, double& q_ref, double& theta_ref) {
    constexpr double k_p_theta = 1.0;   // c172x_ctl.jl:235
    const double theta = v.theta, q = v.w_wb_b.y, r = v.w_wb_b.z;
    if (changed) {
        integ_init(M, FB_CS_Q2E_INT, dT);
        pid_init(M, FB_CS_Q2E_PID, P, -CTL_INF, CTL_INF, dT);
        if (P.k_i != 0) M.S(FB_CS_Q2E_PID) = M.S(FB_CS_TE2TE + 5);
    }
    if (th2q) {
        if (mode == FB_LON_EAS_CLM) {
            if (changed) { pid_init(M, FB_CS_C2THETA_PID, Pc, -CTL_INF, CTL_INF, dT); if (Pc.k_i != 0) M.S(FB_CS_C2THETA_PID) = theta; }
            theta_ref = pid_run(M, FB_CS_C2THETA_PID, Pc, -CTL_INF, CTL_INF, dT, clm_ref - v.clm, sat_ele);
        }
        const double theta_dot_ref = k_p_theta * (theta_ref - theta);
        const double phi_bnd = clampd(v.phi, -PI / 3, PI / 3);
        q_ref = 1 / cos(phi_bnd) * theta_dot_ref + r * tan(phi_bnd);
    }
    const double io = integ_run(M, FB_CS_Q2E_INT, dT, q_ref - q, sat_ele);
    return pid_run(M, FB_CS_Q2E_PID, P, -CTL_INF, CTL_INF, dT, io, sat_ele);
}
// The pitch-axis outer loops as the PARTNER wave of the wave-pair stepper runs them (x2_periodic_half<CTL_HALF_LAT>, ahead of the lateral laws):
// the longitudinal half is the long one of an update — te2te LQR 14 k cycles behind 8 k of outer loops, profiles/r05_x2_half_stamps.txt — and
// of its outer loops only v2t (throttle) stays with it. Returns the elevator reference the longitudinal half's LQR takes (lqr_run_g_late);
// writes the loops' rows, Q_REF and THETA_REF (the longitudinal half, ctl_lon<true, true>, leaves those rows alone). The mode is formed as
// ctl_lon forms it as far as it matters here: an EAS_alt request resolves to thr_EAS or EAS_alt by the altitude state, and neither runs these
// loops, so the altitude state (which the longitudinal half rewrites meanwhile) is not read. The previous mode and the rows read on a mode change
// (LON_MODE, TE2TE + 5) are rewritten by the longitudinal half only BEHIND its wait for this function's result.
template <class TAB, class MEM>
FBD double ctl_lon_pitch_half(const TAB& T, const MEM& M, double dT, const CtlIn& v, int mode_req) {
    double q_ref = M.U(FB_CU_Q_REF), theta_ref = M.U(FB_CU_THETA_REF);
    const double clm_ref = M.U(FB_CU_CLM_REF);
    double elevator_ref = clampd(clampd(M.U(FB_CU_ELEVATOR_AXIS), -1, 1) + clampd(M.U(FB_CU_ELEVATOR_OFFSET), -1, 1), -1, 1);
    const int mode = v.on_gnd ? (int)FB_LON_DIRECT : mode_req;
    const LonLoops L = lon_loops(mode);
    if (L.q2e) {
        const bool changed = mode != (int)M.S(FB_CS_LON_MODE);
        const PidGains P_q2e = pid_gains(T, T.lk(3), v.EAS, v.h_e);
        const PidGains P_c2t = pid_gains(T, T.lk(4), v.EAS, v.h_e);
        const double sat_ele = M.S(FB_CS_TE2TE + 3);
        elevator_ref = lon_pitch_loops(M, dT, v, mode, changed, L.th2q, P_q2e, P_c2t, sat_ele, clm_ref, q_ref, theta_ref);
    }
    M.S(FB_CS_Q_REF) = q_ref; M.S(FB_CS_THETA_REF) = theta_ref;
    return elevator_ref;
}
struct NoLate { FBD double operator()() const { return 0.0; } };
// STREAM: the LQR gains through lqr_gain_src / lqr_run_g (the halves of an update), instead of a record interpolated up front
// SPLIT (with STREAM): the pitch-axis outer loops run on the partner wave (ctl_lon_pitch_half); `late()` waits for it and returns its elevator
// reference, taken where the mode runs those loops
template <bool STREAM = false, bool SPLIT = false, class TAB, class MEM, class Late = NoLate>
FBD void ctl_lon(const TAB& T, const MEM& M, double dT, const CtlIn& v, int mode_req, const Late& late = Late()) {
    static_assert(!SPLIT || STREAM, "the split form exists for the halves of an update");
    double q_ref = M.U(FB_CU_Q_REF), theta_ref = M.U(FB_CU_THETA_REF);
    const double EAS_ref = M.U(FB_CU_EAS_REF), clm_ref = M.U(FB_CU_CLM_REF), h_ref = M.U(FB_CU_H_REF);
    const double EAS = v.EAS, h_e = v.h_e, theta = v.theta;
    const double h_err = h_ref - h_e;
    const int h_state = (int)M.S(FB_CS_H_STATE), mode_prev = (int)M.S(FB_CS_LON_MODE);
    double throttle_ref = clampd(clampd(M.U(FB_CU_THROTTLE_AXIS), 0, 1) + clampd(M.U(FB_CU_THROTTLE_OFFSET), 0, 1), 0, 1);
    double elevator_ref = clampd(clampd(M.U(FB_CU_ELEVATOR_AXIS), -1, 1) + clampd(M.U(FB_CU_ELEVATOR_OFFSET), -1, 1), -1, 1);
    double throttle_cmd = throttle_ref, elevator_cmd = elevator_ref;
    constexpr double h_thr = 10.0, h_hys = 1.0;   // c172x_ctl.jl:233-234
    int mode;
    if (v.on_gnd) mode = FB_LON_DIRECT;
    else if (mode_req == FB_LON_EAS_ALT) {
        if (h_state == FB_ALT_ACQUIRE) {
            mode = FB_LON_THR_EAS;
            throttle_ref = h_err > 0 ? 1.0 : 0.0;   // full throttle to climb, idle to descend
            if (fabs(h_err) < h_thr - h_hys) M.S(FB_CS_H_STATE) = FB_ALT_HOLD;
        } else {
            mode = FB_LON_EAS_ALT;
            if (fabs(h_err) > h_thr + h_hys) M.S(FB_CS_H_STATE) = FB_ALT_ACQUIRE;
        }
    } else mode = mode_req;
    const bool changed = mode != mode_prev;
    const LonLoops L = lon_loops(mode);
    const bool te2te = L.te2te, q2e = L.q2e, th2q = L.th2q, v2t = L.v2t;
    const double lo[2] = {0, -1}, hi[2] = {1, 1};
    const double x_red[8] = {v.w_eb_b.y, theta, EAS, v.alpha, v.alpha_filt, v.n_eng, v.pos[0], v.pos[2]};
    double out[2];
    FB_X2_STAMP(27);
    // The gains of the three PID loops are fetched TOGETHER, ahead of the loops: at one wave per SIMD every lookup placed at its
    // point of use was an exposed gather (and, with a header per lookup, a second round trip ahead of it). A lane whose mode runs
    // fewer loops fetches gains it does not use. (The LQR record stays at its point of use: held across the loops, its 36 values
    // push the update's register footprint to where the calling kernel's allocation suffers.)
    PidGains P_v2t = {0, 0, 0, 0}, P_q2e = {0, 0, 0, 0}, P_c2t = {0, 0, 0, 0};
    if (q2e) {
        P_v2t = pid_gains(T, T.lk(5), EAS, h_e);
        if constexpr (!SPLIT) {
            P_q2e = pid_gains(T, T.lk(3), EAS, h_e);
            P_c2t = pid_gains(T, T.lk(4), EAS, h_e);
        }
    }
    if (te2te) {
        const double sat_thr = M.S(FB_CS_TE2TE + 2), sat_ele = M.S(FB_CS_TE2TE + 3);   // te2te_lqr.y.out_sat of the previous update
        if (v2t) {
            const PidGains P = P_v2t;
            if (changed) { pid_init(M, FB_CS_V2T_PID, P, -CTL_INF, CTL_INF, dT); if (P.k_i != 0) M.S(FB_CS_V2T_PID) = M.S(FB_CS_THROTTLE_CMD); }
            throttle_ref = pid_run(M, FB_CS_V2T_PID, P, -CTL_INF, CTL_INF, dT, EAS_ref - EAS, sat_thr);
        }
        if constexpr (!SPLIT) {
            if (q2e) elevator_ref = lon_pitch_loops(M, dT, v, mode, changed, th2q, P_q2e, P_c2t, sat_ele, clm_ref, q_ref, theta_ref);
        }
        FB_X2_STAMP(28);
        const double z[2] = {v.cmd[0], v.cmd[2]};
        double z_ref[2] = {throttle_ref, elevator_ref};
        if constexpr (STREAM && SPLIT) {
            // (the partner's elevator reference arrives inside the run, behind everything that does not need it; rows TE2TE + 4 / + 5 are written there)
            const double er0 = elevator_ref;
            lqr_run_g_late<8>(M, FB_CS_TE2TE, lqr_gain_src<FB_CTL_LQR8_REC>(T, T.lk(0), EAS, h_e), lo, hi, dT, x_red, z, throttle_ref,
                              [&]() { const double h = late(); return q2e ? h : er0; }, out, elevator_ref);
        } else if constexpr (STREAM) {
            M.S(FB_CS_TE2TE + 4) = z_ref[0]; M.S(FB_CS_TE2TE + 5) = z_ref[1];
            lqr_run_g<8>(M, FB_CS_TE2TE, lqr_gain_src<FB_CTL_LQR8_REC>(T, T.lk(0), EAS, h_e), lo, hi, dT, x_red, z, z_ref, out);
        } else {
            double g8[FB_CTL_LQR8_REC];
            T.template lookup<FB_CTL_LQR8_REC>(T.lk(0), EAS, h_e, g8);
            FB_X2_STAMP(29);
            M.S(FB_CS_TE2TE + 4) = z_ref[0]; M.S(FB_CS_TE2TE + 5) = z_ref[1];
            lqr_run<8>(M, FB_CS_TE2TE, g8, lo, hi, dT, x_red, z, z_ref, out);
        }
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    if (mode == FB_LON_THR_EAS) {
        const double z[2] = {v.cmd[0], EAS}, z_ref[2] = {throttle_ref, EAS_ref};
        if constexpr (STREAM) {
            const auto g = lqr_gain_src<FB_CTL_LQR8_REC>(T, T.lk(1), EAS, h_e);
            if (changed) lqr_init_g<8>(M, FB_CS_TV2TE, g, lo, hi, dT);
            lqr_run_g<8>(M, FB_CS_TV2TE, g, lo, hi, dT, x_red, z, z_ref, out);
        } else {
            double g8[FB_CTL_LQR8_REC];
            T.template lookup<FB_CTL_LQR8_REC>(T.lk(1), EAS, h_e, g8);
            if (changed) lqr_init<8>(M, FB_CS_TV2TE, g8, lo, hi, dT);
            lqr_run<8>(M, FB_CS_TV2TE, g8, lo, hi, dT, x_red, z, z_ref, out);
        }
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    if (mode == FB_LON_EAS_ALT) {
        const double x_full[9] = {v.w_eb_b.y, theta, EAS, v.alpha, h_e, v.alpha_filt, v.n_eng, v.pos[0], v.pos[2]};
        const double z[2] = {EAS, h_e}, z_ref[2] = {EAS_ref, h_ref};
        if constexpr (STREAM) {
            const auto g = lqr_gain_src<FB_CTL_LQR9_REC>(T, T.lk(2), EAS, h_e);
            if (changed) lqr_init_g<9>(M, FB_CS_VH2TE, g, lo, hi, dT);
            lqr_run_g<9>(M, FB_CS_VH2TE, g, lo, hi, dT, x_full, z, z_ref, out);
        } else {
            double g[FB_CTL_LQR9_REC];
            T.template lookup<FB_CTL_LQR9_REC>(T.lk(2), EAS, h_e, g);
            if (changed) lqr_init<9>(M, FB_CS_VH2TE, g, lo, hi, dT);
            lqr_run<9>(M, FB_CS_VH2TE, g, lo, hi, dT, x_full, z, z_ref, out);
        }
        throttle_cmd = out[0]; elevator_cmd = out[1];
    }
    M.S(FB_CS_LON_MODE) = mode;
    M.S(FB_CS_THROTTLE_REF) = clampd(throttle_ref, 0, 1); M.S(FB_CS_ELEVATOR_REF) = clampd(elevator_ref, -1, 1);
    if constexpr (!SPLIT) { M.S(FB_CS_Q_REF) = q_ref; M.S(FB_CS_THETA_REF) = theta_ref; }
    M.S(FB_CS_THROTTLE_CMD) = clampd(throttle_cmd, 0, 1); M.S(FB_CS_ELEVATOR_CMD) = clampd(elevator_cmd, -1, 1);
}

// ---- guidance: GuidanceLaws f_periodic! with SegmentGuidance (c172x/guidance/c172x_gdc.jl:113-149, 232-252, 297-329) -----------
struct GeoPt { double lat, lon, h; };
FBD v3 nvec_of(const GeoPt& p) {
    double sla, cla, slo, clo;
    sincos_step(p.lat, sla, cla); sincos_step(p.lon, slo, clo);
    return {cla * clo, cla * slo, sla};
}
FBD v3 ecef_of(const GeoPt& p) {   // Cartesian(Geographic{LatLon, Ellipsoidal}), geodesy.jl:418-428
    const v3 n = nvec_of(p);
    const double R_E = wgs::a / sqrt(1 - wgs::e2 * n.z * n.z);
    return {(R_E + p.h) * n.x, (R_E + p.h) * n.y, (R_E * (1 - wgs::e2) + p.h) * n.z};
}
// Runs before the control laws and may rewrite their inputs (χ_ref + χ_β request, h_ref + EAS_alt request).
template <class MEM>
FBD void gdc_update(const MEM& M, const CtlIn& v) {
    const int mode = v.on_gnd ? (int)FB_GDC_DIRECT : (int)M.U(FB_CU_GDC_MODE_REQ);
    if (mode == FB_GDC_SEGMENT) {
        const GeoPt p1 = {M.U(FB_CU_SEG_P1), M.U(FB_CU_SEG_P1 + 1), M.U(FB_CU_SEG_P1 + 2)};
        const GeoPt p2 = {M.U(FB_CU_SEG_P2), M.U(FB_CU_SEG_P2 + 1), M.U(FB_CU_SEG_P2 + 2)};
        const GeoPt Ob = {v.lat, v.lon, v.h_e};
        const v3 r_e1_e = ecef_of(p1), r_e2_e = ecef_of(p2), r_eb_e = ecef_of(Ob);
        const quat q_en = ltf_quat(nvec_of(Ob));
        const v3 r_1b_n = qrot_inv(q_en, r_eb_e - r_e1_e), r_12_n = qrot_inv(q_en, r_e2_e - r_e1_e);
        const v3 r_1b_h = {r_1b_n.x, r_1b_n.y, 0.0}, r_12_h = {r_12_n.x, r_12_n.y, 0.0};
        const double s_12 = norm(r_12_h);
        const v3 u_12 = {r_12_h.x / s_12, r_12_h.y / s_12, 0.0};
        const double s_1b = dot(u_12, r_1b_h);
        const double e_sb = cross(u_12, r_1b_h).z;                   // cross-track distance, positive right
        const double h_s = p1.h + (p2.h - p1.h) * s_1b / s_12;       // segment altitude abeam the aircraft
        const double chi_12 = atan2(u_12.y, u_12.x);
        constexpr double dchi_inf = PI / 2, e_sf = 250.0, e_thr = 1000.0;   // c172x_gdc.jl:200-204
        const double dchi = -dchi_inf / (PI / 2) * atan(e_sb / e_sf);
        const double chi_ref = wrap_to_pi(chi_12 + dchi);
        const bool hor = M.U(FB_CU_SEG_HOR_REQ) != 0;
        const bool vrt = fabs(e_sb) < e_thr ? (M.U(FB_CU_SEG_VRT_REQ) != 0) : false;
        M.S(FB_CS_SEG_DCHI) = dchi; M.S(FB_CS_SEG_CHI_REF) = chi_ref; M.S(FB_CS_SEG_H_REF) = h_s;
        M.S(FB_CS_SEG_HOR_GDC) = hor ? 1.0 : 0.0; M.S(FB_CS_SEG_VRT_GDC) = vrt ? 1.0 : 0.0;
        M.S(FB_CS_SEG_E_SB) = e_sb; M.S(FB_CS_SEG_S_1B) = s_1b; M.S(FB_CS_SEG_S_2B) = s_1b - s_12;
        if (hor) { M.U(FB_CU_CHI_REF) = chi_ref; M.U(FB_CU_LAT_MODE_REQ) = FB_LAT_CHI_BETA; }
        if (vrt) { M.U(FB_CU_H_REF) = h_s; M.U(FB_CU_LON_MODE_REQ) = FB_LON_EAS_ALT; }
    }
    M.S(FB_CS_GDC_MODE) = mode;
}

// ---- lateral channel -----------------------------------------------------------------------------------------------
// the gains the lateral mode may need — one eight-state LQR record and one PID record, chosen per lane — fetched in one block
struct LatGains { double g8[FB_CTL_LQR8_REC]; PidGains P; };
template <class TAB>
FBD LatGains ctl_lat_gains(const TAB& T, const CtlIn& v, int mode_req) {
    const int mode = v.on_gnd ? (int)FB_LAT_DIRECT : mode_req;
    LatGains G;
    G.P = {0, 0, 0, 0};
    if (mode == FB_LAT_SAS || mode == FB_LAT_P_BETA || mode == FB_LAT_PHI_BETA || mode == FB_LAT_CHI_BETA) {
        T.template lookup<FB_CTL_LQR8_REC>(mode == FB_LAT_SAS ? T.lk(6) : T.lk(7), v.EAS, v.h_e, G.g8);
        G.P = pid_gains(T, mode == FB_LAT_P_BETA ? T.lk(8) : T.lk(9), v.EAS, v.h_e);
    }
    return G;
}
// STREAM (the halves of an update): G is not read — the LQR gains come through lqr_gain_src / lqr_run_g, the PID gains where their loop runs
template <bool STREAM = false, class TAB, class MEM>
FBD void ctl_lat(const TAB& T, const MEM& M, double dT, const CtlIn& v, int mode_req, const LatGains& G) {
    const double p_ref = M.U(FB_CU_P_REF), beta_ref = M.U(FB_CU_BETA_REF), chi_ref = M.U(FB_CU_CHI_REF);
    double phi_ref = M.U(FB_CU_PHI_REF);
    const double EAS = v.EAS, h_e = v.h_e;
    const int mode_prev = (int)M.S(FB_CS_LAT_MODE);
    const int mode = v.on_gnd ? (int)FB_LAT_DIRECT : mode_req;
    const bool changed = mode != mode_prev;
    const double aileron_ref = clampd(clampd(M.U(FB_CU_AILERON_AXIS), -1, 1) + clampd(M.U(FB_CU_AILERON_OFFSET), -1, 1), -1, 1);
    const double rudder_ref = clampd(clampd(M.U(FB_CU_RUDDER_AXIS), -1, 1) + clampd(M.U(FB_CU_RUDDER_OFFSET), -1, 1), -1, 1);
    double aileron_cmd = aileron_ref, rudder_cmd = rudder_ref;
    const double lo[2] = {-1, -1}, hi[2] = {1, 1};
    const double x_lat[8] = {v.w_eb_b.x, v.w_eb_b.z, v.phi, EAS, v.beta, v.beta_filt, v.pos[1], v.pos[3]};
    double out[2];
    if (mode == FB_LAT_SAS) {
        const double z[2] = {v.cmd[1], v.cmd[3]}, z_ref[2] = {aileron_ref, rudder_ref};
        if constexpr (STREAM) lqr_run_g<8>(M, FB_CS_AR2AR, lqr_gain_src<FB_CTL_LQR8_REC>(T, T.lk(6), EAS, h_e), lo, hi, dT, x_lat, z, z_ref, out);
        else lqr_run<8>(M, FB_CS_AR2AR, G.g8, lo, hi, dT, x_lat, z, z_ref, out);
        aileron_cmd = out[0]; rudder_cmd = out[1];
    }
    if (mode == FB_LAT_P_BETA || mode == FB_LAT_PHI_BETA || mode == FB_LAT_CHI_BETA) {
        const double sat_ail = M.S(FB_CS_PHIBETA2AR + 2);
        if (mode == FB_LAT_P_BETA) {
            const PidGains P = STREAM ? pid_gains(T, T.lk(8), EAS, h_e) : G.P;
            if (changed) {
                integ_init(M, FB_CS_P2PHI_INT, dT);
                pid_init(M, FB_CS_P2PHI_PID, P, -CTL_INF, CTL_INF, dT);
                if (P.k_i != 0) M.S(FB_CS_P2PHI_PID) = M.S(FB_CS_PHIBETA2AR + 4);
            }
            const double io = integ_run(M, FB_CS_P2PHI_INT, dT, p_ref - v.w_wb_b.x, sat_ail);
            phi_ref = pid_run(M, FB_CS_P2PHI_PID, P, -CTL_INF, CTL_INF, dT, io, sat_ail);
        } else if (mode == FB_LAT_CHI_BETA) {
            const PidGains P = STREAM ? pid_gains(T, T.lk(9), EAS, h_e) : G.P;
            if (changed) { pid_init(M, FB_CS_CHI2PHI_PID, P, -PI / 4, PI / 4, dT); if (P.k_i != 0) M.S(FB_CS_CHI2PHI_PID) = M.S(FB_CS_PHIBETA2AR + 4); }
            phi_ref = pid_run(M, FB_CS_CHI2PHI_PID, P, -PI / 4, PI / 4, dT, wrap_to_pi(chi_ref - v.chi), sat_ail);
        }
        const double z[2] = {v.phi, v.beta}, z_ref[2] = {phi_ref, beta_ref};
        if constexpr (STREAM) {
            const auto g = lqr_gain_src<FB_CTL_LQR8_REC>(T, T.lk(7), EAS, h_e);
            if (changed) lqr_init_g<8>(M, FB_CS_PHIBETA2AR, g, lo, hi, dT);
            M.S(FB_CS_PHIBETA2AR + 4) = z_ref[0]; M.S(FB_CS_PHIBETA2AR + 5) = z_ref[1];
            lqr_run_g<8>(M, FB_CS_PHIBETA2AR, g, lo, hi, dT, x_lat, z, z_ref, out);
        } else {
            if (changed) lqr_init<8>(M, FB_CS_PHIBETA2AR, G.g8, lo, hi, dT);
            M.S(FB_CS_PHIBETA2AR + 4) = z_ref[0]; M.S(FB_CS_PHIBETA2AR + 5) = z_ref[1];
            lqr_run<8>(M, FB_CS_PHIBETA2AR, G.g8, lo, hi, dT, x_lat, z, z_ref, out);
        }
        aileron_cmd = out[0]; rudder_cmd = out[1];
    }
    M.S(FB_CS_LAT_MODE) = mode;
    M.S(FB_CS_AILERON_REF) = aileron_ref; M.S(FB_CS_RUDDER_REF) = rudder_ref; M.S(FB_CS_PHI_REF) = phi_ref;
    M.S(FB_CS_AILERON_CMD) = clampd(aileron_cmd, -1, 1); M.S(FB_CS_RUDDER_CMD) = clampd(rudder_cmd, -1, 1);
}

}  // namespace fbd
