// c172_kernels_f32.hpp — the fp32 airborne stepper of Cessna172Sv0 (FB_F32 handles; BASELINE.json configs[4] asks for an fp32 fleet).
//
// Same algorithm and call order as k_step_air<WA> (c172_kernels.hpp), instantiated from the same source over
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
//     the fp64 ground-capable kernel k_step_air<WA, false, true>.
//   * q_wb is renormalised every step (its fp32 drift per step is of the order of the 1e-8 trigger of the reference).
#pragma once
#include "c172_kernels.hpp"

namespace fbf {

constexpr int XP0 = FB_X_Q_EW, XPN = 5;      // states integrated in fp64: q_ew[4], h_e
// Panels as in k_step_air (c172_kernels.hpp): the six contact-regulator states are identically zero in the air and have no
// rows; the state being evaluated lives in an LDS panel (fp32) that emit() updates in place.
template <int STRIDE>
struct StateLdsF {
    lds_cptr p;   // &panel[lane]
    FBD static constexpr int row(int k) { return k < FB_X_LDG_FRC ? k : k - 6; }
    FBD real operator[](int k) const { return (k >= FB_X_LDG_FRC && k < FB_X_LDG_FRC + 6) ? real(0) : p[row(k) * STRIDE]; }
};
constexpr int NRF = FB_NX - 6;                              // 21 panel rows
constexpr int RP0 = XP0 - 6;                                // panel row of the first fp64-integrated state
FBD constexpr int xsrow(int r) { return r < RP0 ? r : r - XPN; }   // panel row -> row of the fp32 x_n panel (r outside [RP0, RP0+5))

// no v_pk_*_f32 in this kernel: the SLP vectoriser pairs fp32 operations at the price of more register shuffles than it saves
// (2 576 -> 2 411 instructions per RHS, scratch 56 B -> 0)
// emit of the fp32 stepper. The stage enters through wave-uniform scalar branches (not through selects as in the fp64 kernel: with
// two waves per SIMD the SALU work of one wave hides behind the other's VALU, and VALU issue is what bounds this kernel), one
// branch per BATCH of consecutive rows: all panel reads first, then the updates and writes, so the LDS round trips of a batch overlap.
// FB_F32_NAL: how many of the 21 stage sums (k1 + 2 k2 + 2 k3) live in LDS; the rest, from the last panel row down, in registers
#ifndef FB_F32_NAL
#define FB_F32_NAL 9   // (12 in registers: 7.92 -> 7.66 ms per launch, profiles/r04_ab_acc_regs.txt; 249 registers, the LDS panel 37 KB smaller per CU)
#endif
constexpr int F32_NAL = FB_F32_NAL;
template <int B>
struct F32Emit {
    typedef void batched_tag;
    typedef __attribute__((address_space(3))) float* lf_ptr;
    typedef __attribute__((address_space(3))) double* ld_ptr;
    using SV = StateLdsF<B>;
    lf_ptr xs_l; ld_ptr xp_l; lf_ptr acc_l, xc_l;
    float eb, ee; double eed; bool last; int t;
    float* acc_r;   // the stage sums of panel rows >= F32_NAL
    __device__ __forceinline__ float aget(int r) const { return r < F32_NAL ? acc_l[r * B + t] : acc_r[r - F32_NAL]; }
    __device__ __forceinline__ void aset(int r, float v) const { if (r < F32_NAL) acc_l[r * B + t] = v; else acc_r[r - F32_NAL] = v; }
    static __device__ __forceinline__ constexpr bool wide(int j) { return j >= XP0 && j < XP0 + XPN; }   // integrated in fp64
    template <int NE>
    __device__ __forceinline__ void batch(int j0, const float (&k)[NE]) const {   // NE consecutive non-contact rows
        float xsf[NE], ac[NE]; double xsd[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int j = j0 + e, r = SV::row(j);
            ac[e] = aget(r);
            if (wide(j)) { xsd[e] = xp_l[(j - XP0) * B + t]; xsf[e] = 0; } else { xsf[e] = xs_l[xsrow(r) * B + t]; xsd[e] = 0; }
        }
        if (last) {
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int j = j0 + e, r = SV::row(j), idx = r * B + t;
                aset(r, 0.0f);
                if (wide(j)) { const double v = xsd[e] + eed * ((double)ac[e] + (double)eb * (double)k[e]); xc_l[idx] = (float)v; xp_l[(j - XP0) * B + t] = v; }
                else { const float v = __builtin_fmaf(ee, __builtin_fmaf(eb, k[e], ac[e]), xsf[e]); xc_l[idx] = v; xs_l[xsrow(r) * B + t] = v; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int j = j0 + e, r = SV::row(j), idx = r * B + t;
                if (wide(j)) { aset(r, (float)((double)ac[e] + (double)eb * (double)k[e])); xc_l[idx] = (float)(xsd[e] + eed * (double)k[e]); }
                else { aset(r, __builtin_fmaf(eb, k[e], ac[e])); xc_l[idx] = __builtin_fmaf(ee, k[e], xsf[e]); }
            }
        }
    }
    __device__ __forceinline__ void operator()(int j, float kj) const {
        if (j >= FB_X_LDG_FRC && j < FB_X_LDG_FRC + 6) return;   // identically zero in the air
        const float k1[1] = {kj};
        batch<1>(j, k1);
    }
};
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of the same translation unit does not know the gfx950 feature name)
#define FB_NO_PK32 __attribute__((target("no-packed-fp32-ops")))
#else
#define FB_NO_PK32
#endif
__global__ __launch_bounds__(fbd::STEP_BLOCK, 2) FB_NO_PK32 void k_step_f32(fbd::KArgs a, int nsteps) {
    constexpr int B = fbd::STEP_BLOCK;
    using SV = StateLdsF<B>;
    __shared__ float lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ float rk[LDS_RK_DOUBLES];
    __shared__ float xs_l[(NRF - XPN) * B];   // x_n of the fp32-integrated states
    __shared__ double xp_l[XPN * B];          // x_n of the fp64-integrated states
    __shared__ float acc_l[(F32_NAL > 0 ? F32_NAL : 1) * B];   // k1 + 2 k2 + 2 k3 (the rows that do not live in registers)
    float acc_r[NRF > F32_NAL ? NRF - F32_NAL : 1];
#pragma unroll
    for (int k = 0; k < (NRF > F32_NAL ? NRF - F32_NAL : 1); k++) acc_r[k] = 0.0f;
    __shared__ float xc_l[NRF * B];           // the state being evaluated, updated in place by emit()
    // tables: fp64 blob in global memory -> fp32 in LDS (propeller compacted to four coefficients like the fp64 stepper)
    for (int k = threadIdx.x; k < AT_SIZE + PT_SIZE; k += blockDim.x) lds[k] = a.tables_f32[k];
    for (int k = threadIdx.x; k < PR_NJ * PR_NM * PR_NC_STEP; k += blockDim.x)
        lds[LDS_PROP + k] = a.tables_f32[LDS_PROP + (k / PR_NC_STEP) * PR_NC + (k % PR_NC_STEP)];
    for (int k = threadIdx.x; k < LDS_RK_KNOTS; k += blockDim.x) rk[k] = (float)(1.0 / (a.tables[k + 1] - a.tables[k]));
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != 0) return;
    const int t = threadIdx.x;
    bool to_ground = false;
#pragma unroll
    for (int k = 0; k < FB_NX; k++) {
        const double v = a.x[(int64_t)k * a.n + i];
        if (k >= FB_X_LDG_FRC && k < FB_X_LDG_FRC + 6) { to_ground = to_ground || (v != 0.0); continue; }
        const int r = SV::row(k);
        xc_l[r * B + t] = (float)v;
        if (r < F32_NAL) acc_l[r * B + t] = 0.0f;
        if (k >= XP0 && k < XP0 + XPN) xp_l[(k - XP0) * B + t] = v;
        else xs_l[xsrow(r) * B + t] = (float)v;
    }
    if (to_ground) { a.redo[i] = 1; return; }
    InputsAgg in;
    {
        fbd::Inputs in_r;
        fbd::load_inputs(a, i, in_r);
        in.de = (float)in_r.de; in.da = (float)in_r.da; in.dr = (float)in_r.dr; in.df = (float)in_r.df;
        in.throttle = (float)in_r.throttle; in.mixture = (float)in_r.mixture;
#pragma unroll
        for (int k = 0; k < 5; k++) in.m_pld[k] = (float)in_r.m_pld[k];
        in.ui = in_r.ui; in.u_glob = nullptr; in.n = 0;
        in.sum_payload();
        in.sum_aero((lds_cptr)lds + LDS_AERO, (lds_cptr)rk + LDS_AERO);
    }
    int stall = a.s[i], eng = a.s[a.n + i];
    const Env env = {(float)a.env.T_sl, (float)a.env.p_sl, (float)a.env.wind_n, (float)a.env.wind_e, (float)a.env.wind_d, (float)a.env.h_trn, a.env.surface, (float)a.env.ln_p_sl, (float)a.env.k_rt};
    const float dt = (float)a.dt, hdt = (float)(a.dt / 2), dt6 = (float)(a.dt / 6);
    const double dtd = a.dt, hdtd = a.dt / 2, dt6d = a.dt / 6;
    // wave-uniform stage machine and branch-free emit, as in fbd::k_step_air (see there)
    int stage = 0, step = 0;
    bool pending_cb = false, redoing = false;
    bool alive = true, run = true, handoff = false;   // (no lane ends its simulation in this kernel: a status bit is a hand-over)
#pragma unroll 1
    while (true) {
        StepAux aux;
        int lds_off = 0;
        asm volatile("" : "+s"(lds_off));   // keeps the loop-invariant table loads inside the loop (see k_step_air)
        const Tables T = {(lds_cptr)lds + lds_off, a.egm96, (lds_cptr)rk + lds_off, (gk_cptr)a.tables_f32 + lds_off};
        const bool last = stage == 3;
        const float eb = (stage == 1 || stage == 2) ? 2.0f : 1.0f, ee = last ? dt6 : (stage == 2 ? dt : hdt);
        const double eed = last ? dt6d : (stage == 2 ? dtd : hdtd);
        int32_t bits = 0;
        if (run) {
            InputsAgg inl = in;
            asm volatile("" : "+v"(inl.throttle), "+v"(inl.mixture));
            const F32Emit<B> emit = {(typename F32Emit<B>::lf_ptr)xs_l, (typename F32Emit<B>::ld_ptr)xp_l, (typename F32Emit<B>::lf_ptr)acc_l,
                                     (typename F32Emit<B>::lf_ptr)xc_l, eb, ee, eed, last, t, acc_r};
            const SV xv = {(lds_cptr)xc_l + t + lds_off};
            bits = rhs<FB_KIN_WA, false, true>(xv, stall, eng, inl, env, T, emit, aux, NoSink{});
            // within reach of the ground, or an exception (altitude / ISA range): nothing is committed, the fp64 ground-capable kernel takes
            // this lane over and — if the exception is real in fp64 too — ends its simulation where the reference would
            if (bits != 0) { handoff = true; alive = false; run = false; bits = 0; }
        }
        if (redoing) { redoing = false; run = alive; }
        else if (stage == 0 && pending_cb) {   // f_step! on x_{n+1} = the x_n panels (aircraftbase.jl:172-181; kinematics.jl:226-229; c172.jl:375-384,715-724; piston.jl:428-453)
            pending_cb = false;
            step++;
            bool mod = false;
            if (run) {
                {   // q_wb: renormalised every step, silently (fp32 drift); its copy in the evaluation panel has already moved on by
                    // this evaluation's emits (x + dt/2 k1): rescaling it by the same factor keeps the two consistent to O(ulp)
                    float q[4], n2 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) { q[k] = xs_l[xsrow(SV::row(FB_X_Q_WB + k)) * B + t]; n2 += q[k] * q[k]; }
                    const float inr = rsqrtf(n2);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        xs_l[xsrow(SV::row(FB_X_Q_WB + k)) * B + t] = q[k] * inr;
                        xc_l[SV::row(FB_X_Q_WB + k) * B + t] *= inr;
                    }
                    double n2d = 0;   // q_ew: the reference's rule, on the fp64 values
#pragma unroll
                    for (int k = 0; k < 4; k++) n2d += xp_l[k * B + t] * xp_l[k * B + t];
                    const double nr = ::sqrt(n2d);
                    if (fabs(nr - 1.0) > 1e-8) {
#pragma unroll
                        for (int k = 0; k < 4; k++) xp_l[k * B + t] = xp_l[k * B + t] / nr;
                        mod = true;
                    }
                }
                const int stall0 = stall, eng0 = eng;
                if (aux.alpha > c172::alpha_stall_hi) stall = 1;
                else if (aux.alpha < c172::alpha_stall_lo) stall = 0;
                const float w = xs_l[xsrow(SV::row(FB_X_ENG_OMEGA)) * B + t];
                const bool fuel = aux.m_avail > 0;
                const bool start = in.ui & FB_UI_ENG_START, stop = in.ui & FB_UI_ENG_STOP;
                if (eng == 0) { if (start) eng = 1; }
                else if (eng == 1) { if (!start) eng = 0; if (w > c172::w_idle && fuel) eng = 2; }
                else if (stop || w < c172::w_stall || !fuel) eng = 0;
                mod = mod || stall != stall0 || eng != eng0;
            }
            if (step == nsteps || __builtin_amdgcn_ballot_w64(alive) == 0) break;
            if (__builtin_amdgcn_ballot_w64(mod) != 0) {   // k1 must be re-evaluated on the modified x_{n+1}: put it back into the evaluation panel
                if (mod) {
#pragma unroll
                    for (int r = 0; r < NRF; r++) {
                        xc_l[r * B + t] = (r >= RP0 && r < RP0 + XPN) ? (float)xp_l[(r - RP0) * B + t] : xs_l[xsrow(r) * B + t];
                        if (r < F32_NAL) acc_l[r * B + t] = 0.0f; else acc_r[r - F32_NAL] = 0.0f;   // (it held the discarded k1)
                    }
                }
                run = mod; redoing = true;
                continue;
            }
        }
        stage = (stage + 1) & 3;
        pending_cb = (stage == 0);
    }
    if (handoff) { a.redo[i] = 1; return; }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < FB_NX; k++) {
        if (k >= FB_X_LDG_FRC && k < FB_X_LDG_FRC + 6) continue;
        const int r = SV::row(k);
        const double v = (k >= XP0 && k < XP0 + XPN) ? xp_l[(k - XP0) * B + t] : (double)xs_l[xsrow(r) * B + t];
        bad = bad || !isfinite(v);
        a.x[(int64_t)k * a.n + i] = v;
    }
    if (bad) a.status[i] |= FB_ST_NAN;
    a.s[i] = stall;
    a.s[a.n + i] = eng;
}

}  // namespace fbf
