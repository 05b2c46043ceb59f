// c172x_kernels.hpp — Cessna172Xv2 kernels besides the stepping kernels (k_step_air<KIN, true, GROUND> in c172_kernels.hpp, which also run the control laws every Δt):
//   k_x2_ctl   f_periodic!(avionics, vehicle): the control laws on the outputs of the last f_ode! (aircraftbase.jl:232-242)
//   k_x2_init  f_init!(aircraft, trim) after the trim solve: actuator states and the control-law initialisation
//              (c172x.jl:285-326; aircraftbase.jl:255-265; c172x_ctl.jl:463-519, 1000-1032)
#pragma once
#include "c172_kernels.hpp"
#include "c172x_ctl_device.hpp"

namespace fbd {

struct CtlArgs {
    const double* gains;   // FB_TABLE_CTL_GAINS blob, global memory
    CtlOffsets off;
    double dT;        // the control laws' sample period Δt
    int use_q_pre;    // 1: take q_wb, q_ew from KArgs::q_pre (the state the last f_ode! of the step saw, before f_step!)
};

// vehicle.y as the control laws see it (the ground-capable form of c172_kernels.hpp's x2_ctl_inputs, discrete states from memory)
template <int KIN, class CmdFn>
FBD CtlIn x2_ctl_inputs(const KArgs& a, int64_t i, const Tables& T, const double (&x)[FB_X2_NX], CmdFn&& cmd_of) {
    return x2_ctl_inputs<true, KIN>(a, i, T, x, a.s[i], a.s[a.n + i], a.ui[i], cmd_of);
}

constexpr int CTL_GAINS_MAX = 6144;   // doubles of LDS reserved for the gains blob (the shipped lookups need 5744)
template <int KIN>
__global__ __launch_bounds__(256) void k_x2_ctl(KArgs a, CtlArgs c) {
    // The partial sink needs none of the aero / engine / propeller tables (everything that reads them is dead code here), so
    // they are not staged; the pointers below are never dereferenced.
    __shared__ double gains_l[CTL_GAINS_MAX];
    for (int k = threadIdx.x; k < c.off.total; k += blockDim.x) gains_l[k] = c.gains[k];
    __syncthreads();
    double* lds = gains_l; double* rk = gains_l;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != 0) return;
    const Tables T = {(lds_cptr)lds, a.egm96, (lds_cptr)rk};
    double x[FB_X2_NX];
#pragma unroll
    for (int k = 0; k < FB_X2_NX; k++) x[k] = a.x[(int64_t)k * a.n + i];
    if (c.use_q_pre) {
#pragma unroll
        for (int k = 0; k < 8; k++) x[FB_X_Q_WB + k] = a.q_pre[(int64_t)k * a.n + i];
    }
    const CtlIn v = x2_ctl_inputs<KIN>(a, i, T, x, [&](int k) { return x2_command(a, i, k); });
    const CtlMem M = {const_cast<double*>(a.cu) + i, a.cs + i, a.n};
    gdc_update(M, v);   // Avionics f_periodic!: guidance first, then the control laws (c172x2.jl:27-37)
    const CtlTabT<ldsd_cptr> tab = ctl_tab((ldsd_cptr)gains_l, c.off, v.EAS, v.h_e);
    ctl_lon(tab, M, c.dT, v, (int)M.U(FB_CU_LON_MODE_REQ));
    const int lat_req = (int)M.U(FB_CU_LAT_MODE_REQ);
    ctl_lat(tab, M, c.dT, v, lat_req, ctl_lat_gains(tab, v, lat_req));
}

// After k_trim has left the trimmed Sv0 state, u and s: actuator states = actuator commands = trim values (c172x.jl:253-271),
// brakes released, then f_init!(avionics, vehicle) for a freshly built model.
template <int KIN>
__global__ __launch_bounds__(256) void k_x2_init(KArgs a, CtlArgs c) {
    __shared__ double lds[LDS_TABLE_DOUBLES_STEP];
    __shared__ double rk[LDS_RK_DOUBLES];
    stage_tables<PR_NC_STEP>(lds, rk, a.tables);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int64_t n = a.n;
    const Tables T = {(lds_cptr)lds, a.egm96, (lds_cptr)rk};
    double* u = const_cast<double*>(a.u);
    double* cu = const_cast<double*>(a.cu);
    double x[FB_X2_NX];
#pragma unroll
    for (int k = 0; k < FB_NX; k++) x[k] = a.x[(int64_t)k * n + i];
    u[(int64_t)FB_U_BRAKE_LEFT * n + i] = 0; u[(int64_t)FB_U_BRAKE_RIGHT * n + i] = 0;
    const double cmd7[FB_NACT] = {u[(int64_t)FB_U_THROTTLE * n + i], u[(int64_t)FB_U_AILERON * n + i], u[(int64_t)FB_U_ELEVATOR * n + i],
                                  u[(int64_t)FB_U_RUDDER * n + i], u[(int64_t)FB_U_FLAPS * n + i], 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < FB_NACT; k++) { x[X2_ACT + k] = cmd7[k]; a.x[(int64_t)(X2_ACT + k) * n + i] = cmd7[k]; }
    const CtlIn v = x2_ctl_inputs<KIN>(a, i, T, x, [&](int k) { return clampd(cmd7[k], k == 0 ? 0.0 : -1.0, 1.0); });
    const CtlMem M = {cu + i, a.cs + i, n};
    for (int k = 0; k < FB_NCS; k++) M.S(k) = 0;
    M.S(FB_CS_H_STATE) = FB_ALT_HOLD;
    auto U = [&](int k) -> double& { return cu[(int64_t)k * n + i]; };
    U(FB_CU_THROTTLE_AXIS) = v.pos[0]; U(FB_CU_ELEVATOR_AXIS) = v.pos[2]; U(FB_CU_THROTTLE_OFFSET) = 0; U(FB_CU_ELEVATOR_OFFSET) = 0;
    U(FB_CU_Q_REF) = v.w_wb_b.y; U(FB_CU_THETA_REF) = v.theta; U(FB_CU_EAS_REF) = v.EAS; U(FB_CU_CLM_REF) = v.clm; U(FB_CU_H_REF) = v.h_e;
    U(FB_CU_AILERON_AXIS) = v.pos[1]; U(FB_CU_RUDDER_AXIS) = v.pos[3]; U(FB_CU_AILERON_OFFSET) = 0; U(FB_CU_RUDDER_OFFSET) = 0;
    U(FB_CU_P_REF) = v.w_wb_b.x; U(FB_CU_PHI_REF) = v.phi; U(FB_CU_BETA_REF) = v.beta; U(FB_CU_CHI_REF) = v.chi;
    // one pass in every SAS-based mode loads the trim point into the LQR trackers; both channels end in `direct`
    const int lon_seq[4] = {FB_LON_SAS, FB_LON_THR_EAS, FB_LON_EAS_ALT, FB_LON_DIRECT};
    const CtlTabT<const double*> tab = ctl_tab(c.gains, c.off, v.EAS, v.h_e);
#pragma unroll 1
    for (int m = 0; m < 4; m++) ctl_lon(tab, M, c.dT, v, lon_seq[m]);
    const int lat_seq[3] = {FB_LAT_SAS, FB_LAT_PHI_BETA, FB_LAT_DIRECT};
#pragma unroll 1
    for (int m = 0; m < 3; m++) ctl_lat(tab, M, c.dT, v, lat_seq[m], ctl_lat_gains(tab, v, lat_seq[m]));
    U(FB_CU_LON_MODE_REQ) = FB_LON_DIRECT; U(FB_CU_LAT_MODE_REQ) = FB_LAT_DIRECT;
    // guidance defaults: mode direct, no requests, target = Segment() (c172x_gdc.jl:85, 206-210, 281-283)
    U(FB_CU_GDC_MODE_REQ) = FB_GDC_DIRECT; U(FB_CU_SEG_HOR_REQ) = 0; U(FB_CU_SEG_VRT_REQ) = 0;
    U(FB_CU_SEG_P1) = 0; U(FB_CU_SEG_P1 + 1) = 0; U(FB_CU_SEG_P1 + 2) = 0;
    U(FB_CU_SEG_P2) = 1e-3; U(FB_CU_SEG_P2 + 1) = 0; U(FB_CU_SEG_P2 + 2) = 0;
}

}  // namespace fbd
