// scenario_kernels.hpp — k_scenario: the device-side `user_callback!` of a batch of Cessna172Xv2.
//
// Reference: the closures handed to Simulation(...; user_callback!) run after every step, behind f_step! and f_periodic! and ahead of the save
// (lib/FlightCore/src/sim.jl:185, 204-218, 334-336); the scripted scenarios are such closures — a `phase` symbol, per phase "set these inputs; if
// <condition on the model's outputs> set those and move on" (lib/FlightApps/demos/c172_demos.jl:423-486 crosswind landing, :525-642 traffic
// pattern). Here the same logic is DATA (FB_TABLE_SCENARIO, include/flightbatch.h; built by flightbatch/scenario.py): phases, rules, actions,
// interpreted per aircraft by this kernel, which fb_step launches behind every `every` steps. One phase word, one entry step, n_par parameters
// and n_rec record slots per aircraft live in device memory; nothing crosses to the host during a run.
//
// What the closures read of the model — vehicle.y.kinematics.h_e / e_nb.ψ, is_on_gnd(vehicle), avionics.y.gdc.seg.data, engine.y.state — comes from
// ONE evaluation of f_ode! at the current state with a partial sink (like k_x2_ctl: everything that does not feed the sink is dead code), the
// control-law record and the discrete states in memory. One lane = one aircraft; lanes of a wave may sit in different phases (the branches are
// paid once per evaluation, not per stage of a step).
#pragma once
#include "c172x_kernels.hpp"

namespace fbd {

struct ScnArgs {
    const double* prog;      // the blob (global memory)
    int n_ph, n_rule, n_act, n_par, n_rec;
    int32_t* phase;          // [n]
    long long* since;        // [n] steps taken when the aircraft entered its phase
    double* par;             // [n_par x n]
    double* rec;             // [n_rec x n]
    long long step;          // steps taken since init (this evaluation stands behind step number `step`)
    double t, dt;
};

// the outputs a scenario may read, tapped from rhs() (rows of the output record, include/flightbatch.h FB_Y_*)
struct ScnSink {
    static constexpr bool enabled = true, full = false;
    double psi, theta, phi, vd, chi, EAS;
    FBD void put(int k, double v) {
        if (k == FB_Y_KIN) psi = v;
        else if (k == FB_Y_KIN + 1) theta = v;
        else if (k == FB_Y_KIN + 2) phi = v;
        else if (k == FB_Y_KIN + 36) vd = v;
        else if (k == FB_Y_KIN + 38) chi = v;
        else if (k == FB_Y_AIR + 20) EAS = v;
    }
};

// which sources come from an evaluation of f_ode! (everything else is a row in memory, the step count, or a constant)
FBD bool scn_needs_y(int kind) { return kind == FB_SCN_SRC_ON_GND || (kind >= FB_SCN_SRC_PSI && kind <= FB_SCN_SRC_CLM); }
static_assert(FB_SCN_SRC_PSI < FB_SCN_SRC_THETA && FB_SCN_SRC_THETA < FB_SCN_SRC_PHI && FB_SCN_SRC_PHI < FB_SCN_SRC_CHI && FB_SCN_SRC_CHI < FB_SCN_SRC_EAS &&
              FB_SCN_SRC_EAS < FB_SCN_SRC_CLM && FB_SCN_SRC_H_E < FB_SCN_SRC_PSI && FB_SCN_SRC_CLM < FB_SCN_SRC_PAR, "the tapped outputs are one range of source kinds");

// The evaluation of f_ode! is what an evaluation of the table costs (the ground-capable instance: weight on wheels), and most evaluations do not
// need it: a phase that waits for `h_e - h_runway < 6` or `s_2b > -200` reads rows in memory. So the table is walked in three stages, in the
// order its semantics prescribe: (A) the `always` actions and the rules, as far as they read nothing of vehicle.y — up to the first rule whose
// condition does, or the first rule that holds; (B) ONE evaluation for the wave if any of its lanes has stopped at something that reads vehicle.y
// (a condition, the actions of the rule that fired, `always` actions); (C) the rest of the walk from where (A) stopped.
template <int KIN>
__global__ __launch_bounds__(256) void k_scenario(KArgs a, ScnArgs sc) {
    __shared__ double dummy_l[8];   // (the partial sink needs none of the staged tables: see k_x2_ctl)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.status[i] != 0) return;   // an aircraft whose simulation has ended gets no callback (FC/sim.jl:561-570)
    const int p = sc.phase[i];
    if (p < 0 || p >= sc.n_ph) return;
    const int64_t n = a.n;
    double* const cu = const_cast<double*>(a.cu);
    double* const uu = const_cast<double*>(a.u);
    int32_t* const uiw = const_cast<int32_t*>(a.ui);
    const double* PH = sc.prog + FB_SCN_HDR;
    const double* RU = PH + FB_SCN_PHASE_REC * sc.n_ph;
    const double* AC = RU + FB_SCN_RULE_REC * sc.n_rule;
    const long long since = sc.since[i];
    bool inputs_changed = false;
    ScnSink tap;
    tap.psi = tap.theta = tap.phi = tap.vd = tap.chi = tap.EAS = 0;
    double on_gnd = 0;
    auto source = [&](int kind, int row) -> double {
        switch (kind) {
            case FB_SCN_SRC_CONST: return 1.0;
            case FB_SCN_SRC_T: return sc.t;
            case FB_SCN_SRC_T_IN_PHASE: return (double)(sc.step - since) * sc.dt;
            case FB_SCN_SRC_X: return a.x[(int64_t)row * n + i];
            case FB_SCN_SRC_CS: return a.cs[(int64_t)row * n + i];
            case FB_SCN_SRC_CU: return cu[(int64_t)row * n + i];
            case FB_SCN_SRC_U: return uu[(int64_t)row * n + i];
            case FB_SCN_SRC_S: return (double)a.s[(int64_t)row * n + i];
            case FB_SCN_SRC_ON_GND: return on_gnd;
            case FB_SCN_SRC_H_E: return a.x[(int64_t)h_e_row<KIN>() * n + i];
            case FB_SCN_SRC_PSI: return tap.psi;
            case FB_SCN_SRC_THETA: return tap.theta;
            case FB_SCN_SRC_PHI: return tap.phi;
            case FB_SCN_SRC_CHI: return tap.chi;
            case FB_SCN_SRC_EAS: return tap.EAS;
            case FB_SCN_SRC_CLM: return -tap.vd;
            case FB_SCN_SRC_PAR: return sc.par[(int64_t)row * n + i];
            case FB_SCN_SRC_REC: return sc.rec[(int64_t)row * n + i];
        }
        return 0.0;
    };
    auto run = [&](const double* ac) {
        double v = ac[3];
        const int nt = (int)ac[4];
        for (int k = 0; k < nt; k++) v = v + ac[7 + 3 * k] * source((int)ac[5 + 3 * k], (int)ac[6 + 3 * k]);
        if (ac[2] != 0) v = wrap_to_pi(v);
        const int dst = (int)ac[0], row = (int)ac[1];
        if (dst == FB_SCN_DST_CU) cu[(int64_t)row * n + i] = v;
        // (an `always` action that assigns what is there already — brakes held, throttle closed, every evaluation of the ground phase — changes nothing:
        // the derivative the steppers carry from launch to launch stays valid)
        else if (dst == FB_SCN_DST_U) { double& r = uu[(int64_t)row * n + i]; if (!(r == v)) { r = v; inputs_changed = true; } }
        else if (dst == FB_SCN_DST_REC) sc.rec[(int64_t)row * n + i] = v;
        else if (dst == FB_SCN_DST_UI) { const int w = uiw[i], w1 = v != 0 ? (w | row) : (w & ~row); if (w1 != w) { uiw[i] = w1; inputs_changed = true; } }
    };
    auto acts_need_y = [&](int first, int count) {
        bool need = false;
        for (int k = 0; k < count; k++) {
            const double* ac = AC + FB_SCN_ACT_REC * (first + k);
            const int nt = (int)ac[4];
            for (int t = 0; t < nt; t++) need = need || scn_needs_y((int)ac[5 + 3 * t]);
        }
        return need;
    };
    auto holds = [&](const double* ru) {
        const int tp = (int)ru[4];
        double lhs = source((int)ru[0], (int)ru[1]);
        if (tp >= 0) lhs = lhs - sc.par[(int64_t)tp * n + i];   // (as the demos write it: h_e - final_leg.p2.h < 6)
        const double thr = ru[3];
        switch ((int)ru[2]) {
            case FB_SCN_LT: return lhs < thr;
            case FB_SCN_GT: return lhs > thr;
            case FB_SCN_GE: return lhs >= thr;
            case FB_SCN_LE: return lhs <= thr;
            case FB_SCN_EQ: return lhs == thr;
            case FB_SCN_NE: return lhs != thr;
        }
        return true;
    };
    const double* ph = PH + FB_SCN_PHASE_REC * p;
    const int a0 = (int)ph[0], na = (int)ph[1], r0 = (int)ph[2], nr = (int)ph[3];
    // ---- (A) as far as the walk reads nothing of vehicle.y ----
    bool need = acts_need_y(a0, na), always_done = false;
    int r = 0, fire = -1;
    if (!need) {
        for (int k = 0; k < na; k++) run(AC + FB_SCN_ACT_REC * (a0 + k));
        always_done = true;
        for (; r < nr; r++) {
            const double* ru = RU + FB_SCN_RULE_REC * (r0 + r);
            if (scn_needs_y((int)ru[0])) { need = true; break; }
            if (holds(ru)) { fire = r; need = acts_need_y((int)ru[5], (int)ru[6]); break; }
        }
    }
    // ---- (B) vehicle.y at the current state, if any lane of the wave has stopped at something that reads it ----
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const Tables T = {(lds_cptr)dummy_l, a.egm96, (lds_cptr)dummy_l};
        double x[FB_X2_NX];
#pragma unroll
        for (int k = 0; k < FB_X2_NX; k++) x[k] = a.x[(int64_t)k * n + i];
        const InputsX in = {&x[X2_ACT], a.u + i, n, a.ui[i]};
        StepAux aux;
        rhs<KIN, true>(x, a.s[i], a.s[n + i], in, env_any(a, i), T, [](int, double) {}, aux, tap);
        on_gnd = aux.wow != 0 ? 1.0 : 0.0;   // is_on_gnd: any strut with weight on wheels (c172.jl:998-1001)
    }
    // ---- (C) the rest of the walk ----
    if (!always_done)
        for (int k = 0; k < na; k++) run(AC + FB_SCN_ACT_REC * (a0 + k));
    if (fire < 0)
        for (; r < nr; r++)
            if (holds(RU + FB_SCN_RULE_REC * (r0 + r))) { fire = r; break; }
    if (fire >= 0) {   // at most one transition per evaluation (the demos' if / elseif chains)
        const double* ru = RU + FB_SCN_RULE_REC * (r0 + fire);
        const int f = (int)ru[5], m = (int)ru[6];
        for (int k = 0; k < m; k++) run(AC + FB_SCN_ACT_REC * (f + k));
        sc.phase[i] = (int)ru[7];
        sc.since[i] = sc.step;
    }
    // the vehicle's inputs have changed under the derivative the stepping kernels carry from launch to launch
    if (inputs_changed && a.k1_valid) a.k1_valid[i] = 0;
}

}  // namespace fbd
