/*
 * flightbatch.h — C ABI of libflightbatch: MI355X-native batched 6-DOF flight-dynamics integrator.
 *
 * Drop-in boundary for ONE hot path of e271828e/Flight.jl: the per-step ODE evaluation and fixed-step
 * RK4 integration of `Model(SimpleWorld(Cessna172Sv0()))`, executed for N independent aircraft at once.
 * Every entry point cites the reference interface it stands behind (paths relative to the reference
 * repository root; FC = lib/FlightCore/src, FP = lib/FlightPhysics/src, FA = lib/FlightApps/src).
 * The reference has no FFI for this path (it is pure Julia); the binding a maintainer would add is a
 * `ccall` shim — see INTEGRATION.md, following the in-tree precedent FC/joysticks.jl:45-53.
 *
 * Conventions
 *  - every function returns int32 status: 0 = ok, negative = error (message via fb_last_error()).
 *  - handles are opaque, owned by the library; one handle = one HIP device + one stream.
 *  - host arrays are caller-owned, contiguous, column-major [N x Nfield] with the aircraft index
 *    fastest (structure-of-arrays): element (i, k) lives at ptr[k*N + i].
 *  - calls on one handle are not thread-safe (mirrors the reference's io_lock, FC/sim.jl:545);
 *    different handles may be driven from different threads.
 *  - there is NO CPU fallback: fb_create fails when no HIP device is available.
 */
#ifndef FLIGHTBATCH_H
#define FLIGHTBATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fb_handle_s* fb_handle;

/* ---- model / kinematics / dtype ids -------------------------------------------------------- */
enum { FB_MODEL_C172S0 = 0, FB_MODEL_C172X2 = 1, FB_MODEL_ROBOT2D = 2 }; /* FA/c172/c172s/c172s0.jl:14-18 */
enum { FB_KIN_WA = 0, FB_KIN_ECEF = 1, FB_KIN_NED = 2 };                 /* FP/kinematics.jl:148,250,329   */
/* The kinematic block of x follows the mechanisation (Modeling.X of each, kinematics.jl:152-153, 252-253, 331-332):
 * WA q_wb[4] q_ew[4] h_e (Nx = 27), ECEF q_eb[4] n_e[3] h_e (Nx = 26), NED psi theta phi lat lon h_e (Nx = 24); the blocks
 * before (rows 0-11) and after it (w_eb_b, v_eb_b) are the same. fb_dims reports Nx. */
enum { FB_F64 = 0, FB_F32 = 1 };

/* ---- continuous state x[FB_NX] of Cessna172Sv0 + WA kinematics (SURVEY.md §8 state layout) ---- */
enum {
    FB_X_ALPHA_FILT = 0, FB_X_BETA_FILT = 1, /* FA/c172/c172.jl:296 */
    FB_X_LDG_FRC = 2,                        /* left[2], right[2], nose[2]; FP/landinggear.jl:380-382 */
    FB_X_FUEL = 8,                           /* FA/c172/c172.jl:601 */
    FB_X_ENG_OMEGA = 9, FB_X_ENG_IDLE = 10, FB_X_ENG_FRC = 11, /* FP/piston.jl:290-291 */
    FB_X_Q_WB = 12, FB_X_Q_EW = 16, FB_X_H_E = 20,             /* FP/kinematics.jl:152-153 */
    FB_X_OMEGA_EB_B = 21, FB_X_V_EB_B = 24,                    /* FP/dynamics.jl:436 */
    FB_NX = 27
};
/* ---- discrete state s[FB_NS] (int32) ---- */
enum {
    FB_S_STALL = 0,     /* FA/c172/c172.jl:272-274 */
    FB_S_ENG_STATE = 1, /* FP/piston.jl:198-202 : 0 off, 1 starting, 2 running */
    FB_NS = 2
};
/* ---- real inputs u[FB_NU] ---- */
enum {
    FB_U_THROTTLE = 0, FB_U_MIXTURE = 1,                                  /* FP/piston.jl:259-267 */
    FB_U_AILERON = 2, FB_U_ELEVATOR = 3, FB_U_RUDDER = 4,                 /* FA/c172/c172s/c172s.jl:62-72 */
    FB_U_AILERON_OFFSET = 5, FB_U_ELEVATOR_OFFSET = 6, FB_U_RUDDER_OFFSET = 7,
    FB_U_FLAPS = 8, FB_U_BRAKE_LEFT = 9, FB_U_BRAKE_RIGHT = 10,
    FB_U_M_PILOT = 11, FB_U_M_COPILOT = 12, FB_U_M_LPASS = 13, FB_U_M_RPASS = 14, FB_U_M_BAGGAGE = 15, /* c172.jl:521-527 */
    FB_NU = 16
};
/* ---- integer inputs ui[1]: bit field ---- */
enum {
    FB_UI_ENG_START = 1, FB_UI_ENG_STOP = 2, /* FP/piston.jl:260-261 */
    FB_UI_MIXTURE_AUTO = 4,                  /* FP/piston.jl:263 */
    FB_UI_STEERING_ENGAGED = 8,              /* FP/landinggear.jl:52-55 */
    FB_UI_DEFAULT = 4 | 8,
    FB_NUI = 1
};
/* ---- per-aircraft status word (exceptions of the reference become sticky bits) ---- */
enum {
    FB_ST_OK = 0,
    FB_ST_ALT_RANGE = 1,      /* ArgumentError, FP/geodesy.jl:218-221 */
    FB_ST_ISA_RANGE = 2,      /* ArgumentError, FP/atmosphere.jl:133  */
    FB_ST_GROUND_CRASH = 4,   /* GroundCrash,   FP/landinggear.jl:331-347 */
    FB_ST_NAN = 8,
    FB_ST_CONTACT_ASSERT = 16, /* @assert, FP/landinggear.jl:321 */
    FB_ST_LOST_BALANCE = 32    /* LostBalance, FA/robot2d/robot2d.jl:531-561 (Robot2D only) */
};
/* ---- where a terminated aircraft stopped (fb_get_termination) ----
 * The reference stops a simulation at the first exception (FC/sim.jl:561-570); what it leaves behind is mdl.x / mdl.s as they stand
 * at the throw (sim.x forwards to mdl.x, FC/sim.jl:261-275). A terminated aircraft of the batch is frozen in exactly that condition:
 *   - f_ode! threw (ArgumentError of FP/geodesy.jl:218-221 or FP/atmosphere.jl:133, @assert of FP/landinggear.jl:321) while the
 *     integrator evaluated RK stage k2, k3 or k4 of the step from t_step to t_step + dt: x = the ARGUMENT of that evaluation
 *     (x_step + c dt k_{j-1}; f_ode_wrapper! has copied it into mdl.x, FC/sim.jl:306), s unchanged, `step` RK updates completed;
 *   - f_ode! threw at the new state x_step right after the RK update (OrdinaryDiffEq's evaluation of f at the new u, whose outputs
 *     the callbacks read), before any callback of that step ran: x = x_step;
 *   - f_step! threw (GroundCrash, FP/landinggear.jl:331-347; LostBalance, FA/robot2d/robot2d.jl:553-561) at x_step: x and s carry the
 *     part of f_step! that precedes the throw in the reference's order — quaternion renormalisation (FP/aircraftbase.jl:178), stall
 *     flag (FA/c172/c172.jl:720), the contact-regulator resets of the landing-gear units ahead of the one that threw
 *     (FA/c172/c172.jl:485, FP/landinggear.jl:539-548) — and not what follows it (engine state machine, f_periodic!);
 *   - f_ode! threw at x_step after the step's callbacks had run: the re-evaluation of a state f_step! modified, or the first
 *     evaluation after init / after the host changed the state.
 * The status word holds the single bit of that first exception. (FB_ST_NAN is this library's own check at the end of a launch and
 * has no termination record.) */
enum {
    FB_TERM_NONE = 0,
    FB_TERM_OUTSIDE_STEP = 1,   /* the status bit was not raised by fb_step: a single-call verb (fb_f_ode / fb_f_step) raised it, or the host
                                   set it (fb_set_status) without a record of its own; step = RK updates completed at that moment */
    FB_TERM_F_ODE_K2 = 2,
    FB_TERM_F_ODE_K3 = 3,
    FB_TERM_F_ODE_K4 = 4,
    FB_TERM_F_ODE_NEW = 5,
    FB_TERM_F_STEP = 6,
    FB_TERM_F_ODE_REEVAL = 7
};
/* ---- output record y[FB_NY] (what cb_save logs of `mdl.y`, FC/sim.jl:345-347) ---- */
enum {
    FB_Y_KIN = 0,    /* KinData, 40 doubles, FP/kinematics.jl:46-63:
                        e_nb(psi,theta,phi) q_nb[4] q_eb[4] q_en[4] lat lon n_e[3] h_e h_o r_eb_e[3]
                        w_wb_b[3] w_eb_b[3] v_eb_b[3] v_eb_n[3] v_gnd chi_gnd gamma_gnd */
    FB_Y_AIR = 40,   /* AirData, 22 doubles, FP/atmosphere.jl:198-215:
                        v_ew_n[3] v_ew_b[3] v_wb_b[3] T p rho a mu M Tt pt dp q TAS EAS CAS */
    FB_Y_AERO = 62,  /* 16: alpha beta alpha_filt_dot beta_filt_dot C_D C_Y C_L C_l C_m C_n F[3] tau[3]; c172.jl:276-294 */
    FB_Y_LDG = 78,   /* 3 x 11 (left,right,nose): dh wow xi xi_dot F_dmp_zs wr_b.F[3] wr_b.tau[3]; landinggear.jl:210-222,384-395 */
    FB_Y_PWP = 111,  /* 22: engine MAP f mdot omega tau_shaft P_shaft SFC idle_out frc_out (piston.jl:269-287);
                            propeller J Mt wr_b.F[3] wr_b.tau[3] hr_b[3] P eta_p (propellers.jl:378-390) */
    FB_Y_FUEL = 133, /* 1: m_total; c172.jl:594-598 */
    FB_Y_DYN = 134,  /* 40: mp_b.m mp_b.r_OG[3] mp_b.J[9 row-major] wr_b.F[3] wr_b.tau[3] wdot_eb_b[3] vdot_eb_b[3]
                            a_eb_b[3] a_ib_b[3] f_c_c[3] alpha_ib_b[3] g_c_c[3]; dynamics.jl:416-434 */
    FB_NY = 174
};
/* ---- Cessna172Xv2 (FB_MODEL_C172X2): fly-by-wire actuators + gain-scheduled autopilot ----------------------
 * FA/c172/c172x/c172x.jl:19-52,112-143 (Actuator1 x 7, FlyByWireActuation), c172x2.jl:18-60 (Avionics{ctl, gdc}),
 * control/c172x_ctl.jl (ControlLawsLon/Lat), FP/control.jl:123-185 (Integrator), :370-471 (PID), :620-743 (LQR),
 * :950-994 (gain lookups).
 * Continuous state x [N x FB_X2_NX], in the order of the reference's ComponentVector (`act` is the last Systems
 * field, c172.jl:678-686): rows 0-11 as Cessna172Sv0 (aero, ldg, fuel, pwp), rows FB_X2_ACT + {THROTTLE..BRAKE_RIGHT}
 * the seven actuator positions, rows 19-27 kinematics (q_wb, q_ew, h_e), rows 28-33 dynamics (w_eb_b, v_eb_b).
 * Cessna172Xv2(kinematics) (c172x2.jl:57-59) with FB_KIN_ECEF / FB_KIN_NED: the kinematic block is that mechanisation's
 * (8 rows q_eb, n_e, h_e / 6 rows psi, theta, phi, lat, lon, h_e — FP/kinematics.jl:250-425), the dynamics rows follow it
 * directly, and x has 33 / 31 rows (fb_dims reports the number).
 * Discrete state s and inputs u/ui as Cessna172Sv0, except that u's throttle/aileron/elevator/rudder (+offset) rows
 * are ignored: those four actuators are commanded by the control laws; FB_U_FLAPS / FB_U_BRAKE_* are the commands of
 * the remaining three actuators (act.flaps.u etc.). Output record y as Cessna172Sv0. */
enum {
    FB_X2_NX = 34, FB_X2_ACT = 12, FB_X2_KIN = 19, FB_X2_DYN = 28,
    FB_ACT_THROTTLE = 0, FB_ACT_AILERON = 1, FB_ACT_ELEVATOR = 2, FB_ACT_RUDDER = 3, FB_ACT_FLAPS = 4,
    FB_ACT_BRAKE_LEFT = 5, FB_ACT_BRAKE_RIGHT = 6, FB_NACT = 7
};
/* control-law inputs cu [N x FB_NCU] (avionics.ctl.u; ControlLawsLonU c172x_ctl.jl:242-253, ControlLawsLatU :826-836);
 * the two mode requests are stored as doubles holding the enum value */
enum {
    FB_CU_LON_MODE_REQ = 0, FB_CU_THROTTLE_AXIS = 1, FB_CU_THROTTLE_OFFSET = 2, FB_CU_ELEVATOR_AXIS = 3,
    FB_CU_ELEVATOR_OFFSET = 4, FB_CU_Q_REF = 5, FB_CU_THETA_REF = 6, FB_CU_EAS_REF = 7, FB_CU_CLM_REF = 8, FB_CU_H_REF = 9,
    FB_CU_LAT_MODE_REQ = 10, FB_CU_AILERON_AXIS = 11, FB_CU_AILERON_OFFSET = 12, FB_CU_RUDDER_AXIS = 13,
    FB_CU_RUDDER_OFFSET = 14, FB_CU_P_REF = 15, FB_CU_BETA_REF = 16, FB_CU_PHI_REF = 17, FB_CU_CHI_REF = 18,
    /* guidance inputs (avionics.gdc.u.mode_req, gdc.seg.u: c172x/guidance/c172x_gdc.jl:206-210, 281-283). The target segment
     * is given by its end points p1, p2 as (latitude, longitude, ellipsoidal altitude); default Segment(): (0,0,0) -> (1e-3,0,0) */
    FB_CU_GDC_MODE_REQ = 19, FB_CU_SEG_HOR_REQ = 20, FB_CU_SEG_VRT_REQ = 21, FB_CU_SEG_P1 = 22, FB_CU_SEG_P2 = 25,
    FB_NCU = 28
};
enum { FB_GDC_DIRECT = 0, FB_GDC_SEGMENT = 1, FB_GDC_CIRCULAR = 2 }; /* ModeGuidance, c172x_gdc.jl:19-23 */
enum { /* ModeControlLon, c172x_ctl.jl:29-39 */
    FB_LON_DIRECT = 0, FB_LON_SAS = 1, FB_LON_THR_Q = 2, FB_LON_THR_THETA = 3, FB_LON_THR_EAS = 4, FB_LON_EAS_Q = 5,
    FB_LON_EAS_THETA = 6, FB_LON_EAS_CLM = 7, FB_LON_EAS_ALT = 8
};
enum { FB_LAT_DIRECT = 0, FB_LAT_SAS = 1, FB_LAT_P_BETA = 2, FB_LAT_PHI_BETA = 3, FB_LAT_CHI_BETA = 4 }; /* :727-733 */
enum { FB_ALT_ACQUIRE = 0, FB_ALT_HOLD = 1 };                                                           /* :41-44 */
/* control-law record cs [N x FB_NCS]: the discrete state (s) of every compensator plus the outputs (y) other code
 * reads (modes, references, actuator commands). LQR blocks: integrator state[2], output saturation[2] (+ the last
 * z_ref[2] where a later mode change reads it); Integrator: x0, sat_out_0; PID: x_i0, x_d0, sat_out_0. */
enum {
    FB_CS_LON_MODE = 0, FB_CS_H_STATE = 1, FB_CS_THROTTLE_CMD = 2, FB_CS_ELEVATOR_CMD = 3, FB_CS_THROTTLE_REF = 4,
    FB_CS_ELEVATOR_REF = 5, FB_CS_Q_REF = 6, FB_CS_THETA_REF = 7,
    FB_CS_TE2TE = 8,    /* int[2] sat[2] z_ref[2] */
    FB_CS_TV2TE = 14,   /* int[2] sat[2] */
    FB_CS_VH2TE = 18,   /* int[2] sat[2] */
    FB_CS_Q2E_INT = 22, /* x0 sat */
    FB_CS_Q2E_PID = 24, FB_CS_C2THETA_PID = 27, FB_CS_V2T_PID = 30,
    FB_CS_LAT_MODE = 33, FB_CS_AILERON_CMD = 34, FB_CS_RUDDER_CMD = 35, FB_CS_AILERON_REF = 36, FB_CS_RUDDER_REF = 37,
    FB_CS_PHI_REF = 38,
    FB_CS_AR2AR = 39,    /* int[2] sat[2] */
    FB_CS_PHIBETA2AR = 43, /* int[2] sat[2] z_ref[2] */
    FB_CS_P2PHI_INT = 49, FB_CS_P2PHI_PID = 51, FB_CS_CHI2PHI_PID = 54,
    /* guidance outputs (gdc.y.mode, gdc.seg.y: c172x_gdc.jl:212-220, 285-289) */
    FB_CS_GDC_MODE = 57, FB_CS_SEG_DCHI = 58, FB_CS_SEG_CHI_REF = 59, FB_CS_SEG_H_REF = 60, FB_CS_SEG_HOR_GDC = 61,
    FB_CS_SEG_VRT_GDC = 62, FB_CS_SEG_E_SB = 63, FB_CS_SEG_S_1B = 64, FB_CS_SEG_S_2B = 65,
    FB_NCS = 66
};
/* FB_TABLE_CTL_GAINS blob: ten lookups in the order te2te, tv2te, vh2te (LQR, NX = 8, 8, 9), q2e, c2theta, v2t (PID),
 * ar2ar, phibeta2ar (LQR, NX = 8), p2phi, chi2phi (PID)  [files FA/c172/c172x/control/data/{te2te,tv2te,vh2te,q2e,c2θ,v2t,
 * ar2ar,φβ2ar,p2φ,χ2φ}.h5, loaded by c172x_ctl.jl:208-213,816-819]. Each lookup = 6 doubles (n_EAS, n_h, EAS_lo, EAS_hi,
 * h_lo, h_hi: the `bounds` dataset and the grid size) followed by n_EAS*n_h records, EAS index fastest. LQR record:
 * K_fbk [2 x NX] column-major, K_fwd [2 x 2], K_int [2 x 2], x_trim [NX], u_trim [2], z_trim [2]; PID record: k_p k_i k_d tau_f.
 * Interpolation: linear in (EAS, h_e), Flat extrapolation (FP/control.jl:950-966). The lookups may sit on different grids;
 * when all ten headers are equal (the reference's data files) the library locates the grid cell once per control update. */
enum { FB_CTL_GRID_HDR = 6, FB_CTL_LQR8_REC = 36, FB_CTL_LQR9_REC = 39, FB_CTL_PID_REC = 4 };

/* ---- Robot2D (FB_MODEL_ROBOT2D; FA/robot2d/robot2d.jl) ------------------------------------------------
 * state record x[FB_R2_NX]: the 4 continuous states (robot2d.jl:43) followed by the discrete state the reference
 * keeps in vehicle.u / controller.s: [w, v, theta, eta | u_m, lqr_int_out, lqr_out_sat, pid_x_i, pid_x_d, pid_sat_out]
 * inputs u[FB_R2_NU] = ControllerU (robot2d.jl:359-364): [mode (0 motor, 1 velocity, 2 position), m_ref, v_ref, eta_ref]
 * outputs y[FB_R2_NY] = VehicleY (robot2d.jl:32-41): [w, v, theta, eta, u_m, tau_m, w_dot, v_dot]; no int state (ns = 0).
 * FB_TABLE_ROBOT2D blob (23 doubles): vehicle L R m_b m_r J_b J_r k_m b_m J_m (robot2d.jl:20-30), then the
 * LQRDataPoint of robot2d.h5 K_fbk[3] K_fwd K_int x_trim[3] u_trim z_trim, then PID k_p k_i k_d tau_f (robot2d.jl:419-436). */
enum { FB_R2_NXC = 4, FB_R2_NX = 10, FB_R2_NU = 4, FB_R2_NY = 8, FB_R2_NINIT = 3, FB_R2_TABLE_SIZE = 23 };

/* ---- trim (FA/c172/c172.jl:796-818) ---- */
enum {
    FB_TP_N_E = 0, /* n_e[3] */ FB_TP_H_E = 3, FB_TP_PSI_NB = 4, FB_TP_EAS = 5, FB_TP_GAMMA_WB_N = 6,
    FB_TP_PSI_WB_DOT = 7, FB_TP_THETA_WB_DOT = 8, FB_TP_BETA_A = 9, FB_TP_FUEL_LOAD = 10, FB_TP_MIXTURE = 11,
    FB_TP_FLAPS = 12, FB_TP_PAYLOAD = 13, /* 5 masses */
    FB_NTP = 18
};
enum {
    FB_TS_ALPHA_A = 0, FB_TS_PHI_NB = 1, FB_TS_N_ENG = 2, FB_TS_THROTTLE = 3, FB_TS_AILERON = 4,
    FB_TS_ELEVATOR = 5, FB_TS_RUDDER = 6,
    FB_NTS = 7
};
/* ---- lookup tables uploaded by the host. All doubles unless noted. ---- */
enum {
    FB_TABLE_EGM96 = 0,     /* float32 [721 x 1441], column-major [lat, lon]; FP/geodesy.jl:186-198 */
    FB_TABLE_PROPELLER = 1, /* [21 x 21 x 6]: (J, Mt, {C_Fx,C_Mx,C_Fz_a,C_Mz_a,C_P,eta_p}); FP/propellers.jl:235-276 */
    FB_TABLE_PISTON = 2,    /* packed blob, layout in csrc/tables.h; FP/piston.jl:70-195 */
    FB_TABLE_AERO = 3,      /* packed blob, layout in csrc/tables.h; FA/c172/c172.jl:51-199 */
    FB_TABLE_ROBOT2D = 4,   /* FB_R2_TABLE_SIZE doubles, see above; FA/robot2d/robot2d.jl:20-30,419-436 */
    FB_TABLE_CTL_GAINS = 5, /* Cessna172Xv2 autopilot gain lookups, layout above; FP/control.jl:879-994 */
    FB_TABLE_SCENARIO = 6   /* Cessna172Xv2: a scripted scenario as a table (below): the device-side user_callback!; FC/sim.jl:185, 334-336 */
};

/* World-level parameters shared by the whole batch (one SimpleWorld each in the reference, identical here) */
typedef struct fb_params {
    double dt;           /* integration step, FC/sim.jl:188 (default 0.02) */
    int32_t periodic_n;  /* Δt/dt ratio for f_periodic!, FC/sim.jl:189 (unused by C172Sv0: no periodic dynamics) */
    int32_t surface;     /* 0 DryTarmac, 1 WetTarmac, 2 IcyTarmac; FP/terrain.jl:13,38 */
    double T_sl, p_sl;   /* TunableSeaLevel, FP/atmosphere.jl:75-78 */
    double wind_ned[3];  /* TunableWind, FP/atmosphere.jl:165 */
    double h_terrain;    /* HorizontalTerrain elevation (orthometric), FP/terrain.jl:34-36 */
} fb_params;

/* Per-aircraft environment (optional). In the reference every simulation owns its world: TunableSeaLevel's T / p, TunableWind's
 * velocity and HorizontalTerrain's elevation are inputs / parameters of THAT world's atmosphere and terrain models
 * (FP/atmosphere.jl:75-84, 156-165, 269-278; FP/terrain.jl:34-48; FP/world.jl:20-32) — N simulations, N environments. fb_params holds ONE
 * block for the whole batch (the SURVEY's configurations share it); fb_set_env replaces its wind / sea-level / terrain-elevation
 * fields by a row per aircraft: env [N x FB_NENV] (aircraft index fastest, like every array of this ABI). The surface type stays
 * batch-wide. NULL returns to the batch-wide block. fb_trim, fb_f_ode, fb_f_step, fb_f_periodic, fb_f_init and fb_step all read the rows;
 * the stepping kernels are compiled both ways, so a handle without rows runs exactly the code it ran before (measured with rows, one
 * MI355X: Cessna172Sv0 13.92 ms per 50-step launch of 1 048 576 against 13.88, Cessna172Xv2 10.19 against 9.85 — WA mechanisation, the
 * wave-pair kernel; ECEF / NED and FB_F32 handles with rows are stepped by the one-wave fp64 kernel). Not for Robot2D (no environment). */
enum { FB_ENV_WIND_N = 0, FB_ENV_WIND_E = 1, FB_ENV_WIND_D = 2, FB_ENV_T_SL = 3, FB_ENV_P_SL = 4, FB_ENV_H_TERRAIN = 5, FB_NENV = 6 };

/* --------------------------------------------------------------------------------------------- */
/* Model(SimpleWorld(...)) + Simulation(mdl; ...) : FC/modeling.jl:103-153, FC/sim.jl:183-255.
 * device_id >= 0 selects the HIP device; there is no CPU backend (device_id < 0 is an error). */
int32_t fb_create(int32_t model_id, int32_t kin_id, int32_t dtype, int64_t n, int32_t device_id, fb_handle* out);
int32_t fb_destroy(fb_handle h);
int64_t fb_size(fb_handle h);
/* per-model array sizes: continuous+discrete real state, int state, real inputs, output record (0 where absent) */
int32_t fb_dims(fb_handle h, int32_t* nx, int32_t* ns, int32_t* nu, int32_t* ny);

/* Run on an externally created HIP stream (hipStream_t passed as void*); NULL = the handle's own stream. */
int32_t fb_set_stream(fb_handle h, void* hip_stream);
/* Use caller-owned DEVICE memory for x and s [N x FB_NS int32] (e.g. a torch tensor's data_ptr), so collectives can run
 * on the state without a host round trip. NULL restores the handle's own buffers. The buffer holds the DEVICE layout:
 * [N x FB_NX] doubles for every Cessna172Sv0 mechanisation (ECEF / NED: their 8 / 6 kinematic states in rows 12.., the
 * remaining kinematic rows zero, then w_eb_b, v_eb_b in rows 21-26) and [N x FB_X2_NX] for every Cessna172Xv2 mechanisation
 * with the Sv0 rows first (laid out as just described) and the seven actuator positions in rows 27-33; fb_get_state /
 * fb_set_state present the reference's order and row count. */
int32_t fb_attach_state(fb_handle h, void* x_dev, void* s_dev);

/* Lookup tables: what the reference builds at construction time (Appendix B of SURVEY.md). */
int32_t fb_set_table(fb_handle h, int32_t kind, const void* data, const int64_t* dims, int32_t ndims);
int32_t fb_set_params(fb_handle h, const fb_params* p);
int32_t fb_get_params(fb_handle h, fb_params* p);
/* world.atmosphere.{sl, wind}.u / terrain elevation, one row per aircraft (see FB_ENV_* above); env = NULL: back to fb_params.
 * fb_get_env fails when no rows are set. */
int32_t fb_set_env(fb_handle h, const double* env);
int32_t fb_get_env(fb_handle h, double* env);
int32_t fb_has_env(fb_handle h);   /* 1: per-aircraft rows are set, 0: the batch-wide block is in force, < 0: error */

/* mdl.x / mdl.s / mdl.u access : FC/modeling.jl:89-101 ; property forwarding FC/sim.jl:261-275.
 * fb_set_state sets an INITIAL condition: like init! (FC/sim.jl:390-414) it also clears the sticky status words and restarts the
 * clock and the phase of the periodic update. fb_assign_state is the plain `mdl.x .= v` / `mdl.s = v` of a user callback in the
 * middle of a run (FC/sim.jl:331-341): t, the step count and the status words stay as they are. */
int32_t fb_set_state(fb_handle h, const double* x, const int32_t* s);
int32_t fb_assign_state(fb_handle h, const double* x, const int32_t* s);
int32_t fb_get_state(fb_handle h, double* x, int32_t* s);
int32_t fb_set_inputs(fb_handle h, const double* u, const int32_t* ui);
int32_t fb_get_inputs(fb_handle h, double* u, int32_t* ui);

/* f_init!(world, C172.TrimParameters): FP/world.jl:49-57 -> FA/c172/c172.jl:883-942.
 * trim_params [N x FB_NTP]; trim_state [N x FB_NTS] in: initial guess (c172.jl:796-804), out: solution.
 * success[i] = 1 when cost <= 1e-16 (the reference's STOPVAL_REACHED criterion, c172.jl:926,934).
 * On return x, s, u hold the trimmed initial condition (assign!, FA/c172/c172s/c172s.jl:227-263). */
int32_t fb_trim(fb_handle h, const double* trim_params, double* trim_state, int32_t* success, double* cost);

/* f_init!(mdl, init) with a plain per-instance initializer: Robot2D InitParameters [N x 3] = (u_m, w, eta),
 * FA/robot2d/robot2d.jl:208-228,563-570. (C172 initialises through fb_trim or fb_set_state.)
 * Cessna172Xv2 with init = NULL, ninit = 0: the avionics half of f_init!(aircraft, C172.Init(...)) (aircraftbase.jl:255-265) on the
 * state already set with fb_set_state / fb_set_inputs — actuator states = the commands in u (c172x.jl:253-271), brakes released,
 * then f_init!(avionics, vehicle). */
int32_t fb_f_init(fb_handle h, const double* init, int32_t ninit);

/* f_ode!(world) : FP/world.jl:26-32. Uses current x, u, s; writes xdot [N x FB_NX] (may be NULL)
 * and refreshes the output record y. */
int32_t fb_f_ode(fb_handle h, double* xdot);
/* f_step!(world) : FP/world.jl:34-39. Acts on y of the last fb_f_ode, like the reference. */
int32_t fb_f_step(fb_handle h);
/* f_periodic!(Unconditional(), world) : FP/world.jl:41-47 (no-op for C172Sv0: @no_periodic everywhere). */
int32_t fb_f_periodic(fb_handle h);
/* Cessna172Xv2: avionics.ctl.u (inputs cu [N x FB_NCU]) and the control laws' record (cs [N x FB_NCS]: s and the parts of
 * y other code reads). fb_f_periodic runs the control laws once (f_periodic!(Unconditional(), world)); fb_step runs them
 * every periodic_n steps, after the step's f_step!, on the outputs of the step's last f_ode! (FC/sim.jl:204-218). */
int32_t fb_set_ctl_inputs(fb_handle h, const double* cu);
int32_t fb_get_ctl_inputs(fb_handle h, double* cu);
int32_t fb_set_ctl_state(fb_handle h, const double* cs);
int32_t fb_get_ctl_state(fb_handle h, double* cs);
/* mdl.y of the last fb_f_ode / fb_step : y [N x FB_NY] */
int32_t fb_get_outputs(fb_handle h, double* y);
/* The same with SURVEY.md §8(b)'s field mask: only the blocks of the output record named by `field_mask` (FB_YF_* bits) cross
 * PCIe, packed one after the other in FB_Y_* order: y [N x (sum of the selected blocks' widths)] (reading `mdl.y.vehicle.kinematics`
 * alone costs 320 B per aircraft instead of 1392). fb_get_outputs(h, y) == fb_get_output_fields(h, FB_YF_ALL, y). */
enum { FB_YF_KIN = 1, FB_YF_AIR = 2, FB_YF_AERO = 4, FB_YF_LDG = 8, FB_YF_PWP = 16, FB_YF_FUEL = 32, FB_YF_DYN = 64, FB_YF_ALL = 127 };
int32_t fb_get_output_fields(fb_handle h, uint32_t field_mask, double* y);

/* nsteps x step!(sim) : FC/sim.jl:386 — RK4 (OrdinaryDiffEq, fixed dt) + callbacks in the order
 * cb_step, cb_periodic, cb_user(no-op), FC/sim.jl:204-218. Asynchronous on the handle's stream. */
int32_t fb_step(fb_handle h, int64_t nsteps);
/* steps fused per kernel launch (state stays in registers between them). Default 1. */
int32_t fb_set_steps_per_launch(fb_handle h, int32_t k);
int32_t fb_sync(fb_handle h);
/* sim.t : FC/sim.jl:261-275 */
double fb_time(fb_handle h);

/* On-device TimeSeries log: the SavingCallback of Simulation (FC/sim.jl:210-212: log = SavedValues(Float64, Y),
 * cb_save_function = deepcopy(mdl.y), :345-347) and TimeSeries(sim) (:644-704).
 * A sample = `nrows` rows for all N aircraft, taken every `every` steps after the step's callbacks (cb_save is the last
 * callback of the set, :217) into a device buffer of `capacity` samples; rows[j] in [0, Ny) selects output row y[rows[j]]
 * (an f_ode! at the saved instant refreshes y first), FB_LOG_X0 + k selects state row x[k]. every = 0 turns logging off.
 * fb_log_clear drops the samples and restarts the save phase (init!, :395-397); fb_log_record saves the current instant
 * (init! saves y(t0) through reinit!, :405-410). fb_log_read copies samples [first, first+count) to the host:
 * t [count], data [count x nrows x N] (sample slowest, aircraft fastest). fb_step fails when a sample is due and the
 * buffer is full. */
enum { FB_LOG_X0 = 1000 };
int32_t fb_log_configure(fb_handle h, int64_t every, int64_t capacity, const int32_t* rows, int32_t nrows);
int32_t fb_log_clear(fb_handle h);
int32_t fb_log_record(fb_handle h);
int32_t fb_log_count(fb_handle h, int64_t* count);
int32_t fb_log_read(fb_handle h, int64_t first, int64_t count, double* t, double* data);

/* SimulationTermination / ArgumentError mapping: per-aircraft sticky status bits (FB_ST_*). */
int32_t fb_status(fb_handle h, int32_t* status);
/* The termination record of every aircraft: step [N] = number of RK updates completed since the last init when the exception was
 * thrown (-1: not terminated), where [N] = FB_TERM_* (see above). Either pointer may be NULL. ≙ sim.t and the stack frame of the
 * exception the reference reports (FC/sim.jl:561-570). */
int32_t fb_get_termination(fb_handle h, int64_t* step, int32_t* where);
/* Checkpoint restore of that record (beside fb_set_status; there is no reference counterpart: a Julia session keeps its exception).
 * step [N], where [N] as fb_get_termination returned them; entries of aircraft whose status word is zero are ignored. Every call that
 * clears the status words (fb_trim, fb_set_state, Robot2D's fb_f_init) clears the record too; fb_set_status alone gives an
 * aircraft whose word becomes non-zero without a record (FB_TERM_OUTSIDE_STEP, the current step count). */
int32_t fb_set_termination(fb_handle h, const int64_t* step, const int32_t* where);

/* Trajectory collection across GPUs: one RCCL all-gather of the state panels over xGMI; no other
 * communication exists on this path. One process per GPU; rank 0 calls fb_comm_unique_id and the host distributes the 128
 * bytes out of band; every rank calls fb_comm_init (collective: it also exchanges the ranks' shard sizes, readable through
 * fb_comm_shard_sizes), then fb_gather_state enqueues, on the handle's stream, an all-gather of its x (DEVICE layout,
 * [N x FB_NX] or [N x FB_X2_NX] doubles, see fb_attach_state) into recv_dev (device memory, rank-major)
 * [world x Nx x n_max], n_max = the largest shard. Equal shards (n_of[r] == N on every rank): exactly [world x Nx x N], gathered
 * in place from x. Ragged shards: every rank's rows are padded to n_max through a staging copy; rank r's row k holds
 * n_of[r] valid entries at recv_dev[(r * Nx + k) * n_max ...], the rest of the row is unspecified. The handle passed to
 * fb_gather_state must be the one the communicator was initialised with (same N), otherwise the call fails. RCCL is loaded on
 * first use. */
int32_t fb_comm_unique_id(char* id128);
int32_t fb_comm_init(fb_handle h, int32_t world, int32_t rank, const char* id128, void** comm);
int32_t fb_comm_shard_sizes(void* comm, int64_t* n_of /* [world] or NULL */, int64_t* n_max /* or NULL */);
int32_t fb_gather_state(fb_handle h, void* comm, double* recv_dev);
int32_t fb_comm_destroy(void* comm);

/* Checkpoint / restore: together with fb_get/set_state, fb_get/set_inputs and (Xv2) fb_get/set_ctl_inputs|state these
 * capture everything a resumed run needs: the number of steps taken since the last init (the phase of the periodic
 * update, FC/modeling.jl:99 `_n`), sim.t, and the sticky status words (fb_set_state clears them, like init!). */
int32_t fb_get_step_count(fb_handle h, int64_t* count);
int32_t fb_set_step_count(fb_handle h, int64_t count, double t);
int32_t fb_set_status(fb_handle h, const int32_t* status);

/* ---- Scripted scenarios: `user_callback!` on the device (Cessna172Xv2) -------------------------------------------------------------
 * The reference's scenarios are closures that run after every step, behind f_step! / f_periodic! and ahead of the save (FC/sim.jl:185,
 * 204-218, 334-336): a phase symbol and, per phase, "set these inputs; if <condition on the model's outputs> set those and go on to the next
 * phase" (FA demos/c172_demos.jl:423-486 crosswind landing, :525-642 traffic pattern). For a batch the same logic is a TABLE, interpreted per
 * aircraft by a kernel that fb_step launches behind every `every`-th step (fb_step cuts its stepping launches there, as it does at the log's
 * save instants); per aircraft: a phase word (starts at 0), the step count at which the phase was entered, n_par parameter rows (host-set)
 * and n_rec record rows (written by actions, read back by the host). Nothing crosses to the host during a run.
 * One evaluation of one aircraft (status word 0 only): the `always` actions of its phase, in order; then its rules, in order — the FIRST whose
 * condition holds runs its actions and sets the next phase (at most one transition per evaluation, like the demos' if / elseif chains).
 * Blob (fb_set_table(h, FB_TABLE_SCENARIO, blob, &len, 1); doubles; integers as doubles):
 *   header [FB_SCN_HDR]:        magic 5000001, n_phase, n_rule, n_act, n_par, n_rec, 0, 0
 *   phases [n_phase][FB_SCN_PHASE_REC]: first `always` action, their number, first rule, number of rules
 *   rules  [n_rule][FB_SCN_RULE_REC]:   source kind, source row, comparison (FB_SCN_LT ...), constant c, parameter row p or -1, first action,
 *                                       number of actions, next phase.   Holds when  source - (p >= 0 ? par[p] : 0)  <cmp>  c
 *   actions [n_act][FB_SCN_ACT_REC]:    destination kind, destination row (FB_SCN_DST_UI: the bit mask), wrap flag, c0, number of terms (<= 3),
 *                                       3 x (source kind, source row, coefficient).   value = c0 + sum coefficient * source, summed in order,
 *                                       then Attitude.wrap_to_pi (FP/attitude.jl:478) if the flag is set; FB_SCN_DST_UI sets the bits where value != 0
 *                                       and clears them otherwise. A value is read when its action runs (it sees the actions before it).
 * Sources: FB_SCN_SRC_T = sim.t behind the step (steps taken x dt), T_IN_PHASE = (steps taken - step of entry) x dt, X / CS / CU / U / S = rows of
 * the state (device row order) / control-law record / control-law inputs / vehicle inputs / discrete states, PAR / REC = the aircraft's own rows,
 * and of vehicle.y at the current state: ON_GND (is_on_gnd, 0 / 1; c172.jl:998-1001), H_E, PSI / THETA / PHI (e_nb), CHI, EAS, CLM (climb rate). */
enum { FB_SCN_HDR = 8, FB_SCN_PHASE_REC = 4, FB_SCN_RULE_REC = 8, FB_SCN_ACT_REC = 14, FB_SCN_NTERM = 3 };
enum { FB_SCN_SRC_CONST = 0, FB_SCN_SRC_T, FB_SCN_SRC_T_IN_PHASE, FB_SCN_SRC_X, FB_SCN_SRC_CS, FB_SCN_SRC_CU, FB_SCN_SRC_U, FB_SCN_SRC_S,
       FB_SCN_SRC_ON_GND, FB_SCN_SRC_H_E, FB_SCN_SRC_PSI, FB_SCN_SRC_THETA, FB_SCN_SRC_PHI, FB_SCN_SRC_CHI, FB_SCN_SRC_EAS, FB_SCN_SRC_CLM,
       FB_SCN_SRC_PAR, FB_SCN_SRC_REC, FB_SCN_NSRC };
enum { FB_SCN_DST_CU = 0, FB_SCN_DST_U, FB_SCN_DST_UI, FB_SCN_DST_REC, FB_SCN_NDST };
enum { FB_SCN_LT = 0, FB_SCN_GT, FB_SCN_GE, FB_SCN_LE, FB_SCN_EQ, FB_SCN_NE, FB_SCN_ALWAYS };
/* fb_set_table(FB_TABLE_SCENARIO) validates every index of the blob, allocates the per-aircraft rows (phase 0, entry step 0, parameters and
 * records zero) and switches the evaluation on with every = 1. fb_scenario_configure: the period in steps (>= 1), or 0: scenario off, rows freed.
 * The scenario's state is NOT touched by fb_trim / fb_set_state (the reference's closures keep their phase across init!): set it explicitly. */
int32_t fb_scenario_configure(fb_handle h, int32_t every);
int32_t fb_scenario_set_params(fb_handle h, const double* par /* [n_par x N] */);
int32_t fb_scenario_get_params(fb_handle h, double* par /* [n_par x N] */);
int32_t fb_scenario_get_state(fb_handle h, int32_t* phase /* [N] or NULL */, int64_t* since_step /* [N] or NULL */, double* rec /* [n_rec x N] or NULL */);
int32_t fb_scenario_set_state(fb_handle h, const int32_t* phase, const int64_t* since_step, const double* rec /* each [..N] or NULL: left as is */);

/* HIP-event timing on the handle's stream around the fb_step launches issued between begin and end;
 * reports total elapsed ms and the number of stepping-kernel launches. */
int32_t fb_timing_begin(fb_handle h);
int32_t fb_timing_end(fb_handle h, float* ms, int64_t* n_launches);
/* The same window with ONE event pair per stepping launch (both passes of a launch between its pair), up to max_launches of them;
 * after fb_timing_end, fb_timing_launches returns their durations in launch order (ms[0 .. min(*n, cap))), *n = pairs recorded.
 * A measurement aid (bench.py reports median / min / max over launches and warms up until launches agree); no reference counterpart. */
int32_t fb_timing_begin_per_launch(fb_handle h, int64_t max_launches);
int32_t fb_timing_launches(fb_handle h, float* ms, int64_t cap, int64_t* n);

const char* fb_last_error(void);
const char* fb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLIGHTBATCH_H */
