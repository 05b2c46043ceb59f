/*
 * tables.h — layout of the lookup-table blobs handed to fb_set_table().
 *
 * The host (Julia through ccall, or the Python mirror in flightbatch/tables.py) builds the tables the
 * way the reference does at construction time and ships them as flat arrays of doubles; the kernels
 * stage them into LDS once per workgroup. Offsets are in doubles. 2-D tables are column-major
 * [n1 x n2] like the Julia arrays they come from, except the propeller table which is interleaved
 * (see below) so that the six coefficients of one grid corner are contiguous in LDS.
 *
 * Reference: lib/FlightApps/src/c172/c172.jl:51-199 (aero), lib/FlightPhysics/src/piston.jl:70-195,
 * lib/FlightPhysics/src/propellers.jl:235-276.
 */
#ifndef FLIGHTBATCH_TABLES_H
#define FLIGHTBATCH_TABLES_H

/* ---------------- aerodynamic coefficient tables (c172.jl:51-199) ---------------- */
enum {
    AT_GE_K = 0,           /* 13 knots: non-dimensional height (C_D.ge, C_L.ge share them) */
    AT_CD_GE_V = 13,       /* 13 */
    AT_CL_GE_V = 26,       /* 13 */
    AT_DF4_K = 39,         /* 4 knots: flap deflection 0,10,20,30 deg in rad */
    AT_CD_DF_V = 43,       /* 4 */
    AT_CL_DF_V = 47,       /* 4 */
    AT_CM_DF_V = 51,       /* 4 */
    AT_UNIT3_K = 55,       /* 3 knots: -1, 0, 1 (C_D.δe, C_D.β) */
    AT_CD_DE_V = 58,       /* 3 */
    AT_CD_BETA_V = 61,     /* 3 */
    AT_CD_ALPHA_K = 64,    /* 26 knots */
    AT_CD_ALPHA_DF_V = 90, /* 26 x 4 */
    AT_CY_BETA_K = 194,    /* 3 knots */
    AT_DF2_K = 197,        /* 2 knots: 0, 30 deg in rad */
    AT_CY_BETA_DF_V = 199, /* 3 x 2 */
    AT_ALPHA2_K = 205,     /* 2 knots: 0, 0.094 */
    AT_CY_P_V = 207,       /* 2 x 2 */
    AT_CY_R_V = 211,       /* 2 x 2 */
    AT_CL_R_V = 215,       /* 2 x 2  (C_l.r) */
    AT_CL_ALPHA_K = 219,   /* 17 knots */
    AT_CL_ALPHA_V = 236,   /* 17 x 2 (second axis: stall 0/1) */
    AT_SCALARS = 270,      /* 21 scalar derivatives, order below */
    AT_SIZE = 291
};
enum { /* offsets inside AT_SCALARS */
    AS_CD_ZERO = 0, AS_CY_DR, AS_CY_DA, AS_CL_DE, AS_CL_Q, AS_CL_ALPHA_DOT, AS_Cl_DA, AS_Cl_DR, AS_Cl_BETA, AS_Cl_P,
    AS_CM_ZERO, AS_CM_DE, AS_CM_ALPHA, AS_CM_Q, AS_CM_ALPHA_DOT, AS_CN_DR, AS_CN_DA, AS_CN_BETA, AS_CN_P, AS_CN_R,
    AS_COUNT
};

/* ---------------- piston engine maps (piston.jl:70-195) ---------------- */
enum {
    PT_DELTA_WOT_V = 0, /* 2 x 9 ; n in range(0.667,1,2), mu in range(0.401,0.936,9); Line */
    PT_MU_WOT_V = 18,   /* 2 x 9 ; n in range(0.667,1,2), delta in range(0.441,1,9); Line */
    PT_PISTD_N_K = 36,  /* 13 */
    PT_PISTD_MU_K = 49, /* 3 */
    PT_PISTD_V = 52,    /* 13 x 3 ; Flat */
    PT_PIWOT_N_K = 91,  /* 5 */
    PT_PIWOT_D_K = 96,  /* 3 */
    PT_PIWOT_V = 99,    /* 5 x 3 ; n Flat, delta Flat below / Line above */
    PT_F_K = 114,       /* 11 knots: fuel-to-air ratio */
    PT_PI_RATIO_V = 125,  /* 11 ; Flat */
    PT_SFC_RATIO_V = 136, /* 11 ; Flat */
    PT_SFC_N_K = 147,   /* 5 */
    PT_SFC_PI_K = 152,  /* 8 */
    PT_SFC_POW_V = 160, /* 5 x 8 ; Line */
    PT_SIZE = 200
};

/* ---------------- propeller table (propellers.jl:235-276) ----------------
 * host side: [21 x 21 x 6] column-major (J, Mt, coefficient) as the reference's Coefficients{Array}
 * device side: interleaved, element (iJ, iMt, c) at ((iJ + 21*iMt) * 6 + c)                      */
enum { PR_NJ = 21, PR_NM = 21, PR_NC = 6, PR_SIZE = 21 * 21 * 6 };
/* kernels that do not log C_P / eta_p keep only the first four coefficients in LDS (7 KB less per workgroup) */
enum { PR_NC_STEP = 4, PR_SIZE_STEP = 21 * 21 * 4 };

/* LDS blob = [aero | piston | propeller] */
enum { LDS_AERO = 0, LDS_PISTON = AT_SIZE, LDS_PROP = AT_SIZE + PT_SIZE, LDS_TABLE_DOUBLES = AT_SIZE + PT_SIZE + PR_SIZE,
       LDS_TABLE_DOUBLES_STEP = AT_SIZE + PT_SIZE + PR_SIZE_STEP };

/* Behind the blob in the device buffer: contiguous copies of the knots the two-level scans (grid_locate) compare from SGPRs —
 * per table AUX_STRIDE doubles [k_0, k_{N-1}, k_S, k_2S, ..., k_GS, +inf ...], S = 4 for N >= 20 else 3, G = (N-2)/S — so that one
 * scalar load brings what six strided ones did. Written by the host whenever the aero / piston blob is uploaded. */
enum { AUX_GE = 0, AUX_AL26 = 1, AUX_AL17 = 2, AUX_N13 = 3, AUX_F11 = 4, AUX_TABLES = 5, AUX_STRIDE = 8,
       LDS_AUX = LDS_TABLE_DOUBLES, TABLE_BUF_DOUBLES = LDS_TABLE_DOUBLES + AUX_TABLES * AUX_STRIDE };

#endif
