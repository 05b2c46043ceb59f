// fb_capi.hip — C ABI of libflightbatch (include/flightbatch.h) over the gfx950 kernels.
// One handle = one HIP device + one stream + SoA device buffers. No CPU fallback: every compute entry
// point launches a kernel; fb_create fails loudly when no HIP device is present.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "c172_kernels.hpp"
#include "c172x_kernels.hpp"
#include "c172_kernels_f32.hpp"
#include "robot2d_kernels.hpp"
#include "scenario_kernels.hpp"

using namespace fbd;

static thread_local std::string g_err;
template <class... A>
static int32_t fail(const char* fmt, A... args) {
    char buf[512];
    if constexpr (sizeof...(A) == 0) std::snprintf(buf, sizeof buf, "%s", fmt);
    else std::snprintf(buf, sizeof buf, fmt, args...);
    g_err = buf;
    return -1;
}
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char b_[512];                                                                   \
            std::snprintf(b_, sizeof b_, "%s failed: %s", #expr, hipGetErrorString(e_));    \
            g_err = b_;                                                                     \
            return -2;                                                                      \
        }                                                                                   \
    } while (0)

struct fb_handle_s {
    int32_t model = 0, kin = 0, dtype = 0, device = 0;
    int64_t n = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    double *x_own = nullptr, *x = nullptr;
    int32_t *s_own = nullptr, *s = nullptr;
    double* u = nullptr;
    int32_t* ui = nullptr;
    int32_t* status = nullptr;
    long long* term_step = nullptr;   // [n] termination record (fb_get_termination), valid where status != 0
    int32_t* term_where = nullptr;    // [n]
    double* y = nullptr;      // [FB_NY x n], allocated on first use
    double* xdot = nullptr;   // scratch [FB_NX x n]
    double* tables = nullptr; // LDS_TABLE_DOUBLES
    float* tables_f32 = nullptr;  // fp32 mirror of `tables` (FB_F32 handles), refreshed when a table is uploaded
    bool tables_f32_stale = true;
    float* egm96 = nullptr;
    double* trim_buf = nullptr;  // tp | ts | cost
    int32_t* trim_ok = nullptr;
    double* env_rows = nullptr;  // [ENV_DEV_ROWS x n] per-aircraft environment (fb_set_env), null: the batch-wide fb_params block
    double* trim_ws = nullptr;   // k_trim's workspace: TRIM_WS_ROWS rows per resident lane, allocated by the first fb_trim and kept until fb_destroy (~100 MB on 256 CUs)
    bool have_table[4] = {false, false, false, false};
    fb_params params;
    int32_t steps_per_launch = 1;
    bool duo = false;   // the wave-specialised airborne stepper (see env_step_duo)
    double t = 0.0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false;
    int64_t launches = 0;
    std::vector<hipEvent_t> lev;   // fb_timing_begin_per_launch: one event pair per stepping launch (both passes), created on demand and kept
    int64_t lev_max = 0, lev_used = 0;   // pairs wanted in this timing window / pairs recorded
    struct R2State* r2 = nullptr;  // Robot2D handles only
    // Cessna172Xv2 handles only
    double* cs = nullptr;      // [FB_NCS x n] control-law record
    double* cu = nullptr;      // [FB_NCU x n] control-law inputs
    double* q_pre = nullptr;   // [8 x n]
    double* ctl_bak = nullptr; // [(FB_NCS + FB_NCU) x n] scratch of the airborne pass (see KArgs::ctl_bak)
    double* duo_pld = nullptr; // [DUO_NCONST x n] scratch of k_step_duo (see KArgs::duo_pld)
    double* duo_tap = nullptr; // [DUO_NTAP x n] Cessna172Xv2 on k_step_duo: the hand-over rows of a control update (KArgs::duo_tap)
    int32_t* redo = nullptr;   // [n] hand-over flags between the two passes of the stepping kernel
    double* k1 = nullptr;      // [FB_NX x n] Cessna172Xv2: FSAL derivative carried from launch to launch
    int32_t* k1_valid = nullptr;
    double* gains = nullptr;   // FB_TABLE_CTL_GAINS blob
    int64_t gains_off[10] = {0};
    int64_t gains_total = 0;
    bool gains_same_grid = false;   // all ten lookups on one (EAS, h) grid: CtlOffsets::same_grid
    bool have_gains = false;
    int64_t steps_done = 0;    // steps since the last init (phase of the periodic update)
    struct LogState* log = nullptr;  // on-device TimeSeries log (fb_log_*)
    // scripted scenario (FB_TABLE_SCENARIO): the program blob and the per-aircraft rows, all in device memory
    double* scn_prog = nullptr; int scn_nph = 0, scn_nrule = 0, scn_nact = 0, scn_npar = 0, scn_nrec = 0, scn_every = 0;
    int32_t* scn_phase = nullptr; long long* scn_since = nullptr; double* scn_par = nullptr; double* scn_rec = nullptr;
};

static KArgs make_args(fb_handle h) {
    KArgs a;
    a.x = h->x; a.s = h->s; a.u = h->u; a.ui = h->ui; a.status = h->status; a.tables = h->tables; a.tables_f32 = h->tables_f32; a.egm96 = h->egm96;
    a.n = h->n;
    a.env = {h->params.T_sl, h->params.p_sl, h->params.wind_ned[0], h->params.wind_ned[1], h->params.wind_ned[2], h->params.h_terrain, h->params.surface,
             log(h->params.p_sl / 101325.0),
             exp(0.5 * 6.5e-3 * 287.05287 / 9.80665 * log(h->params.p_sl / 101325.0)) / sqrt(h->params.T_sl)};
    a.env_rows = h->env_rows;
    a.dt = h->params.dt;
    a.cs = h->cs; a.cu = h->cu; a.q_pre = h->q_pre; a.redo = h->redo; a.k1 = h->k1; a.k1_valid = h->k1_valid;
    if (getenv("FB_NO_FSAL_CARRY")) a.k1 = nullptr;   // A/B switch for measurements
    a.gains = h->gains; a.ctl_bak = h->ctl_bak; a.duo_pld = h->duo_pld; a.duo_tap = h->duo_tap;
    a.term_step = h->term_step; a.term_where = h->term_where; a.step0 = h->steps_done;
    for (int k = 0; k < 10; k++) a.ctl_off.off[k] = (int)h->gains_off[k];
    a.ctl_off.total = (int)h->gains_total;
    a.ctl_off.same_grid = h->gains_same_grid ? 1 : 0;
    const int ratio = h->params.periodic_n > 0 ? h->params.periodic_n : 1;
    a.ctl_dT = h->params.dt * ratio;
    a.ctl_ratio = h->model == FB_MODEL_C172X2 ? ratio : 0;
    a.ctl_phase = (int)(h->steps_done % ratio);
    return a;
}
static int32_t check_ready(fb_handle h) {
    if (!h) return fail("null handle");
    for (int k = 0; k < 4; k++)
        if (!h->have_table[k]) {
            static const char* names[4] = {"EGM96", "PROPELLER", "PISTON", "AERO"};
            return fail("table %s has not been uploaded (fb_set_table)", names[k]);
        }
    return 0;
}
static dim3 grid_for(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }
// init! clears terminations (FC/sim.jl:390-414): the status words AND the termination record go together, so that a record can never
// outlive the exception it describes
static hipError_t clear_terminations(fb_handle h) {
    if (hipError_t e = hipMemsetAsync(h->status, 0, sizeof(int32_t) * h->n, h->stream)) return e;
    if (hipError_t e = hipMemsetAsync(h->term_step, 0, sizeof(long long) * h->n, h->stream)) return e;
    return hipMemsetAsync(h->term_where, 0, sizeof(int32_t) * h->n, h->stream);
}
// fb_set_status: an aircraft whose word the host makes non-zero without a record gets (step count, FB_TERM_OUTSIDE_STEP); one whose word
// is cleared loses its record
__global__ void k_mark_host_status(const int32_t* status, long long* term_step, int32_t* term_where, long long step0, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool term = (status[i] & ~FB_ST_NAN) != 0;
    if (!term) { term_where[i] = FB_TERM_NONE; term_step[i] = 0; }
    else if (term_where[i] == FB_TERM_NONE) { term_where[i] = FB_TERM_OUTSIDE_STEP; term_step[i] = step0; }
}
static bool is_x2(fb_handle h) { return h->model == FB_MODEL_C172X2; }
// anything that changes x, s, u or the environment from outside the stepping kernel invalidates the carried FSAL derivative
static void fsal_invalidate(fb_handle h) {
    if (!h->k1_valid) return;
    (void)hipSetDevice(h->device);
    (void)hipMemsetAsync(h->k1_valid, 0, sizeof(int32_t) * h->n, h->stream);
}
// states of the C ABI: 27 (WA), 26 (ECEF: q_eb[4] n_e[3] h_e), 24 (NED: ψ θ φ ϕ λ h_e), Cessna172Xv2 seven more (34 / 33 / 31); the device keeps
// 27 (34) rows for every mechanisation, the unused kinematic rows stay zero
static int nx_of(fb_handle h) { return (is_x2(h) ? (int)FB_X2_NX : (int)FB_NX) - (h->kin == FB_KIN_ECEF ? 1 : (h->kin == FB_KIN_NED ? 3 : 0)); }
static int ecef_dev_row(int k) { return k < FB_X_Q_WB + 8 ? k : k + 1; }
static int ned_dev_row(int k) { return k < FB_X_Q_WB + 6 ? k : k + 3; }
// Cessna172X: row of the C ABI state layout (reference order: act after pwp) -> device row (actuators last)
static int x2_dev_row(int k) { return k < FB_X2_ACT ? k : (k < FB_X2_KIN ? FB_NX + (k - FB_X2_ACT) : k - FB_NACT); }
static int x2_ecef_dev_row(int k) { return k < FB_X2_KIN ? x2_dev_row(k) : ecef_dev_row(k - FB_NACT); }
static int x2_ned_dev_row(int k) { return k < FB_X2_KIN ? x2_dev_row(k) : ned_dev_row(k - FB_NACT); }
typedef int (*row_map_t)(int);
static row_map_t row_map_of(fb_handle h);
static int32_t check_ready_x2(fb_handle h) {
    if (int32_t rc = check_ready(h)) return rc;
    if (is_x2(h) && !h->have_gains) return fail("table CTL_GAINS has not been uploaded (fb_set_table)");
    return 0;
}
static row_map_t row_map_of(fb_handle h) {
    if (is_x2(h)) return h->kin == FB_KIN_ECEF ? x2_ecef_dev_row : (h->kin == FB_KIN_NED ? x2_ned_dev_row : x2_dev_row);
    if (h->kin == FB_KIN_ECEF) return ecef_dev_row;
    if (h->kin == FB_KIN_NED) return ned_dev_row;
    return nullptr;
}
// one launch statement per (model, kinematics) instance of a kernel template
#define FB_LAUNCH_MK(KERNEL, GRID, BLOCK, ...)                                                                                       \
    do {                                                                                                                              \
        if (is_x2(h) && h->kin == FB_KIN_ECEF) hipLaunchKernelGGL((KERNEL<true, FB_KIN_ECEF>), GRID, BLOCK, 0, h->stream, __VA_ARGS__); \
        else if (is_x2(h) && h->kin == FB_KIN_NED) hipLaunchKernelGGL((KERNEL<true, FB_KIN_NED>), GRID, BLOCK, 0, h->stream, __VA_ARGS__); \
        else if (is_x2(h)) hipLaunchKernelGGL((KERNEL<true, FB_KIN_WA>), GRID, BLOCK, 0, h->stream, __VA_ARGS__);                     \
        else if (h->kin == FB_KIN_ECEF) hipLaunchKernelGGL((KERNEL<false, FB_KIN_ECEF>), GRID, BLOCK, 0, h->stream, __VA_ARGS__);     \
        else if (h->kin == FB_KIN_NED) hipLaunchKernelGGL((KERNEL<false, FB_KIN_NED>), GRID, BLOCK, 0, h->stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL((KERNEL<false, FB_KIN_WA>), GRID, BLOCK, 0, h->stream, __VA_ARGS__);                                  \
    } while (0)
// FLIGHTBATCH_DUO (read when a handle is created; A/B switch for measurements): Cessna172Sv0 and Cessna172Xv2 in fp64 (any mechanisation) are stepped
// by the wave-specialised k_step_duo<KIN, X> (two waves per SIMD) unless it is 0, which selects the one-wave-per-SIMD k_step_air<KIN, X>
static bool env_step_duo() { const char* e = getenv("FLIGHTBATCH_DUO"); return e ? atoi(e) != 0 : true; }
// the two passes of the stepping kernel (airborne instance, then the ground-capable one over the lanes it handed over)
#define FB_STEP_PERENV(KIN, X, GRID, A, K)                                                                                            \
    do {   /* per-aircraft environment rows (KArgs::env_rows): the wave-pair kernel in every mechanisation (FB_F32 handles: the one-wave fp64 kernel) */ \
        if (h->duo) hipLaunchKernelGGL((k_step_duo<KIN, X, true>), grid_for(h->n, DUO_B), dim3(2 * DUO_B), 0, h->stream, A, K);         \
        else hipLaunchKernelGGL((k_step_air<KIN, X, false, true>), GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                      \
        hipLaunchKernelGGL((k_step_air<KIN, X, true, true>), grid_for(h->n, step_block<X, true>()), dim3(step_block<X, true>()), 0, h->stream, A, K); \
    } while (0)
#define FB_STEP_X2(KIN, GRID, A, K)                                                                                                   \
    do {                                                                                                                              \
        if (h->env_rows) { FB_STEP_PERENV(KIN, true, GRID, A, K); break; }                                                            \
        if (h->duo) hipLaunchKernelGGL((k_step_duo<KIN, true>), grid_for(h->n, DUO_B), dim3(2 * DUO_B), 0, h->stream, A, K);         \
        else hipLaunchKernelGGL((k_step_air<KIN, true>), GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                                \
        hipLaunchKernelGGL((k_step_air<KIN, true, true>), grid_for(h->n, step_block<true, true>()), dim3(step_block<true, true>()), 0, h->stream, A, K); \
    } while (0)
// the Cessna172Xv2 kernels that are not stepping kernels take the mechanisation alone
#define FB_LAUNCH_X2K(KERNEL, N, ...)                                                                                                 \
    do {                                                                                                                              \
        if (h->kin == FB_KIN_ECEF) hipLaunchKernelGGL(KERNEL<FB_KIN_ECEF>, grid_for(N, 256), dim3(256), 0, h->stream, __VA_ARGS__);  \
        else if (h->kin == FB_KIN_NED) hipLaunchKernelGGL(KERNEL<FB_KIN_NED>, grid_for(N, 256), dim3(256), 0, h->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<FB_KIN_WA>, grid_for(N, 256), dim3(256), 0, h->stream, __VA_ARGS__);                          \
    } while (0)
#define FB_LAUNCH_STEP(GRID, A, K)                                                                                                    \
    do {                                                                                                                              \
        if (is_x2(h) && h->kin == FB_KIN_ECEF) FB_STEP_X2(FB_KIN_ECEF, GRID, A, K);                                                   \
        else if (is_x2(h) && h->kin == FB_KIN_NED) FB_STEP_X2(FB_KIN_NED, GRID, A, K);                                                \
        else if (is_x2(h)) FB_STEP_X2(FB_KIN_WA, GRID, A, K);                                                                         \
        else if (h->env_rows && h->kin == FB_KIN_ECEF) FB_STEP_PERENV(FB_KIN_ECEF, false, GRID, A, K);                                \
        else if (h->env_rows && h->kin == FB_KIN_NED) FB_STEP_PERENV(FB_KIN_NED, false, GRID, A, K);                                  \
        else if (h->env_rows) FB_STEP_PERENV(FB_KIN_WA, false, GRID, A, K);   /* (FB_F32 handles too: the fp32 stepper is batch-wide only) */ \
        else if (h->kin == FB_KIN_ECEF) {                                                                                           \
            if (h->duo) hipLaunchKernelGGL(k_step_duo<FB_KIN_ECEF>, grid_for(h->n, DUO_B), dim3(2 * DUO_B), 0, h->stream, A, K);  \
            else hipLaunchKernelGGL(k_step_air<FB_KIN_ECEF>, GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                            \
            hipLaunchKernelGGL((k_step_air<FB_KIN_ECEF, false, true>), grid_for(h->n, step_block<false, true>()), dim3(step_block<false, true>()), 0, h->stream, A, K);                  \
        } else if (h->kin == FB_KIN_NED) {                                                                                            \
            if (h->duo) hipLaunchKernelGGL(k_step_duo<FB_KIN_NED>, grid_for(h->n, DUO_B), dim3(2 * DUO_B), 0, h->stream, A, K);   \
            else hipLaunchKernelGGL(k_step_air<FB_KIN_NED>, GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                             \
            hipLaunchKernelGGL((k_step_air<FB_KIN_NED, false, true>), grid_for(h->n, step_block<false, true>()), dim3(step_block<false, true>()), 0, h->stream, A, K);                   \
        } else if (h->dtype == FB_F32) {   /* fp32 airborne stepper; lanes near the ground go to the fp64 ground-capable kernel */    \
            hipLaunchKernelGGL(fbf::k_step_f32, GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                                         \
            hipLaunchKernelGGL((k_step_air<FB_KIN_WA, false, true>), grid_for(h->n, step_block<false, true>()), dim3(step_block<false, true>()), 0, h->stream, A, K);                    \
        } else {                                                                                                                      \
            if (h->duo) hipLaunchKernelGGL(k_step_duo<FB_KIN_WA>, grid_for(h->n, DUO_B), dim3(2 * DUO_B), 0, h->stream, A, K);    \
            else hipLaunchKernelGGL(k_step_air<FB_KIN_WA>, GRID, dim3(STEP_BLOCK), 0, h->stream, A, K);                              \
            hipLaunchKernelGGL((k_step_air<FB_KIN_WA, false, true>), grid_for(h->n, step_block<false, true>()), dim3(step_block<false, true>()), 0, h->stream, A, K);                    \
        }                                                                                                                             \
    } while (0)
static CtlArgs ctl_args(fb_handle h, int use_q_pre) {
    CtlArgs c;
    c.gains = h->gains;
    for (int k = 0; k < 10; k++) c.off.off[k] = (int)h->gains_off[k];
    c.off.total = (int)h->gains_total;
    c.off.same_grid = h->gains_same_grid ? 1 : 0;
    c.dT = h->params.dt * (h->params.periodic_n > 0 ? h->params.periodic_n : 1);
    c.use_q_pre = use_q_pre;
    return c;
}
// host [nrows x n] <-> device rows through a row map (identity when map == nullptr)
static int32_t copy_rows(fb_handle h, double* dev, const double* host_in, double* host_out, int nrows, int (*map)(int)) {
    const int64_t n = h->n;
    for (int k = 0; k < nrows; k++) {
        double* d = dev + (int64_t)(map ? map(k) : k) * n;
        if (host_in) HIPCHK(hipMemcpyAsync(d, host_in + (int64_t)k * n, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
        if (host_out) HIPCHK(hipMemcpyAsync(host_out + (int64_t)k * n, d, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    }
    return 0;
}

#include "fb_robot2d.inc"
#include "fb_log.inc"
struct ncclUniqueIdBlob { char internal[128]; };   // ncclUniqueId (rccl.h:40-43), passed by value

// The scratch rows of the stepping kernels (ctl_bak, duo_pld, duo_tap) are zeroed when a handle is created: hipMalloc hands out zero pages in
// a fresh process and whatever the previous owner left in a block that is reused, and nothing a launch does — its time included — may depend
// on which (docs/design/roofline.md, "Round 6": what the driver's round-5 bench run measured). FLIGHTBATCH_SCRATCH_FILL is the diagnostic
// behind that paragraph: a byte value fills the rows with that byte instead (255: NaNs, 127: 1.4e306), "none" leaves them as allocated.
static int scratch_fill() {
    const char* e = getenv("FLIGHTBATCH_SCRATCH_FILL");
    if (!e || !*e) return 0;
    if (!strcmp(e, "none")) return -1;
    return atoi(e) & 255;
}
// device-side resources of a new handle; on failure fb_create destroys the partially built handle (nothing leaks)
static int32_t create_resources(fb_handle h, int32_t model_id, int32_t dtype, int64_t n) {
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    HIPCHK(hipEventCreate(&h->ev0));
    HIPCHK(hipEventCreate(&h->ev1));
    if (model_id == FB_MODEL_ROBOT2D) {
        return r2_create(h, dtype);
    }
    const int nx = is_x2(h) ? (int)FB_X2_NX : (int)FB_NX;   // device rows
    const int sf = scratch_fill();
    HIPCHK(hipMalloc(&h->x_own, sizeof(double) * nx * n));
    HIPCHK(hipMalloc(&h->s_own, sizeof(int32_t) * FB_NS * n));
    if (is_x2(h)) {
        HIPCHK(hipMalloc(&h->cs, sizeof(double) * FB_NCS * n));
        HIPCHK(hipMalloc(&h->cu, sizeof(double) * FB_NCU * n));
        HIPCHK(hipMalloc(&h->q_pre, sizeof(double) * 8 * n));
        HIPCHK(hipMalloc(&h->ctl_bak, sizeof(double) * (FB_NCS + FB_NCU) * n));
        if (sf >= 0) HIPCHK(hipMemsetAsync(h->ctl_bak, sf, sizeof(double) * (FB_NCS + FB_NCU) * n, h->stream));   // (scratch rows start defined: see scratch_fill())
        HIPCHK(hipMemsetAsync(h->cs, 0, sizeof(double) * FB_NCS * n, h->stream));
        HIPCHK(hipMemsetAsync(h->cu, 0, sizeof(double) * FB_NCU * n, h->stream));
        HIPCHK(hipMemsetAsync(h->q_pre, 0, sizeof(double) * 8 * n, h->stream));
        HIPCHK(hipMalloc(&h->k1, sizeof(double) * FB_NX * n));
        HIPCHK(hipMalloc(&h->k1_valid, sizeof(int32_t) * n));
        HIPCHK(hipMemsetAsync(h->k1_valid, 0, sizeof(int32_t) * n, h->stream));
    }
    HIPCHK(hipMalloc(&h->u, sizeof(double) * FB_NU * n));
    HIPCHK(hipMalloc(&h->ui, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&h->status, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&h->term_step, sizeof(long long) * n));
    HIPCHK(hipMalloc(&h->term_where, sizeof(int32_t) * n));
    HIPCHK(hipMemsetAsync(h->term_step, 0, sizeof(long long) * n, h->stream));
    HIPCHK(hipMemsetAsync(h->term_where, 0, sizeof(int32_t) * n, h->stream));
    HIPCHK(hipMalloc(&h->redo, sizeof(int32_t) * n));
    if (h->duo) {   // (here, not in the first fb_step: hipMalloc synchronises the device)
        HIPCHK(hipMalloc(&h->duo_pld, sizeof(double) * DUO_NCONST * n));
        if (sf >= 0) HIPCHK(hipMemsetAsync(h->duo_pld, sf, sizeof(double) * DUO_NCONST * n, h->stream));
    }
    if (h->duo && model_id == FB_MODEL_C172X2) {
        HIPCHK(hipMalloc(&h->duo_tap, sizeof(double) * DUO_NTAP * n));
        if (sf >= 0) HIPCHK(hipMemsetAsync(h->duo_tap, sf, sizeof(double) * DUO_NTAP * n, h->stream));
    }
    HIPCHK(hipMemsetAsync(h->redo, 0, sizeof(int32_t) * n, h->stream));
    HIPCHK(hipMalloc(&h->tables, sizeof(double) * TABLE_BUF_DOUBLES));
    HIPCHK(hipMalloc(&h->egm96, sizeof(float) * 721 * 1441));
    h->x = h->x_own; h->s = h->s_own;
    HIPCHK(hipMemsetAsync(h->x, 0, sizeof(double) * nx * n, h->stream));
    HIPCHK(hipMemsetAsync(h->s, 0, sizeof(int32_t) * FB_NS * n, h->stream));
    HIPCHK(hipMemsetAsync(h->u, 0, sizeof(double) * FB_NU * n, h->stream));
    HIPCHK(hipMemsetAsync(h->status, 0, sizeof(int32_t) * n, h->stream));
    {
        std::vector<int32_t> ui((size_t)n, FB_UI_DEFAULT);
        HIPCHK(hipMemcpyAsync(h->ui, ui.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// ---- scripted scenarios (FB_TABLE_SCENARIO, include/flightbatch.h; kernel: scenario_kernels.hpp) ----
static void scn_free(fb_handle h) {
    hipFree(h->scn_prog); hipFree(h->scn_phase); hipFree(h->scn_since); hipFree(h->scn_par); hipFree(h->scn_rec);
    h->scn_prog = nullptr; h->scn_phase = nullptr; h->scn_since = nullptr; h->scn_par = nullptr; h->scn_rec = nullptr;
    h->scn_nph = h->scn_nrule = h->scn_nact = h->scn_npar = h->scn_nrec = h->scn_every = 0;
}
// every index the kernel will follow is checked HERE, on the host: a table is data from outside, and an out-of-range row on the device is a fault
static int32_t scn_load(fb_handle h, const double* b, int64_t len) {
    if (!is_x2(h)) return fail("FB_TABLE_SCENARIO: scenarios drive the Cessna172Xv2's inputs (control-law inputs, vehicle inputs); this handle is another model");
    if (len < FB_SCN_HDR) return fail("scenario blob: %lld doubles, shorter than its header", (long long)len);
    if (b[0] != 5000001.0) return fail("scenario blob: unknown layout version %g", b[0]);
    auto as_int = [&](double v, int64_t lo, int64_t hi, const char* what, int64_t* out) -> bool {
        if (!(v >= (double)lo && v <= (double)hi) || v != floor(v)) { fail("scenario blob: %s = %g is not an integer in [%lld, %lld]", what, v, (long long)lo, (long long)hi); return false; }
        *out = (int64_t)v; return true;
    };
    int64_t nph, nrule, nact, npar, nrec;
    if (!as_int(b[1], 1, 4096, "n_phase", &nph) || !as_int(b[2], 0, 65536, "n_rule", &nrule) || !as_int(b[3], 0, 65536, "n_act", &nact) ||
        !as_int(b[4], 0, 4096, "n_par", &npar) || !as_int(b[5], 0, 4096, "n_rec", &nrec)) return -1;
    const int64_t need = FB_SCN_HDR + FB_SCN_PHASE_REC * nph + FB_SCN_RULE_REC * nrule + FB_SCN_ACT_REC * nact;
    if (len != need) return fail("scenario blob: %lld doubles given, its header needs %lld", (long long)len, (long long)need);
    const double* PH = b + FB_SCN_HDR; const double* RU = PH + FB_SCN_PHASE_REC * nph; const double* AC = RU + FB_SCN_RULE_REC * nrule;
    const int nx_dev = (int)FB_X2_NX;
    auto src_ok = [&](double kind, double row, const char* where, int64_t idx) -> bool {
        int64_t k, r;
        if (!as_int(kind, 0, FB_SCN_NSRC - 1, "a source kind", &k)) return false;
        const int64_t lim = k == FB_SCN_SRC_X ? nx_dev : k == FB_SCN_SRC_CS ? (int64_t)FB_NCS : k == FB_SCN_SRC_CU ? (int64_t)FB_NCU : k == FB_SCN_SRC_U ? (int64_t)FB_NU :
                            k == FB_SCN_SRC_S ? (int64_t)FB_NS : k == FB_SCN_SRC_PAR ? npar : k == FB_SCN_SRC_REC ? nrec : (int64_t)1 << 30;
        if (!as_int(row, 0, lim - 1, "a source row", &r)) { g_err += std::string(" (") + where + " " + std::to_string(idx) + ")"; return false; }
        return true;
    };
    for (int64_t p = 0; p < nph; p++) {
        int64_t a0, na, r0, nr;
        if (!as_int(PH[4 * p], 0, nact, "a phase's first action", &a0) || !as_int(PH[4 * p + 1], 0, nact - a0, "a phase's action count", &na) ||
            !as_int(PH[4 * p + 2], 0, nrule, "a phase's first rule", &r0) || !as_int(PH[4 * p + 3], 0, nrule - r0, "a phase's rule count", &nr)) return -1;
    }
    for (int64_t r = 0; r < nrule; r++) {
        const double* ru = RU + FB_SCN_RULE_REC * r;
        int64_t v, f;
        if (!src_ok(ru[0], ru[1], "rule", r)) return -1;
        if (!as_int(ru[2], 0, FB_SCN_ALWAYS, "a comparison", &v) || !as_int(ru[4], -1, npar - 1, "a rule's parameter row", &v) ||
            !as_int(ru[5], 0, nact, "a rule's first action", &f) || !as_int(ru[6], 0, nact - f, "a rule's action count", &v) ||
            !as_int(ru[7], 0, nph - 1, "a rule's next phase", &v)) return -1;
        if (!std::isfinite(ru[3])) return fail("scenario blob: rule %lld compares with a non-finite constant", (long long)r);
    }
    for (int64_t k = 0; k < nact; k++) {
        const double* ac = AC + FB_SCN_ACT_REC * k;
        int64_t dst, row, nt, v;
        if (!as_int(ac[0], 0, FB_SCN_NDST - 1, "a destination kind", &dst)) return -1;
        const int64_t lim = dst == FB_SCN_DST_CU ? (int64_t)FB_NCU : dst == FB_SCN_DST_U ? (int64_t)FB_NU : dst == FB_SCN_DST_REC ? nrec : (int64_t)1 << 30;
        if (!as_int(ac[1], 0, lim - 1, "a destination row", &row) || !as_int(ac[2], 0, 1, "a wrap flag", &v) || !as_int(ac[4], 0, FB_SCN_NTERM, "a term count", &nt)) return -1;
        for (int64_t t = 0; t < nt; t++) if (!src_ok(ac[5 + 3 * t], ac[6 + 3 * t], "action", k)) return -1;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    scn_free(h);
    const int64_t n = h->n;
    HIPCHK(hipMalloc(&h->scn_prog, sizeof(double) * len));
    HIPCHK(hipMemcpy(h->scn_prog, b, sizeof(double) * len, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->scn_phase, sizeof(int32_t) * n));
    HIPCHK(hipMalloc(&h->scn_since, sizeof(long long) * n));
    HIPCHK(hipMalloc(&h->scn_par, sizeof(double) * (npar > 0 ? npar : 1) * n));
    HIPCHK(hipMalloc(&h->scn_rec, sizeof(double) * (nrec > 0 ? nrec : 1) * n));
    HIPCHK(hipMemset(h->scn_phase, 0, sizeof(int32_t) * n));
    HIPCHK(hipMemset(h->scn_since, 0, sizeof(long long) * n));
    HIPCHK(hipMemset(h->scn_par, 0, sizeof(double) * (npar > 0 ? npar : 1) * n));
    HIPCHK(hipMemset(h->scn_rec, 0, sizeof(double) * (nrec > 0 ? nrec : 1) * n));
    h->scn_nph = (int)nph; h->scn_nrule = (int)nrule; h->scn_nact = (int)nact; h->scn_npar = (int)npar; h->scn_nrec = (int)nrec; h->scn_every = 1;
    return 0;
}
// one evaluation of the table for every aircraft, behind the step that has just completed (steps_done counts it)
static int32_t scn_evaluate(fb_handle h) {
    ScnArgs sc;
    sc.prog = h->scn_prog; sc.n_ph = h->scn_nph; sc.n_rule = h->scn_nrule; sc.n_act = h->scn_nact; sc.n_par = h->scn_npar; sc.n_rec = h->scn_nrec;
    sc.phase = h->scn_phase; sc.since = h->scn_since; sc.par = h->scn_par; sc.rec = h->scn_rec;
    sc.step = h->steps_done; sc.dt = h->params.dt; sc.t = (double)h->steps_done * h->params.dt;   // sim.t = t_start + nstep dt with t_start = 0 (FC/sim.jl:261-275)
    FB_LAUNCH_X2K(k_scenario, h->n, make_args(h), sc);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" {

const char* fb_last_error(void) { return g_err.c_str(); }
const char* fb_version(void) { return "flightbatch 0.1 (gfx950)"; }

int32_t fb_create(int32_t model_id, int32_t kin_id, int32_t dtype, int64_t n, int32_t device_id, fb_handle* out) {
    if (!out) return fail("out is null");
    *out = nullptr;
    if (model_id != FB_MODEL_C172S0 && model_id != FB_MODEL_C172X2 && model_id != FB_MODEL_ROBOT2D) return fail("unknown model id");
    if ((model_id == FB_MODEL_C172S0 || model_id == FB_MODEL_C172X2) && kin_id != FB_KIN_WA && kin_id != FB_KIN_ECEF && kin_id != FB_KIN_NED) return fail("unknown kinematics id");
    if (model_id == FB_MODEL_C172X2 && dtype != FB_F64) return fail("Cessna172Xv2: only FB_F64 is implemented");
    if (model_id == FB_MODEL_C172S0 && dtype == FB_F32 && kin_id != FB_KIN_WA) return fail("Cessna172Sv0 in fp32: only FB_KIN_WA is implemented");
    if (dtype != FB_F64 && dtype != FB_F32) return fail("unknown dtype");
    if (n <= 0) return fail("n must be positive");
    if (device_id < 0) return fail("device_id < 0: libflightbatch has no CPU backend");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: libflightbatch requires a GPU");
    if (device_id >= ndev) return fail("device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    fb_handle h = new fb_handle_s();
    // the wave-specialised stepper exists for Cessna172Sv0 and Cessna172Xv2 in fp64 (any mechanisation); every other handle steps with k_step_air and needs no duo_pld
    h->duo = (model_id == FB_MODEL_C172S0 || model_id == FB_MODEL_C172X2) && dtype == FB_F64 && env_step_duo();
    h->model = model_id; h->kin = kin_id; h->dtype = dtype; h->device = device_id; h->n = n;
    h->params.dt = 0.02; h->params.periodic_n = 1; h->params.surface = 0;
    h->params.T_sl = isa::T_std; h->params.p_sl = isa::p_std;
    h->params.wind_ned[0] = h->params.wind_ned[1] = h->params.wind_ned[2] = 0.0;
    h->params.h_terrain = 0.0;
    if (int32_t rc = create_resources(h, model_id, dtype, n)) {
        const std::string msg = g_err;   // fb_destroy must not clobber the reason
        fb_destroy(h);
        g_err = msg;
        return rc;
    }
    *out = h;
    return 0;
}
int32_t fb_destroy(fb_handle h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    log_free(h);
    scn_free(h);
    r2_destroy(h);
    hipFree(h->x_own); hipFree(h->s_own); hipFree(h->u); hipFree(h->ui); hipFree(h->status); hipFree(h->term_step); hipFree(h->term_where); hipFree(h->y); hipFree(h->xdot);
    hipFree(h->tables); hipFree(h->tables_f32); hipFree(h->egm96); hipFree(h->trim_buf); hipFree(h->trim_ok); hipFree(h->trim_ws); hipFree(h->env_rows);
    hipFree(h->cs); hipFree(h->cu); hipFree(h->q_pre); hipFree(h->ctl_bak); hipFree(h->duo_pld); hipFree(h->duo_tap); hipFree(h->gains); hipFree(h->redo); hipFree(h->k1); hipFree(h->k1_valid);
    hipEventDestroy(h->ev0); hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->lev) hipEventDestroy(e);
    hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}
int64_t fb_size(fb_handle h) { return h ? h->n : -1; }
int32_t fb_dims(fb_handle h, int32_t* nx, int32_t* ns, int32_t* nu, int32_t* ny) {
    if (!h) return fail("null handle");
    const bool r2 = h->model == FB_MODEL_ROBOT2D;
    if (nx) *nx = r2 ? (int)FB_R2_NX : nx_of(h);
    if (ns) *ns = r2 ? 0 : FB_NS;
    if (nu) *nu = r2 ? FB_R2_NU : FB_NU;
    if (ny) *ny = r2 ? FB_R2_NY : FB_NY;
    return 0;
}

int32_t fb_set_stream(fb_handle h, void* hip_stream) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return 0;
}
int32_t fb_attach_state(fb_handle h, void* x_dev, void* s_dev) {
    if (h) fsal_invalidate(h);
    if (!h) return fail("null handle");
    if (h->model == FB_MODEL_ROBOT2D) return fail("fb_attach_state: not supported for Robot2D");
    if ((x_dev == nullptr) != (s_dev == nullptr)) return fail("x_dev and s_dev must both be given or both be NULL");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->x = x_dev ? (double*)x_dev : h->x_own;
    h->s = s_dev ? (int32_t*)s_dev : h->s_own;
    return 0;
}

// the coarse-knot copy of one table behind the blob (csrc/tables.h: LDS_AUX)
static int32_t put_coarse_knots(fb_handle h, int slot, const double* k, int n) {
    const int S = n >= 20 ? 4 : 3, G = (n - 2) / S;
    double buf[AUX_STRIDE];
    for (int e = 0; e < AUX_STRIDE; e++) buf[e] = HUGE_VAL;
    buf[0] = k[0]; buf[1] = k[n - 1];
    for (int m = 1; m <= G; m++) buf[1 + m] = k[S * m];
    HIPCHK(hipMemcpy(h->tables + LDS_AUX + AUX_STRIDE * slot, buf, sizeof(buf), hipMemcpyHostToDevice));
    return 0;
}
int32_t fb_set_table(fb_handle h, int32_t kind, const void* data, const int64_t* dims, int32_t ndims) {
    if (h) fsal_invalidate(h);
    if (!h || !data || !dims) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    // fb_step is asynchronous on the handle's non-blocking stream: the uploads below (null-stream copies, a re-allocated gains
    // blob) must not overtake stepping kernels that are still queued or running with the old tables
    HIPCHK(hipStreamSynchronize(h->stream));
    int64_t count = 1;
    for (int k = 0; k < ndims; k++) count *= dims[k];
    if ((kind == FB_TABLE_ROBOT2D) != (h->model == FB_MODEL_ROBOT2D)) return fail("table kind does not belong to this model");
    if (kind == FB_TABLE_ROBOT2D) {
        if (count != FB_R2_TABLE_SIZE) return fail("Robot2D blob must hold FB_R2_TABLE_SIZE doubles");
        std::memcpy(h->r2->table, data, sizeof(double) * FB_R2_TABLE_SIZE);
        h->r2->have_table = true;
        return 0;
    }
    if (kind == FB_TABLE_SCENARIO) return scn_load(h, (const double*)data, ndims == 1 ? dims[0] : -1);
    if (kind == FB_TABLE_CTL_GAINS) {
        if (!is_x2(h)) return fail("table kind does not belong to this model");
        const double* b = (const double*)data;
        static const int rec[10] = {FB_CTL_LQR8_REC, FB_CTL_LQR8_REC, FB_CTL_LQR9_REC, FB_CTL_PID_REC, FB_CTL_PID_REC, FB_CTL_PID_REC,
                                    FB_CTL_LQR8_REC, FB_CTL_LQR8_REC, FB_CTL_PID_REC, FB_CTL_PID_REC};
        int64_t off = 0;
        for (int k = 0; k < 10; k++) {
            if (off + FB_CTL_GRID_HDR > count) return fail("control-law gains blob is truncated");
            const int64_t nE = (int64_t)b[off], nH = (int64_t)b[off + 1];
            if (nE < 1 || nH < 1 || nE > 64 || nH > 64) return fail("control-law gains blob: implausible grid size in lookup %d", k);
            h->gains_off[k] = off;
            off += FB_CTL_GRID_HDR + nE * nH * rec[k];
        }
        if (off != count) return fail("control-law gains blob: %lld doubles given, layout needs %lld", (long long)count, (long long)off);
        if (count > CTL_GAINS_MAX) return fail("control-law gains blob: %lld doubles exceed the %d the periodic kernel stages in LDS", (long long)count, (int)CTL_GAINS_MAX);
        h->gains_total = count;
        h->gains_same_grid = true;
        for (int k = 1; k < 10; k++)
            if (std::memcmp(b + h->gains_off[k], b + h->gains_off[0], sizeof(double) * FB_CTL_GRID_HDR) != 0) h->gains_same_grid = false;
        if (const char* e = getenv("FLIGHTBATCH_CTL_SAME_GRID")) if (e[0] == '0') h->gains_same_grid = false;   // diagnostic: force the per-lookup headers
        if (h->gains) { (void)hipFree(h->gains); h->gains = nullptr; }
        HIPCHK(hipMalloc(&h->gains, sizeof(double) * count));
        HIPCHK(hipMemcpy(h->gains, data, sizeof(double) * count, hipMemcpyHostToDevice));
        h->have_gains = true;
        return 0;
    }
    switch (kind) {
        case FB_TABLE_EGM96:
            if (ndims != 2 || dims[0] != 721 || dims[1] != 1441) return fail("EGM96 table must be float32 [721 x 1441]");
            HIPCHK(hipMemcpy(h->egm96, data, sizeof(float) * 721 * 1441, hipMemcpyHostToDevice));
            break;
        case FB_TABLE_PROPELLER: {
            if (ndims != 3 || dims[0] != PR_NJ || dims[1] != PR_NM || dims[2] != PR_NC) return fail("propeller table must be [21 x 21 x 6]");
            const double* src = (const double*)data;  // column-major (J, Mt, c)
            std::vector<double> il(PR_SIZE);
            for (int c = 0; c < PR_NC; c++)
                for (int j = 0; j < PR_NM; j++)
                    for (int i = 0; i < PR_NJ; i++) il[(i + PR_NJ * j) * PR_NC + c] = src[i + PR_NJ * (j + PR_NM * c)];
            HIPCHK(hipMemcpy(h->tables + LDS_PROP, il.data(), sizeof(double) * PR_SIZE, hipMemcpyHostToDevice));
            break;
        }
        case FB_TABLE_PISTON:
            if (count != PT_SIZE) return fail("piston blob must hold PT_SIZE doubles (csrc/tables.h)");
            HIPCHK(hipMemcpy(h->tables + LDS_PISTON, data, sizeof(double) * PT_SIZE, hipMemcpyHostToDevice));
            if (int32_t rc = put_coarse_knots(h, AUX_N13, (const double*)data + PT_PISTD_N_K, 13)) return rc;
            if (int32_t rc = put_coarse_knots(h, AUX_F11, (const double*)data + PT_F_K, 11)) return rc;
            break;
        case FB_TABLE_AERO:
            if (count != AT_SIZE) return fail("aero blob must hold AT_SIZE doubles (csrc/tables.h)");
            HIPCHK(hipMemcpy(h->tables + LDS_AERO, data, sizeof(double) * AT_SIZE, hipMemcpyHostToDevice));
            if (int32_t rc = put_coarse_knots(h, AUX_GE, (const double*)data + AT_GE_K, 13)) return rc;
            if (int32_t rc = put_coarse_knots(h, AUX_AL26, (const double*)data + AT_CD_ALPHA_K, 26)) return rc;
            if (int32_t rc = put_coarse_knots(h, AUX_AL17, (const double*)data + AT_CL_ALPHA_K, 17)) return rc;
            break;
        default: return fail("unknown table kind");
    }
    h->have_table[kind] = true;
    h->tables_f32_stale = true;
    return 0;
}
// TunableSeaLevelU holds T and p as Ranged values: an assignment saturates to [T_std - 50, T_std + 50] K and [p_std - 10000, p_std + 10000] Pa
// (FP/atmosphere.jl:69-77) — so does every sea-level value that enters through this ABI (fb_get_params / fb_get_env return what is in force)
static double sat_T_sl(double T) { return fmin(fmax(T, isa::T_std - 50.0), isa::T_std + 50.0); }
static double sat_p_sl(double p) { return fmin(fmax(p, isa::p_std - 10000.0), isa::p_std + 10000.0); }
int32_t fb_set_params(fb_handle h, const fb_params* p) {
    if (h) fsal_invalidate(h);
    if (!h || !p) return fail("null argument");
    if (!(p->dt > 0)) return fail("dt must be positive");
    if (!std::isfinite(p->T_sl) || !std::isfinite(p->p_sl) || !std::isfinite(p->wind_ned[0]) || !std::isfinite(p->wind_ned[1]) || !std::isfinite(p->wind_ned[2]) ||
        !std::isfinite(p->h_terrain)) return fail("fb_set_params: T_sl, p_sl, wind_ned and h_terrain must be finite");
    h->params = *p;
    h->params.T_sl = sat_T_sl(p->T_sl); h->params.p_sl = sat_p_sl(p->p_sl);
    return 0;
}
int32_t fb_get_params(fb_handle h, fb_params* p) {
    if (!h || !p) return fail("null argument");
    *p = h->params;
    return 0;
}
// Per-aircraft environment rows (include/flightbatch.h, FB_ENV_*). The two derived rows are filled here the way make_args fills
// Env::ln_p_sl / Env::k_rt for the batch-wide block (same expressions, the host's libm).
int32_t fb_set_env(fb_handle h, const double* env) {
    if (!h) return fail("null handle");
    if (h->model == FB_MODEL_ROBOT2D) return fail("Robot2D has no environment");
    fsal_invalidate(h);
    HIPCHK(hipSetDevice(h->device));
    if (!env) {
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(h->env_rows);
        h->env_rows = nullptr;
        return 0;
    }
    const int64_t n = h->n;
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < FB_NENV; k++)
            if (!std::isfinite(env[(int64_t)k * n + i])) return fail("fb_set_env: aircraft %lld has a non-finite value in row %d (%g)", (long long)i, k, env[(int64_t)k * n + i]);
    std::vector<double> rows((size_t)fbd::ENV_DEV_ROWS * n);
    std::memcpy(rows.data(), env, sizeof(double) * FB_NENV * n);
    for (int64_t i = 0; i < n; i++) {
        // (the reference's Ranged sea-level inputs saturate; a row outside their range is brought to the bound, as `atmosphere.sl.u.T = ...` would)
        const double T = sat_T_sl(env[(int64_t)FB_ENV_T_SL * n + i]), p = sat_p_sl(env[(int64_t)FB_ENV_P_SL * n + i]);
        rows[(size_t)FB_ENV_T_SL * n + i] = T; rows[(size_t)FB_ENV_P_SL * n + i] = p;
        const double lnp = log(p / 101325.0);
        rows[(size_t)fbd::ENV_DEV_LN_P * n + i] = lnp;
        rows[(size_t)fbd::ENV_DEV_K_RT * n + i] = exp(0.5 * 6.5e-3 * 287.05287 / 9.80665 * lnp) / sqrt(T);
    }
    if (!h->env_rows && h->dtype == FB_F32) {   // (say so once — the fp32 stepper reads the batch-wide block only)
        static bool said = false;
        if (!said) { said = true; std::fprintf(stderr, "flightbatch: per-aircraft environment rows on an FB_F32 handle: the fp32 stepper has no instance that reads them, "
                                                       "this handle is stepped by the fp64 one-wave kernel k_step_air<WA, false, *, true> from now on\n"); }
    }
    if (!h->env_rows) HIPCHK(hipMalloc(&h->env_rows, sizeof(double) * fbd::ENV_DEV_ROWS * n));
    HIPCHK(hipMemcpyAsync(h->env_rows, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_has_env(fb_handle h) {
    if (!h) return fail("null handle");
    return h->env_rows ? 1 : 0;
}
int32_t fb_get_env(fb_handle h, double* env) {
    if (!h || !env) return fail("null argument");
    if (!h->env_rows) return fail("no per-aircraft environment rows are set (fb_set_env): the batch-wide block is fb_get_params'");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(env, h->env_rows, sizeof(double) * FB_NENV * h->n, hipMemcpyDeviceToHost));
    return 0;
}

static int32_t set_state_impl(fb_handle h, const double* x, const int32_t* s, bool init) {
    if (h) fsal_invalidate(h);
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->model == FB_MODEL_ROBOT2D) {
        if (x) { if (int32_t rc = r2_upload(h, h->r2, h->r2->r, x, FB_R2_NX)) return rc; }
        if (init) HIPCHK(clear_terminations(h));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (init) { h->r2->steps_done = 0; h->t = 0.0; }
        return 0;
    }
    if (x) { if (int32_t rc = copy_rows(h, h->x, x, nullptr, nx_of(h), row_map_of(h))) return rc; }
    if (s) HIPCHK(hipMemcpyAsync(h->s, s, sizeof(int32_t) * FB_NS * h->n, hipMemcpyHostToDevice, h->stream));
    if (init) HIPCHK(clear_terminations(h));  // init! clears terminations (sim.jl:390-414)
    HIPCHK(hipStreamSynchronize(h->stream));
    if (init) { h->t = 0.0; h->steps_done = 0; }
    return 0;
}
int32_t fb_set_state(fb_handle h, const double* x, const int32_t* s) { return set_state_impl(h, x, s, true); }
int32_t fb_assign_state(fb_handle h, const double* x, const int32_t* s) { return set_state_impl(h, x, s, false); }
int32_t fb_get_state(fb_handle h, double* x, int32_t* s) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->model == FB_MODEL_ROBOT2D) return x ? r2_download(h, h->r2, x, h->r2->r, FB_R2_NX) : 0;
    if (x) { if (int32_t rc = copy_rows(h, h->x, nullptr, x, nx_of(h), row_map_of(h))) return rc; }
    if (s) HIPCHK(hipMemcpyAsync(s, h->s, sizeof(int32_t) * FB_NS * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_set_inputs(fb_handle h, const double* u, const int32_t* ui) {
    if (h) fsal_invalidate(h);
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->model == FB_MODEL_ROBOT2D) return u ? r2_upload(h, h->r2, h->r2->u, u, FB_R2_NU) : 0;
    if (u) HIPCHK(hipMemcpyAsync(h->u, u, sizeof(double) * FB_NU * h->n, hipMemcpyHostToDevice, h->stream));
    if (ui) HIPCHK(hipMemcpyAsync(h->ui, ui, sizeof(int32_t) * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_inputs(fb_handle h, double* u, int32_t* ui) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->model == FB_MODEL_ROBOT2D) return u ? r2_download(h, h->r2, u, h->r2->u, FB_R2_NU) : 0;
    if (u) HIPCHK(hipMemcpyAsync(u, h->u, sizeof(double) * FB_NU * h->n, hipMemcpyDeviceToHost, h->stream));
    if (ui) HIPCHK(hipMemcpyAsync(ui, h->ui, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int32_t fb_f_init(fb_handle h, const double* init, int32_t ninit) {
    if (!h || (!init && ninit != 0)) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->model == FB_MODEL_ROBOT2D) return r2_f_init(h, init, ninit);
    if (is_x2(h) && ninit == 0) {   // f_init!(avionics, vehicle) on the state the host has set
        if (int32_t rc = check_ready_x2(h)) return rc;
        fsal_invalidate(h);
        FB_LAUNCH_X2K(k_x2_init, h->n, make_args(h), ctl_args(h, 0));
        HIPCHK(hipGetLastError());
        h->steps_done = 0;
        h->t = 0.0;
        return 0;
    }
    return fail("fb_f_init: Cessna172Sv0 initialises through fb_trim (TrimParameters) or fb_set_state");
}
int32_t fb_trim(fb_handle h, const double* trim_params, double* trim_state, int32_t* success, double* cost) {
    if (h) fsal_invalidate(h);
    if (h && h->model == FB_MODEL_ROBOT2D) return fail("fb_trim: Robot2D has no trim (use fb_f_init)");
    if (int32_t rc = check_ready_x2(h)) return rc;
    if (!trim_params || !trim_state) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    const int64_t n = h->n;
    if (!h->trim_buf) {
        HIPCHK(hipMalloc(&h->trim_buf, sizeof(double) * ((FB_NTP + FB_NTS + 1) * n + 1)));   // (+ k_trim's queue position)
        HIPCHK(hipMalloc(&h->trim_ok, sizeof(int32_t) * n));
    }
    double* d_tp = h->trim_buf;
    double* d_ts = d_tp + (int64_t)FB_NTP * n;
    double* d_cost = d_ts + (int64_t)FB_NTS * n;
    HIPCHK(hipMemcpyAsync(d_tp, trim_params, sizeof(double) * FB_NTP * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_ts, trim_state, sizeof(double) * FB_NTS * n, hipMemcpyHostToDevice, h->stream));
    // a persistent kernel: one wave per SIMD (k_trim holds the whole register file), aircraft taken from a queue
    unsigned long long* d_next = reinterpret_cast<unsigned long long*>(d_cost + n);
    HIPCHK(hipMemsetAsync(d_next, 0, sizeof(unsigned long long), h->stream));
    int n_cu = 0;
    HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    const int64_t trim_waves = std::min<int64_t>((n + 63) / 64, (int64_t)n_cu * 4);
    if (!h->trim_ws) HIPCHK(hipMalloc(&h->trim_ws, sizeof(double) * fbd::TRIM_WS_ROWS * 64 * trim_waves));
    if (h->env_rows) hipLaunchKernelGGL(k_trim<true>, dim3((unsigned)trim_waves), dim3(64), 0, h->stream, make_args(h), (const double*)d_tp, d_ts, h->trim_ok, d_cost, d_next, h->trim_ws);
    else hipLaunchKernelGGL(k_trim<false>, dim3((unsigned)trim_waves), dim3(64), 0, h->stream, make_args(h), (const double*)d_tp, d_ts, h->trim_ok, d_cost, d_next, h->trim_ws);
    HIPCHK(hipGetLastError());
    if (h->kin == FB_KIN_ECEF) hipLaunchKernelGGL(k_kin_convert<FB_KIN_ECEF>, grid_for(n, 256), dim3(256), 0, h->stream, make_args(h), (const double*)d_tp);
    if (h->kin == FB_KIN_NED) hipLaunchKernelGGL(k_kin_convert<FB_KIN_NED>, grid_for(n, 256), dim3(256), 0, h->stream, make_args(h), (const double*)d_tp);
    if (is_x2(h)) {  // f_init!(aircraft, trim): actuator states, then f_init!(avionics, vehicle) (aircraftbase.jl:255-265)
        FB_LAUNCH_X2K(k_x2_init, n, make_args(h), ctl_args(h, 0));
        HIPCHK(hipGetLastError());
    }
    h->steps_done = 0;
    HIPCHK(clear_terminations(h));
    HIPCHK(hipMemcpyAsync(trim_state, d_ts, sizeof(double) * FB_NTS * n, hipMemcpyDeviceToHost, h->stream));
    if (success) HIPCHK(hipMemcpyAsync(success, h->trim_ok, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->stream));
    if (cost) HIPCHK(hipMemcpyAsync(cost, d_cost, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->t = 0.0;
    return 0;
}

int32_t fb_f_ode(fb_handle h, double* xdot) {
    if (h && h->model == FB_MODEL_ROBOT2D) { HIPCHK(hipSetDevice(h->device)); return r2_f_ode(h, xdot); }
    if (int32_t rc = check_ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    const int64_t n = h->n;
    if (!h->y) HIPCHK(hipMalloc(&h->y, sizeof(double) * FB_NY * n));
    if (xdot && !h->xdot) HIPCHK(hipMalloc(&h->xdot, sizeof(double) * (is_x2(h) ? (int)FB_X2_NX : (int)FB_NX) * n));
    FB_LAUNCH_MK(k_f_ode, grid_for(n, 256), dim3(256), make_args(h), xdot ? h->xdot : (double*)nullptr, h->y);
    HIPCHK(hipGetLastError());
    if (xdot) {
        if (int32_t rc = copy_rows(h, h->xdot, nullptr, xdot, nx_of(h), row_map_of(h))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}
int32_t fb_f_step(fb_handle h) {
    if (h) fsal_invalidate(h);
    if (h && h->model == FB_MODEL_ROBOT2D) {
        if (int32_t rc = r2_ready(h)) return rc;
        HIPCHK(hipSetDevice(h->device));
        R2State* R = h->r2;
        R2_DISPATCH(k_r2_f_step, (long long)R->steps_done);
        return 0;
    }
    if (int32_t rc = check_ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    FB_LAUNCH_MK(k_f_step, grid_for(h->n, 256), dim3(256), make_args(h));
    HIPCHK(hipGetLastError());
    return 0;
}
int32_t fb_f_periodic(fb_handle h) {
    if (!h) return fail("null handle");
    if (h->model == FB_MODEL_ROBOT2D) {
        if (int32_t rc = r2_ready(h)) return rc;
        HIPCHK(hipSetDevice(h->device));
        R2State* R = h->r2;
        R2_DISPATCH(k_r2_f_periodic);
        return 0;
    }
    if (is_x2(h)) {  // f_periodic!(Unconditional(), world): the control laws on the outputs of an f_ode! at the current x
        if (int32_t rc = check_ready_x2(h)) return rc;
        HIPCHK(hipSetDevice(h->device));
        FB_LAUNCH_X2K(k_x2_ctl, h->n, make_args(h), ctl_args(h, 0));
        HIPCHK(hipGetLastError());
        return 0;
    }
    return 0;  // Cessna172Sv0: NoAvionics and @no_periodic systems — nothing to do (c172.jl:695; aircraftbase.jl:131)
}
/* avionics.ctl.u / avionics.ctl.{s,y} of Cessna172Xv2 */
int32_t fb_set_ctl_inputs(fb_handle h, const double* cu) {
    if (!h || !cu) return fail("null argument");
    if (!is_x2(h)) return fail("fb_set_ctl_inputs: only Cessna172Xv2 has control laws");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->cu, cu, sizeof(double) * FB_NCU * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_ctl_inputs(fb_handle h, double* cu) {
    if (!h || !cu) return fail("null argument");
    if (!is_x2(h)) return fail("fb_get_ctl_inputs: only Cessna172Xv2 has control laws");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(cu, h->cu, sizeof(double) * FB_NCU * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_set_ctl_state(fb_handle h, const double* cs) {
    if (!h || !cs) return fail("null argument");
    if (!is_x2(h)) return fail("fb_set_ctl_state: only Cessna172Xv2 has control laws");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->cs, cs, sizeof(double) * FB_NCS * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_ctl_state(fb_handle h, double* cs) {
    if (!h || !cs) return fail("null argument");
    if (!is_x2(h)) return fail("fb_get_ctl_state: only Cessna172Xv2 has control laws");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(cs, h->cs, sizeof(double) * FB_NCS * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_outputs(fb_handle h, double* y) {
    if (!h || !y) return fail("null argument");
    if (h->model == FB_MODEL_ROBOT2D) { HIPCHK(hipSetDevice(h->device)); return r2_download(h, h->r2, y, h->r2->y, FB_R2_NY); }
    if (!h->y) return fail("no outputs yet: call fb_f_ode first");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(y, h->y, sizeof(double) * FB_NY * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_output_fields(fb_handle h, uint32_t field_mask, double* y) {
    if (!h || !y) return fail("null argument");
    if (h->model == FB_MODEL_ROBOT2D) return fail("fb_get_output_fields: Robot2D's output record has no blocks (use fb_get_outputs)");
    if (!h->y) return fail("no outputs yet: call fb_f_ode first");
    if (field_mask == 0 || (field_mask & ~(uint32_t)FB_YF_ALL)) return fail("fb_get_output_fields: unknown bits in field_mask 0x%x", field_mask);
    static const int first[8] = {FB_Y_KIN, FB_Y_AIR, FB_Y_AERO, FB_Y_LDG, FB_Y_PWP, FB_Y_FUEL, FB_Y_DYN, FB_NY};
    HIPCHK(hipSetDevice(h->device));
    double* dst = y;
    for (int b = 0; b < 7; b++) {
        if (!(field_mask & (1u << b))) continue;
        const size_t rows = (size_t)(first[b + 1] - first[b]);   // a block's rows are contiguous in the [FB_NY][n] device record
        HIPCHK(hipMemcpyAsync(dst, h->y + (size_t)first[b] * h->n, sizeof(double) * rows * h->n, hipMemcpyDeviceToHost, h->stream));
        dst += rows * h->n;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int32_t fb_set_steps_per_launch(fb_handle h, int32_t k) {
    if (!h) return fail("null handle");
    if (k < 1) return fail("steps per launch must be >= 1");
    h->steps_per_launch = k;
    return 0;
}
// nsteps of the stepping kernel, no logging
static int32_t step_raw(fb_handle h, int64_t nsteps) {
    if (h->model == FB_MODEL_ROBOT2D) return r2_step(h, nsteps);
    if (h->dtype == FB_F32 && h->tables_f32_stale) {   // fp32 mirror of the table blob for the fp32 stepper
        HIPCHK(hipStreamSynchronize(h->stream));       // kernels already queued still read the old mirror
        std::vector<double> d(TABLE_BUF_DOUBLES);
        HIPCHK(hipMemcpy(d.data(), h->tables, sizeof(double) * TABLE_BUF_DOUBLES, hipMemcpyDeviceToHost));
        std::vector<float> f(d.begin(), d.end());
        if (!h->tables_f32) HIPCHK(hipMalloc(&h->tables_f32, sizeof(float) * TABLE_BUF_DOUBLES));
        HIPCHK(hipMemcpy(h->tables_f32, f.data(), sizeof(float) * TABLE_BUF_DOUBLES, hipMemcpyHostToDevice));
        h->tables_f32_stale = false;
    }
    KArgs a = make_args(h);
    int64_t left = nsteps;
    // Cessna172Xv2: the control laws run inside the stepping kernels every Δt/dt steps (cb_periodic after cb_step, FC/sim.jl:204-218,
    // 366-381), so a launch spans steps_per_launch steps like everyone else's; the kernels get the phase of the periodic update
    while (left > 0) {
        const int k = (int)(left < h->steps_per_launch ? left : h->steps_per_launch);
        a.ctl_phase = a.ctl_ratio > 0 ? (int)(h->steps_done % a.ctl_ratio) : 0;
        a.step0 = h->steps_done;
        const bool stamp = h->timing && h->lev_used < h->lev_max;
        if (stamp) HIPCHK(hipEventRecord(h->lev[2 * h->lev_used], h->stream));
        FB_LAUNCH_STEP(grid_for(h->n, 256), a, k);
        if (stamp) { HIPCHK(hipEventRecord(h->lev[2 * h->lev_used + 1], h->stream)); h->lev_used++; }
        left -= k;
        h->steps_done += k;
        h->launches++;
    }
    HIPCHK(hipGetLastError());
    h->t += (double)nsteps * h->params.dt;
    return 0;
}
int32_t fb_step(fb_handle h, int64_t nsteps) {
    if (h && h->model == FB_MODEL_ROBOT2D) {
        if (int32_t rc = r2_ready(h)) return rc;
    } else if (int32_t rc = check_ready_x2(h)) return rc;
    if (nsteps < 0) return fail("nsteps must be >= 0");
    HIPCHK(hipSetDevice(h->device));
    LogState* L = h->log;
    const bool logging = L && L->every > 0, scripted = h->scn_every > 0;
    if (!logging && !scripted) return step_raw(h, nsteps);
    // launches are cut at the scenario's evaluation instants and at the save instants; behind a step the callbacks run in the reference's order:
    // the user callback (here: the scenario table, evaluated on the device), then the save (cb_save comes last, FC/sim.jl:204-218)
    int64_t left = nsteps;
    while (left > 0) {
        int64_t k = left;
        if (logging) { const int64_t to_save = L->every - (L->step_index % L->every); if (to_save < k) k = to_save; }
        if (scripted) { const int64_t to_scn = h->scn_every - (h->steps_done % h->scn_every); if (to_scn < k) k = to_scn; }
        const bool saves = logging && (L->step_index + k) % L->every == 0;
        if (saves && L->count >= L->capacity) return fail("log capacity (%lld samples) exhausted", (long long)L->capacity);
        if (int32_t rc = step_raw(h, k)) return rc;
        if (logging) L->step_index += k;
        left -= k;
        if (scripted && h->steps_done % h->scn_every == 0)
            if (int32_t rc = scn_evaluate(h)) return rc;
        if (saves)
            if (int32_t rc = log_record(h)) return rc;
    }
    return 0;
}
/* ---- on-device TimeSeries log ---- */
int32_t fb_log_configure(fb_handle h, int64_t every, int64_t capacity, const int32_t* rows, int32_t nrows) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    log_free(h);
    if (every <= 0) return 0;  // logging off
    if (capacity <= 0 || nrows <= 0 || !rows) return fail("fb_log_configure: capacity and the row list must be non-empty");
    const int ny = h->model == FB_MODEL_ROBOT2D ? FB_R2_NY : FB_NY, nx = h->model == FB_MODEL_ROBOT2D ? (int)FB_R2_NX : nx_of(h);
    LogState* L = new LogState();
    std::vector<int32_t> dev_rows(rows, rows + nrows);   // state rows are given in the C ABI's order; the gather works on device rows
    const row_map_t map = h->model == FB_MODEL_ROBOT2D ? nullptr : row_map_of(h);
    for (int j = 0; j < nrows; j++) {
        const int r = rows[j];
        const bool ok = (r >= 0 && r < ny) || (r >= FB_LOG_X0 && r < FB_LOG_X0 + nx);
        if (!ok) { delete L; return fail("fb_log_configure: row %d is neither an output row [0,%d) nor FB_LOG_X0 + state row [0,%d)", r, ny, nx); }
        if (r < FB_LOG_X0) L->need_y = true;
        else if (map) dev_rows[j] = FB_LOG_X0 + map(r - FB_LOG_X0);
    }
    h->log = L;
    L->every = every; L->capacity = capacity; L->nrows = nrows;
    HIPCHK(hipMalloc(&L->rows_dev, sizeof(int32_t) * nrows));
    HIPCHK(hipMemcpy(L->rows_dev, dev_rows.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice));
    const size_t bytes = (size_t)capacity * nrows * h->n * log_esize(h);
    if (hipMalloc(&L->data, bytes) != hipSuccess) {
        (void)hipGetLastError();
        log_free(h);
        return fail("fb_log_configure: cannot allocate %zu bytes of device memory for the log", bytes);
    }
    L->t.reserve((size_t)capacity);
    return 0;
}
int32_t fb_log_clear(fb_handle h) {
    if (!h) return fail("null handle");
    if (h->log) { h->log->count = 0; h->log->step_index = 0; h->log->t.clear(); }
    return 0;
}
int32_t fb_log_record(fb_handle h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    return log_record(h);
}
int32_t fb_log_count(fb_handle h, int64_t* count) {
    if (!h || !count) return fail("null argument");
    *count = h->log ? h->log->count : 0;
    return 0;
}
int32_t fb_log_read(fb_handle h, int64_t first, int64_t count, double* t, double* data) {
    if (!h) return fail("null handle");
    LogState* L = h->log;
    if (!L) return fail("the log is not configured (fb_log_configure)");
    if (first < 0 || count < 0 || first + count > L->count) return fail("fb_log_read: samples [%lld, %lld) requested, %lld recorded", (long long)first, (long long)(first + count), (long long)L->count);
    HIPCHK(hipSetDevice(h->device));
    if (t) for (int64_t k = 0; k < count; k++) t[k] = L->t[(size_t)(first + k)];
    if (data && count > 0) {
        const size_t e = log_esize(h), per = (size_t)L->nrows * h->n;
        const char* src = (const char*)L->data + (size_t)first * per * e;
        if (e == 8) {
            HIPCHK(hipMemcpyAsync(data, src, (size_t)count * per * 8, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        } else {
            std::vector<float> f((size_t)count * per);
            HIPCHK(hipMemcpyAsync(f.data(), src, f.size() * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (size_t k = 0; k < f.size(); k++) data[k] = (double)f[k];
        }
    }
    return 0;
}
/* checkpoint / restore support: the step counter (phase of the periodic update) and the sticky status words */
int32_t fb_get_step_count(fb_handle h, int64_t* count) {
    if (!h || !count) return fail("null argument");
    *count = h->model == FB_MODEL_ROBOT2D ? (int64_t)h->r2->steps_done : h->steps_done;
    return 0;
}
int32_t fb_set_step_count(fb_handle h, int64_t count, double t) {
    if (h) fsal_invalidate(h);
    if (!h) return fail("null handle");
    if (count < 0) return fail("step count must be >= 0");
    if (h->model == FB_MODEL_ROBOT2D) h->r2->steps_done = count;
    h->steps_done = count;
    if (h->log) h->log->step_index = count;
    h->t = t;
    return 0;
}
int32_t fb_set_status(fb_handle h, const int32_t* status) {
    if (!h || !status) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->status, status, sizeof(int32_t) * h->n, hipMemcpyHostToDevice, h->stream));
    const long long step0 = h->model == FB_MODEL_ROBOT2D ? (long long)h->r2->steps_done : (long long)h->steps_done;
    hipLaunchKernelGGL(k_mark_host_status, grid_for(h->n, 256), dim3(256), 0, h->stream, (const int32_t*)h->status, h->term_step, h->term_where, step0, h->n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_set_termination(fb_handle h, const int64_t* step, const int32_t* where) {
    if (!h || !step || !where) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    std::vector<int32_t> st((size_t)h->n), wh((size_t)h->n);
    std::vector<long long> ts((size_t)h->n);
    HIPCHK(hipMemcpyAsync(st.data(), h->status, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->n; i++) {
        const bool term = (st[(size_t)i] & ~FB_ST_NAN) != 0;
        if (term && (where[i] < FB_TERM_OUTSIDE_STEP || where[i] > FB_TERM_F_ODE_REEVAL || step[i] < 0))
            return fail("fb_set_termination: aircraft %lld carries a termination bit and the record (%lld, %d) is not one fb_get_termination returns", (long long)i, (long long)step[i], (int)where[i]);
        ts[(size_t)i] = term ? (long long)step[i] : 0;
        wh[(size_t)i] = term ? where[i] : (int32_t)FB_TERM_NONE;
    }
    HIPCHK(hipMemcpyAsync(h->term_step, ts.data(), sizeof(long long) * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->term_where, wh.data(), sizeof(int32_t) * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
#if defined(FB_STAMP) || defined(FB_TRIM_STAMP)
// diagnostic builds only (tools/stamp_profile.py): the per-phase cycle accumulators of c172_device_impl.inc
int32_t fb_debug_stamps(unsigned long long* acc, unsigned long long* cnt, int32_t reset) {
    if (acc) HIPCHK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(fbd::g_stamp_acc), sizeof(unsigned long long) * 32));
    if (cnt) HIPCHK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(fbd::g_stamp_cnt), sizeof(unsigned long long) * 32));
    if (reset) {
        unsigned long long z[32] = {0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(fbd::g_stamp_acc), z, sizeof z));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(fbd::g_stamp_cnt), z, sizeof z));
    }
    return 0;
}
#endif
#ifdef FB_DUO_SYNC_DEBUG
int32_t fb_debug_duo_sync(unsigned* out40) { HIPCHK(hipMemcpyFromSymbol(out40, HIP_SYMBOL(fbd::g_duo_sync_dbg), sizeof(unsigned) * 40)); return 0; }
#endif
int32_t fb_sync(fb_handle h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
double fb_time(fb_handle h) { return h ? h->t : 0.0; }

int32_t fb_status(fb_handle h, int32_t* status) {
    if (!h || !status) return fail("null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(status, h->status, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_get_termination(fb_handle h, int64_t* step, int32_t* where) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    std::vector<int32_t> st((size_t)h->n);
    std::vector<long long> ts(step ? (size_t)h->n : 0);
    HIPCHK(hipMemcpyAsync(st.data(), h->status, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (step) HIPCHK(hipMemcpyAsync(ts.data(), h->term_step, sizeof(long long) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (where) HIPCHK(hipMemcpyAsync(where, h->term_where, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // the record is written when an aircraft terminates and means something only next to a termination bit (whatever clears the status
    // words clears the record with them: clear_terminations)
    for (int64_t i = 0; i < h->n; i++) {
        const bool term = (st[(size_t)i] & ~FB_ST_NAN) != 0;
        if (step) step[i] = term ? (int64_t)ts[(size_t)i] : -1;
        if (where && !term) where[i] = FB_TERM_NONE;
    }
    return 0;
}

int32_t fb_scenario_configure(fb_handle h, int32_t every) {
    if (!h) return fail("null handle");
    if (every < 0) return fail("fb_scenario_configure: the evaluation period must be >= 1 step (0: scenario off)");
    HIPCHK(hipSetDevice(h->device));
    if (every == 0) { HIPCHK(hipStreamSynchronize(h->stream)); scn_free(h); return 0; }
    if (!h->scn_prog) return fail("fb_scenario_configure: no scenario table is loaded (fb_set_table FB_TABLE_SCENARIO)");
    h->scn_every = every;
    return 0;
}
int32_t fb_scenario_set_params(fb_handle h, const double* par) {
    if (!h || !par) return fail("null argument");
    if (!h->scn_prog) return fail("no scenario table is loaded (fb_set_table FB_TABLE_SCENARIO)");
    HIPCHK(hipSetDevice(h->device));
    if (h->scn_npar > 0) HIPCHK(hipMemcpyAsync(h->scn_par, par, sizeof(double) * h->scn_npar * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_scenario_get_params(fb_handle h, double* par) {
    if (!h || !par) return fail("null argument");
    if (!h->scn_prog) return fail("no scenario table is loaded (fb_set_table FB_TABLE_SCENARIO)");
    HIPCHK(hipSetDevice(h->device));
    if (h->scn_npar > 0) HIPCHK(hipMemcpyAsync(par, h->scn_par, sizeof(double) * h->scn_npar * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_scenario_get_state(fb_handle h, int32_t* phase, int64_t* since_step, double* rec) {
    if (!h) return fail("null handle");
    if (!h->scn_prog) return fail("no scenario table is loaded (fb_set_table FB_TABLE_SCENARIO)");
    HIPCHK(hipSetDevice(h->device));
    static_assert(sizeof(long long) == sizeof(int64_t), "");
    if (phase) HIPCHK(hipMemcpyAsync(phase, h->scn_phase, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (since_step) HIPCHK(hipMemcpyAsync(since_step, h->scn_since, sizeof(int64_t) * h->n, hipMemcpyDeviceToHost, h->stream));
    if (rec && h->scn_nrec > 0) HIPCHK(hipMemcpyAsync(rec, h->scn_rec, sizeof(double) * h->scn_nrec * h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t fb_scenario_set_state(fb_handle h, const int32_t* phase, const int64_t* since_step, const double* rec) {
    if (!h) return fail("null handle");
    if (!h->scn_prog) return fail("no scenario table is loaded (fb_set_table FB_TABLE_SCENARIO)");
    if (phase)
        for (int64_t i = 0; i < h->n; i++)
            if (phase[i] < 0 || phase[i] >= h->scn_nph) return fail("fb_scenario_set_state: aircraft %lld in phase %d, the table has %d", (long long)i, (int)phase[i], h->scn_nph);
    HIPCHK(hipSetDevice(h->device));
    if (phase) HIPCHK(hipMemcpyAsync(h->scn_phase, phase, sizeof(int32_t) * h->n, hipMemcpyHostToDevice, h->stream));
    if (since_step) HIPCHK(hipMemcpyAsync(h->scn_since, since_step, sizeof(int64_t) * h->n, hipMemcpyHostToDevice, h->stream));
    if (rec && h->scn_nrec > 0) HIPCHK(hipMemcpyAsync(h->scn_rec, rec, sizeof(double) * h->scn_nrec * h->n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int32_t fb_timing_begin(fb_handle h) {
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->device));
    h->launches = 0;
    h->lev_max = h->lev_used = 0;
    h->timing = true;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    return 0;
}
int32_t fb_timing_begin_per_launch(fb_handle h, int64_t max_launches) {
    if (!h) return fail("null handle");
    if (h->model == FB_MODEL_ROBOT2D) return fail("fb_timing_begin_per_launch: not supported for Robot2D");
    if (max_launches < 0 || max_launches > 65536) return fail("fb_timing_begin_per_launch: max_launches must be in [0, 65536]");
    HIPCHK(hipSetDevice(h->device));
    while ((int64_t)h->lev.size() < 2 * max_launches) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreate(&e));
        h->lev.push_back(e);
    }
    if (int32_t rc = fb_timing_begin(h)) return rc;
    h->lev_max = max_launches;
    return 0;
}
int32_t fb_timing_launches(fb_handle h, float* ms, int64_t cap, int64_t* n) {
    if (!h || !n) return fail("null argument");
    if (h->timing) return fail("fb_timing_launches: call fb_timing_end first");
    HIPCHK(hipSetDevice(h->device));
    *n = h->lev_used;
    for (int64_t k = 0; ms && k < h->lev_used && k < cap; k++) {
        HIPCHK(hipEventSynchronize(h->lev[2 * k + 1]));
        HIPCHK(hipEventElapsedTime(&ms[k], h->lev[2 * k], h->lev[2 * k + 1]));
    }
    return 0;
}
int32_t fb_timing_end(fb_handle h, float* ms, int64_t* n_launches) {
    if (!h) return fail("null handle");
    if (!h->timing) return fail("fb_timing_end without fb_timing_begin");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, h->ev0, h->ev1));
    if (ms) *ms = t;
    if (n_launches) *n_launches = h->launches;
    h->timing = false;
    return 0;
}

}  // extern "C"

#include "fb_comm.inc"

