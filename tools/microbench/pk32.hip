// Does packed fp32 (v_pk_fma_f32, two aircraft per lane) buy issue slots on gfx950? Cycles per instruction of independent v_fma_f32,
// v_pk_fma_f32, v_pk_mul_f32 / v_pk_add_f32, v_cndmask_b32 and a 1:1 mix, for ONE wave per SIMD (256 threads, one workgroup per CU)
// and TWO (512 threads) — the fp32 stepper (fbf::k_step_f32) runs two. 128-instruction groups on fixed registers, timed with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 pk32.hip -o pk32      (run by tools/microbench/run_pk32.sh -> profiles/r05_pk32_microbench.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","vcc"
// every wave of the workgroup runs the body; the SPAN from the first wave's start to the last wave's end is what the SIMDs needed for all of them
#define TIME(idx, body) { __syncthreads(); __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(body ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin[idx], t0); atomicMax(&tmax[idx], t1); } }
// eight independent accumulators
#define FMA8 "v_fma_f32 v10, v2, v4, v10\n v_fma_f32 v11, v2, v4, v11\n v_fma_f32 v12, v2, v4, v12\n v_fma_f32 v13, v2, v4, v13\n" \
             "v_fma_f32 v14, v2, v4, v14\n v_fma_f32 v15, v2, v4, v15\n v_fma_f32 v16, v2, v4, v16\n v_fma_f32 v17, v2, v4, v17\n"
#define PKFMA8 "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n v_pk_fma_f32 v[14:15], v[2:3], v[4:5], v[14:15]\n v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n" \
               "v_pk_fma_f32 v[18:19], v[2:3], v[4:5], v[18:19]\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[22:23]\n v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[24:25]\n"
#define PKMULADD8 "v_pk_mul_f32 v[10:11], v[2:3], v[4:5]\n v_pk_add_f32 v[12:13], v[2:3], v[4:5]\n v_pk_mul_f32 v[14:15], v[2:3], v[4:5]\n v_pk_add_f32 v[16:17], v[2:3], v[4:5]\n" \
                  "v_pk_mul_f32 v[18:19], v[2:3], v[4:5]\n v_pk_add_f32 v[20:21], v[2:3], v[4:5]\n v_pk_mul_f32 v[22:23], v[2:3], v[4:5]\n v_pk_add_f32 v[24:25], v[2:3], v[4:5]\n"
#define CND8 "v_cndmask_b32 v10, v6, v7, vcc\n v_cndmask_b32 v11, v6, v7, vcc\n v_cndmask_b32 v12, v6, v7, vcc\n v_cndmask_b32 v13, v6, v7, vcc\n" \
             "v_cndmask_b32 v14, v6, v7, vcc\n v_cndmask_b32 v15, v6, v7, vcc\n v_cndmask_b32 v16, v6, v7, vcc\n v_cndmask_b32 v17, v6, v7, vcc\n"
#define MIX8 "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[10:11]\n v_cndmask_b32 v20, v6, v7, vcc\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n v_cndmask_b32 v21, v6, v7, vcc\n" \
             "v_pk_fma_f32 v[14:15], v[2:3], v[4:5], v[14:15]\n v_cndmask_b32 v22, v6, v7, vcc\n v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n v_cndmask_b32 v23, v6, v7, vcc\n"
#define RCP8 "v_rcp_f32 v10, v2\n v_rcp_f32 v11, v2\n v_rcp_f32 v12, v2\n v_rcp_f32 v13, v2\n v_rcp_f32 v14, v2\n v_rcp_f32 v15, v2\n v_rcp_f32 v16, v2\n v_rcp_f32 v17, v2\n"

__global__ void k_pk(unsigned long long* out, int iters) {
    __shared__ unsigned long long tmin[8], tmax[8];
    if (threadIdx.x < 8) { tmin[threadIdx.x] = ~0ull; tmax[threadIdx.x] = 0; }
    asm volatile("v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 0.5\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 0\n v_mov_b32 v7, 1\n" ::: "v2", "v3", "v4", "v5", "v6", "v7");
    TIME(0, REP16(FMA8))        // 128 x v_fma_f32            (64 fma lanes each)
    TIME(1, REP16(PKFMA8))      // 128 x v_pk_fma_f32         (128 fma lanes each)
    TIME(2, REP16(PKMULADD8))   // 128 x v_pk_mul / v_pk_add
    TIME(3, REP16(CND8))        // 128 x v_cndmask_b32
    TIME(4, REP16(MIX8))        // 64 x (v_pk_fma_f32 + v_cndmask_b32)
    TIME(5, REP16(RCP8))        // 128 x v_rcp_f32
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 8) out[threadIdx.x] = tmax[threadIdx.x] - tmin[threadIdx.x];
}

int main() {
    unsigned long long* out; CHK(hipMalloc(&out, 16 * 8));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32 / v_pk_add_f32", "v_cndmask_b32", "v_pk_fma_f32 + v_cndmask_b32 (per instruction)", "v_rcp_f32"};
    const int iters = 200;
    for (int threads : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_pk, dim3(256), dim3(threads), 0, 0, out, iters); CHK(hipDeviceSynchronize()); }
        unsigned long long r[8]; CHK(hipMemcpy(r, out, 8 * 8, hipMemcpyDeviceToHost));
        const int wps = threads / 256;
        printf("--- %d wave(s) per SIMD (a %d-thread workgroup per CU): SIMD cycles per wave-instruction = span / (instructions per wave x waves per SIMD)\n", wps, threads);
        for (int k = 0; k < 6; k++) {
            const double c = (double)r[k] / iters / 128 / wps;
            const double lanes = (k == 1 || k == 2) ? 128.0 : (k == 4 ? 96.0 : 64.0);
            printf("%-50s %6.2f cycles per instruction  (%5.1f lane-operations per SIMD cycle)\n", names[k], c, lanes / c);
        }
    }
    return 0;
}
