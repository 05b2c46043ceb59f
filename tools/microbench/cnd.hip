// What does a select cost on gfx950? v_cndmask_b32 measured 17-19 cycles back to back (tools/microbench/issue.hip, pk32.hip) but ~4.4 between other
// VALU instructions: this separates the cases. One wave per SIMD and two; span of all waves of the workgroup.
// Build + run: tools/microbench/run_cnd.sh -> gpurun_out/r05_cnd_microbench.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)
#define REP64(x) REP4(REP16(x))
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","vcc","s20","s21","s22","s23"
#define TIME(idx, body) { __syncthreads(); __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(body ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin[idx], t0); atomicMax(&tmax[idx], t1); } }
// every body below is 4 instructions; REP32 -> 128 instructions
#define B_CND_VCC   "v_cndmask_b32 v10, v6, v7, vcc\n v_cndmask_b32 v11, v8, v9, vcc\n v_cndmask_b32 v12, v6, v7, vcc\n v_cndmask_b32 v13, v8, v9, vcc\n"
#define B_CND_SGPR  "v_cndmask_b32 v10, v6, v7, s[20:21]\n v_cndmask_b32 v11, v8, v9, s[20:21]\n v_cndmask_b32 v12, v6, v7, s[20:21]\n v_cndmask_b32 v13, v8, v9, s[20:21]\n"
#define B_CND_2SG   "v_cndmask_b32 v10, v6, v7, s[20:21]\n v_cndmask_b32 v11, v8, v9, s[22:23]\n v_cndmask_b32 v12, v6, v7, s[20:21]\n v_cndmask_b32 v13, v8, v9, s[22:23]\n"
#define B_PAIR_FMA  "v_cndmask_b32 v10, v6, v7, vcc\n v_cndmask_b32 v11, v8, v9, vcc\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n"
#define B_ALT_FMA   "v_cndmask_b32 v10, v6, v7, vcc\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_cndmask_b32 v11, v8, v9, vcc\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n"
#define B_FMA       "v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n v_fma_f64 v[18:19], v[2:3], v[4:5], v[18:19]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[20:21]\n"
#define B_CMP_CND   "v_cmp_lt_f64 vcc, v[2:3], v[4:5]\n v_cndmask_b32 v10, v6, v7, vcc\n v_cndmask_b32 v11, v8, v9, vcc\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n"
#define B_CMPS_CND  "v_cmp_lt_f64 s[20:21], v[2:3], v[4:5]\n v_cndmask_b32 v10, v6, v7, s[20:21]\n v_cndmask_b32 v11, v8, v9, s[20:21]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n"
#define B_MAX       "v_max_f64 v[10:11], v[2:3], v[4:5]\n v_min_f64 v[12:13], v[2:3], v[4:5]\n v_max_f64 v[14:15], v[2:3], v[4:5]\n v_min_f64 v[16:17], v[2:3], v[4:5]\n"
#define B_MOV       "v_mov_b32 v10, v6\n v_mov_b32 v11, v7\n v_mov_b32 v12, v8\n v_mov_b32 v13, v9\n"
#define B_CND_DEP   "v_cndmask_b32 v10, v10, v7, vcc\n v_cndmask_b32 v11, v11, v9, vcc\n v_cndmask_b32 v10, v10, v7, vcc\n v_cndmask_b32 v11, v11, v9, vcc\n"
#define B_CND_NOP   "v_cndmask_b32 v10, v6, v7, vcc\n s_nop 0\n v_cndmask_b32 v11, v8, v9, vcc\n s_nop 0\n"
#define NB 13
__global__ void k_cnd(unsigned long long* out, int iters) {
    __shared__ unsigned long long tmin[NB], tmax[NB];
    if (threadIdx.x < NB) { tmin[threadIdx.x] = ~0ull; tmax[threadIdx.x] = 0; }
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3fe00000\n v_mov_b32 v6, 0\n v_mov_b32 v7, 1\n v_mov_b32 v8, 2\n v_mov_b32 v9, 3\n"
                 "s_mov_b64 s[20:21], 0x5555\n s_mov_b64 s[22:23], 0x3333\n s_mov_b64 vcc, 0x5555\n" ::: CLOB);
    TIME(0, REP32(B_CND_VCC)) TIME(1, REP32(B_CND_SGPR)) TIME(2, REP32(B_CND_2SG)) TIME(3, REP32(B_PAIR_FMA)) TIME(4, REP32(B_ALT_FMA)) TIME(5, REP32(B_FMA))
    TIME(6, REP32(B_CMP_CND)) TIME(7, REP32(B_CMPS_CND)) TIME(8, REP32(B_MAX)) TIME(9, REP32(B_MOV)) TIME(10, REP32(B_CND_DEP)) TIME(11, REP32(B_CND_NOP))
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < NB) out[threadIdx.x] = tmax[threadIdx.x] - tmin[threadIdx.x];
}
int main() {
    unsigned long long* out; CHK(hipMalloc(&out, NB * 8));
    const char* names[] = {"4 x v_cndmask_b32 (vcc)", "4 x v_cndmask_b32 (one SGPR-pair mask)", "4 x v_cndmask_b32 (two SGPR-pair masks alternating)",
                           "cndmask, cndmask, fma_f64, fma_f64 (an fp64 select between arithmetic)", "cndmask, fma_f64, cndmask, fma_f64", "4 x v_fma_f64",
                           "v_cmp_lt_f64 vcc; cndmask; cndmask; fma_f64 (compare + fp64 select)", "v_cmp_lt_f64 s[20:21]; cndmask; cndmask; fma_f64",
                           "v_max_f64, v_min_f64, v_max_f64, v_min_f64", "4 x v_mov_b32", "4 x v_cndmask_b32 (vcc), each dependent on the one two before", "cndmask, s_nop 0, cndmask, s_nop 0"};
    const int iters = 200;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_cnd, dim3(256), dim3(threads), 0, 0, out, iters); CHK(hipDeviceSynchronize()); }
        unsigned long long r[NB]; CHK(hipMemcpy(r, out, NB * 8, hipMemcpyDeviceToHost));
        const int wps = threads / 256;
        printf("--- %d wave(s) per SIMD: SIMD cycles per GROUP of four instructions (span / groups per wave / waves per SIMD)\n", wps);
        for (int k = 0; k < 12; k++) printf("%-75s %7.2f\n", names[k], (double)r[k] / iters / 32 / wps);
    }
    return 0;
}
