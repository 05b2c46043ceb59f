// Issue cost of instruction mixes for ONE wave per SIMD (256-thread workgroup, one per CU) and TWO (512 threads) on gfx950.
// Each pattern is 128 repetitions of an instruction group, written in inline asm on fixed registers, timed with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
// eight independent fp64 accumulators v[10:25], operands v[2:3], v[4:5]
#define FMA8 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n" \
             "v_fma_f64 v[18:19], v[2:3], v[4:5], v[18:19]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[20:21]\n v_fma_f64 v[22:23], v[2:3], v[4:5], v[22:23]\n v_fma_f64 v[24:25], v[2:3], v[4:5], v[24:25]\n"
#define FMA_S(other) "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n" other "v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n" other "v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n" other \
                     "v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n" other "v_fma_f64 v[18:19], v[2:3], v[4:5], v[18:19]\n" other "v_fma_f64 v[20:21], v[2:3], v[4:5], v[20:21]\n" other \
                     "v_fma_f64 v[22:23], v[2:3], v[4:5], v[22:23]\n" other "v_fma_f64 v[24:25], v[2:3], v[4:5], v[24:25]\n" other
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","s20","s21","s22","s23","a0","a1","vcc"
#define TIME(idx, body) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(body ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); r[idx] = __builtin_amdgcn_s_memtime() - t0; }

__global__ void k_issue(unsigned long long* out, int iters) {
    __shared__ double lds[2048];
    lds[threadIdx.x] = 1.0; lds[threadIdx.x + 512] = 2.0;
    __syncthreads();
    unsigned long long r[16] = {0};
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_lshlrev_b32 v6, 3, %0\n v_mov_b32 v7, 0\n" :: "v"(threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7");
    TIME(0, REP16(FMA8))                                          // 128 independent fma (8 accumulators)
    TIME(1, REP16(FMA_S("s_mov_b32 s20, 0x12345\n")))            // 128 x (fma + s_mov)
    TIME(2, REP16(FMA_S("v_mov_b32 v26, v6\n")))                 // 128 x (fma + v_mov)
    TIME(3, REP16(FMA_S("v_accvgpr_write_b32 a0, v6\n")))        // 128 x (fma + accvgpr write)
    TIME(4, REP16(FMA_S("s_nop 0\n")))                           // 128 x (fma + s_nop)
    TIME(5, REP64("v_mov_b32 v26, v6\n v_mov_b32 v27, v6\n"))    // 128 x v_mov
    TIME(6, REP64("s_mov_b32 s20, 0x123\n s_mov_b32 s21, 0x456\n")) // 128 x s_mov
    TIME(7, REP16(FMA_S("ds_read_b64 v[28:29], v6\n")) "s_waitcnt lgkmcnt(0)\n")   // 128 x (fma + ds_read), one wait at the end
    TIME(8, REP64("v_mul_f64 v[10:11], v[2:3], v[4:5]\n v_mul_f64 v[12:13], v[2:3], v[4:5]\n"))   // 128 x independent v_mul_f64
    TIME(9, REP64("v_add_f64 v[10:11], v[2:3], v[4:5]\n v_add_f64 v[12:13], v[2:3], v[4:5]\n"))   // 128 x v_add_f64
    TIME(10, REP64("v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n"))   // 128 dependent fma
    TIME(11, REP16(FMA_S("v_mov_b32 v26, v6\n s_mov_b32 s20, 0x12345\n")))   // 128 x (fma + v_mov + s_mov)
    TIME(12, REP64("v_cndmask_b32 v26, v6, v7, vcc\n v_cndmask_b32 v27, v6, v7, vcc\n"))       // 128 x v_cndmask
    TIME(13, REP64("v_rcp_f64 v[10:11], v[2:3]\n v_rcp_f64 v[12:13], v[2:3]\n"))               // 128 x v_rcp_f64
    TIME(14, REP64("v_fmac_f64 v[10:11], v[2:3], v[4:5]\n v_fmac_f64 v[12:13], v[2:3], v[4:5]\n"))   // 128 x v_fmac_f64 e32
    TIME(15, REP64("v_max_f64 v[10:11], v[2:3], v[4:5]\n v_cmp_le_f64 vcc, v[2:3], v[4:5]\n")) // 64 x (max + cmp) -> reported per 128 "groups" = half a pair
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < 16; k++) out[k] = r[k];
    if (blockIdx.x == 0 && threadIdx.x == blockDim.x - 1) for (int k = 0; k < 16; k++) out[16 + k] = r[k];
}

int main() {
    unsigned long long* out; CHK(hipMalloc(&out, 32 * 8));
    const char* names[] = {"v_fma_f64 independent", "v_fma_f64 + s_mov_b32", "v_fma_f64 + v_mov_b32", "v_fma_f64 + v_accvgpr_write", "v_fma_f64 + s_nop 0", "v_mov_b32",
                           "s_mov_b32", "v_fma_f64 + ds_read_b64 (1 wait per 128)", "v_mul_f64", "v_add_f64", "v_fma_f64 dependent", "v_fma_f64 + v_mov + s_mov", "v_cndmask_b32",
                           "v_rcp_f64", "v_fmac_f64_e32", "(v_max_f64 + v_cmp_le_f64) / 2"};
    const int iters = 200;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_issue, dim3(256), dim3(threads), 0, 0, out, iters); CHK(hipDeviceSynchronize()); }
        unsigned long long r[32]; CHK(hipMemcpy(r, out, 32 * 8, hipMemcpyDeviceToHost));
        printf("--- %d threads per workgroup (%d wave(s) per SIMD): cycles per group (a group = one line's instruction set)\n", threads, threads / 256);
        for (int k = 0; k < 16; k++) printf("%-44s first wave %7.2f   last wave %7.2f\n", names[k], (double)r[k] / iters / 128, (double)r[16 + k] / iters / 128);
    }
    return 0;
}
