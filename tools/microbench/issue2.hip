// More issue-cost probes for ONE wave per SIMD on gfx950 (see issue.hip): selects, compares, carries, lane moves, LDS, SGPR operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s20","s21","s22","s23","s24","s25","a0","a1","vcc"
#define TIME(idx, n, body) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(body ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); r[idx] = (__builtin_amdgcn_s_memtime() - t0); cnt[idx] = n; }
#define NT 32
__global__ void k_issue(unsigned long long* out, int iters) {
    __shared__ double lds[4096];
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) lds[k] = 1.0;
    __syncthreads();
    unsigned long long r[NT] = {0}; int cnt[NT] = {0};
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_lshlrev_b32 v6, 3, %0\n v_mov_b32 v7, 0\n"
                 "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0x40000000\n s_mov_b64 s[24:25], exec\n s_mov_b64 vcc, exec\n"
                 :: "v"(threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "s20", "s21", "s22", "s23", "s24", "s25", "vcc");
    TIME(0, 128, REP64("v_cndmask_b32 v26, v6, v7, vcc\n v_cndmask_b32 v27, v6, v7, vcc\n"))
    TIME(1, 128, REP64("v_cndmask_b32_e64 v26, v6, v7, s[24:25]\n v_cndmask_b32_e64 v27, v6, v7, s[24:25]\n"))
    TIME(2, 128, REP64("v_cmp_le_f64 vcc, v[2:3], v[4:5]\n v_cndmask_b32 v26, v6, v7, vcc\n"))                 // cmp -> select (2 instructions per group of 2)
    TIME(3, 192, REP64("v_cmp_le_f64 vcc, v[2:3], v[4:5]\n v_cndmask_b32 v26, v6, v7, vcc\n v_cndmask_b32 v27, v6, v7, vcc\n"))   // fp64 select: cmp + 2 cndmask
    TIME(4, 128, REP64("v_cmp_le_f64 vcc, v[2:3], v[4:5]\n v_addc_co_u32 v26, vcc, 0, v7, vcc\n"))               // compare-and-count
    TIME(5, 128, REP64("v_max_f64 v[10:11], v[2:3], v[4:5]\n v_min_f64 v[12:13], v[2:3], v[4:5]\n"))
    TIME(6, 128, REP64("v_readlane_b32 s20, v6, 3\n v_readlane_b32 s21, v6, 4\n"))
    TIME(7, 128, REP64("v_writelane_b32 v26, s22, 3\n v_writelane_b32 v26, s23, 4\n"))
    TIME(8, 128, REP64("v_accvgpr_read_b32 v26, a0\n v_accvgpr_read_b32 v27, a1\n"))
    TIME(9, 128, REP64("ds_read_b64 v[28:29], v6\n ds_read_b64 v[30:31], v6 offset:2048\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(10, 128, REP64("ds_read2_b64 v[28:31], v6 offset1:8\n ds_read2_b64 v[10:13], v6 offset0:16 offset1:24\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(11, 128, REP64("ds_write_b64 v6, v[2:3]\n ds_write_b64 v6, v[4:5] offset:2048\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(12, 128, REP64("ds_read_b32 v28, v6\n ds_read_b32 v30, v6 offset:2048\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(13, 128, REP64("v_mul_f64 v[10:11], s[20:21], v[4:5]\n v_mul_f64 v[12:13], s[22:23], v[4:5]\n"))
    TIME(14, 128, REP64("v_fma_f64 v[10:11], s[20:21], v[4:5], v[10:11]\n v_fma_f64 v[12:13], s[22:23], v[4:5], v[12:13]\n"))
    TIME(15, 128, REP64("v_fma_f64 v[10:11], v[2:3], v[4:5], 0\n v_fma_f64 v[12:13], v[2:3], v[4:5], 0\n"))     // a multiply written as fma(a, b, 0)
    TIME(16, 128, REP64("v_fma_f64 v[10:11], v[2:3], 1.0, v[4:5]\n v_fma_f64 v[12:13], v[2:3], 1.0, v[4:5]\n"))   // an add written as fma(a, 1, b)
    TIME(17, 128, REP64("v_mul_f64 v[10:11], v[2:3], v[4:5]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n"))   // mul + fma alternating
    TIME(18, 128, REP64("v_add_f64 v[10:11], v[2:3], -v[4:5]\n v_add_f64 v[12:13], v[2:3], v[4:5]\n"))
    TIME(19, 128, REP64("v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n ds_read_b64 v[28:29], v6\n") "s_waitcnt lgkmcnt(0)\n")   // 64 x (fma + ds_read): per 2 instructions
    TIME(20, 192, REP64("v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n ds_read_b64 v[28:29], v6\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(21, 320, REP64("v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n ds_read_b64 v[28:29], v6\n") "s_waitcnt lgkmcnt(0)\n")
    TIME(22, 128, REP64("v_rsq_f64 v[10:11], v[2:3]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n"))          // does an fma hide in the transcendental's shadow?
    TIME(23, 128, REP64("v_cvt_f32_f64 v26, v[2:3]\n v_cvt_f64_i32 v[10:11], v6\n"))
    TIME(24, 128, REP64("v_mov_b64 v[10:11], v[2:3]\n v_mov_b64 v[12:13], v[4:5]\n"))
    TIME(25, 128, REP64("v_lshl_add_u32 v26, v6, 3, v7\n v_add_u32 v27, v6, v7\n"))
    TIME(26, 128, REP64("v_cmp_le_f64 s[24:25], v[2:3], v[4:5]\n v_cmp_gt_f64 vcc, v[2:3], v[4:5]\n"))
    TIME(27, 128, REP64("s_waitcnt lgkmcnt(0)\n s_waitcnt vmcnt(0)\n"))
    TIME(28, 128, REP64("v_ldexp_f64 v[10:11], v[2:3], v7\n v_floor_f64 v[12:13], v[2:3]\n"))
    TIME(29, 128, REP64("v_cndmask_b32 v26, v6, v7, vcc\n v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n"))      // select + fma alternating
    TIME(30, 128, REP64("v_bfi_b32 v26, v6, v7, v6\n v_and_b32 v27, v6, v7\n"))
    TIME(31, 128, REP64("v_fma_f64 v[10:11], -v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], |v[2:3]|, v[4:5], v[12:13]\n"))
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < NT; k++) { out[2 * k] = r[k]; out[2 * k + 1] = cnt[k]; }
}
int main() {
    unsigned long long* out; CHK(hipMalloc(&out, NT * 16));
    const char* names[NT] = {"v_cndmask_b32 (vcc)", "v_cndmask_b32_e64 (sgpr pair mask)", "v_cmp_le_f64 vcc + v_cndmask", "v_cmp_le_f64 + 2 v_cndmask (fp64 select)", "v_cmp_le_f64 + v_addc_co_u32",
        "v_max_f64 / v_min_f64", "v_readlane_b32", "v_writelane_b32", "v_accvgpr_read_b32", "ds_read_b64", "ds_read2_b64", "ds_write_b64", "ds_read_b32", "v_mul_f64 (sgpr operand)",
        "v_fma_f64 (sgpr operand)", "v_fma_f64 a*b+0", "v_fma_f64 a*1+b", "v_mul_f64, v_fma_f64 alternating", "v_add_f64 (neg modifier)", "fma + ds_read_b64", "2 fma + ds_read_b64",
        "4 fma + ds_read_b64", "v_rsq_f64 + v_fma_f64", "v_cvt_f32_f64 / v_cvt_f64_i32", "v_mov_b64", "v_lshl_add_u32 / v_add_u32", "v_cmp_f64 -> sgpr / vcc", "s_waitcnt (nothing pending)",
        "v_ldexp_f64 / v_floor_f64", "v_cndmask + v_fma_f64 alternating", "v_bfi_b32 / v_and_b32", "v_fma_f64 with neg / abs modifiers"};
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_issue, dim3(256), dim3(256), 0, 0, out, iters); CHK(hipDeviceSynchronize()); }
    unsigned long long r[NT * 2]; CHK(hipMemcpy(r, out, NT * 16, hipMemcpyDeviceToHost));
    printf("one wave per SIMD: cycles per INSTRUCTION (average over the group)\n");
    for (int k = 0; k < NT; k++) printf("%-44s %7.2f\n", names[k], (double)r[2 * k] / iters / (double)r[2 * k + 1]);
    return 0;
}
