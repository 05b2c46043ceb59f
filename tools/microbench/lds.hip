// Cost of LDS instructions to ONE wave per SIMD when they are mixed into an fp64 VALU stream (gfx950): cycles per group, and the
// cost attributed to the LDS instruction = group - n_fma * 4.3. All four waves of the workgroup run the same stream (as in the stepper).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define F4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n"
#define F8 F4 "v_fma_f64 v[18:19], v[2:3], v[4:5], v[18:19]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[20:21]\n v_fma_f64 v[22:23], v[2:3], v[4:5], v[22:23]\n v_fma_f64 v[24:25], v[2:3], v[4:5], v[24:25]\n"
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","vcc"
#define TIME(idx, nf, body) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(REP64(body) "s_waitcnt lgkmcnt(0)\n" ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); r[idx] = (__builtin_amdgcn_s_memtime() - t0); nfma[idx] = nf; }
#define NT 20
__global__ __launch_bounds__(256) void k_lds(unsigned long long* out, int iters) {
    extern __shared__ double lds[];
    for (int k = threadIdx.x; k < 16384; k += blockDim.x) lds[k] = 1.0;
    __syncthreads();
    unsigned long long r[NT] = {0}; int nfma[NT] = {0};
    // v6: lane-contiguous 8-byte address (panel row), v7: a 16-byte aligned per-lane table address (scattered), v8: 8-byte-aligned-only table address
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_lshlrev_b32 v6, 3, %0\n"
                 "v_mul_u32_u24 v7, 112, %0\n v_and_b32 v7, 0x3ff0, v7\n v_add_u32 v8, 8, v7\n"
                 :: "v"(threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "v8");
    TIME(0, 8, F8)
    TIME(1, 8, F8 "ds_read_b64 v[28:29], v6\n")
    TIME(2, 8, F8 "ds_read2st64_b64 v[28:31], v6 offset1:4\n")
    TIME(3, 8, F8 "ds_read_b64 v[28:29], v6\n ds_read_b64 v[30:31], v6 offset:2048\n")
    TIME(4, 8, F8 "ds_write_b64 v6, v[2:3]\n")
    TIME(5, 8, F8 "ds_write2st64_b64 v6, v[2:3], v[4:5] offset1:4\n")
    TIME(6, 8, F8 "ds_write_b64 v6, v[2:3]\n ds_write_b64 v6, v[4:5] offset:2048\n")
    TIME(7, 8, F8 "ds_read_b64 v[28:29], v7\n")                          // scattered table read
    TIME(8, 8, F8 "ds_read2_b64 v[28:31], v7 offset1:1\n")               // adjacent pair, 16-byte aligned
    TIME(9, 8, F8 "ds_read2_b64 v[28:31], v8 offset1:1\n")               // adjacent pair, only 8-byte aligned
    TIME(10, 8, F8 "ds_read_b128 v[28:31], v7\n")                        // 16-byte aligned
    TIME(11, 8, F8 "ds_read_b128 v[28:31], v8\n")                        // 8-byte aligned
    TIME(12, 8, F8 "ds_read_b64 v[28:29], v8\n ds_read_b64 v[30:31], v8 offset:8\n")
    TIME(13, 8, F8 "ds_read2_b64 v[28:31], v8 offset1:1\n ds_read2_b64 v[32:35], v8 offset0:26 offset1:27\n")   // lerp2: four values as two read2
    TIME(14, 8, F8 "ds_read_b128 v[28:31], v8\n ds_read_b128 v[32:35], v8 offset:208\n")
    TIME(15, 8, F8 "ds_read_b64 v[28:29], v8\n ds_read_b64 v[30:31], v8 offset:8\n ds_read_b64 v[32:33], v8 offset:208\n ds_read_b64 v[34:35], v8 offset:216\n")
    TIME(16, 8, F8 "ds_read_b32 v28, v6\n")
    TIME(17, 8, F8 "ds_read_b64 v[28:29], v6\n s_waitcnt lgkmcnt(0)\n")  // read + immediate wait: exposed latency
    TIME(18, 4, F4 "ds_read_b64 v[28:29], v6\n" F4 "s_waitcnt lgkmcnt(0)\n")   // 4 fma between read and wait
    TIME(19, 16, F8 "ds_read_b64 v[28:29], v6\n" F8 "s_waitcnt lgkmcnt(0)\n")  // 8 fma between read and wait (16 fma per group)
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < NT; k++) { out[2 * k] = r[k]; out[2 * k + 1] = nfma[k]; }
}
int main() {
    unsigned long long* out; CHK(hipMalloc(&out, NT * 16));
    const char* names[NT] = {"8 fma", "+ ds_read_b64 (row)", "+ ds_read2st64_b64 (two rows)", "+ 2 ds_read_b64 (two rows)", "+ ds_write_b64 (row)", "+ ds_write2st64_b64 (two rows)",
        "+ 2 ds_write_b64 (two rows)", "+ ds_read_b64 (scattered)", "+ ds_read2_b64 adjacent, 16 B aligned", "+ ds_read2_b64 adjacent, 8 B aligned", "+ ds_read_b128, 16 B aligned",
        "+ ds_read_b128, 8 B aligned", "+ 2 ds_read_b64 adjacent", "+ 2 ds_read2_b64 (bilinear corners)", "+ 2 ds_read_b128 (bilinear corners)", "+ 4 ds_read_b64 (bilinear corners)",
        "+ ds_read_b32 (row)", "+ ds_read_b64, wait at once", "4 fma, read, 4 fma, wait (per 8 fma)", "8 fma, read, 8 fma, wait (per 16 fma)"};
    const int iters = 100;
    CHK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 131072, 0, out, iters); CHK(hipDeviceSynchronize()); }
    unsigned long long r[NT * 2]; CHK(hipMemcpy(r, out, NT * 16, hipMemcpyDeviceToHost));
    printf("one workgroup of 4 waves per CU, all running the same stream: cycles per group, and minus 4.31 per fma\n");
    for (int k = 0; k < NT; k++) { const double g = (double)r[2 * k] / iters / 64; printf("%-44s %7.2f   LDS part %7.2f\n", names[k], g, g - 4.31 * (double)r[2 * k + 1]); }
    return 0;
}
