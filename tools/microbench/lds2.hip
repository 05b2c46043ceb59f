// Panel-layout probe for the stepper's RK4 panels (gfx950, 4 waves per CU all running the same stream): two rows per lane as two
// ds_write_b64 / ds_read_b64 (rows 2 KB apart) against one ds_write_b128 / ds_read_b128 on a lane-interleaved pair layout (16 B per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define F4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[14:15]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[16:17]\n"
#define F8 F4 "v_fma_f64 v[18:19], v[2:3], v[4:5], v[18:19]\n v_fma_f64 v[20:21], v[2:3], v[4:5], v[20:21]\n v_fma_f64 v[22:23], v[2:3], v[4:5], v[22:23]\n v_fma_f64 v[24:25], v[2:3], v[4:5], v[24:25]\n"
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","vcc"
#define TIME(idx, body) { __builtin_amdgcn_s_waitcnt(0); unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int k = 0; k < iters; k++) asm volatile(REP64(body) "s_waitcnt lgkmcnt(0)\n" ::: CLOB, "memory"); \
    __builtin_amdgcn_s_waitcnt(0); r[idx] = (__builtin_amdgcn_s_memtime() - t0); }
#define NT 12
__global__ __launch_bounds__(256) void k_lds(unsigned long long* out, int iters) {
    extern __shared__ double lds[];
    for (int k = threadIdx.x; k < 16384; k += blockDim.x) lds[k] = 1.0;
    __syncthreads();
    unsigned long long r[NT] = {0};
    // v6: 8-byte lane stride (row layout), v7: 16-byte lane stride (pair layout)
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_lshlrev_b32 v6, 3, %0\n v_lshlrev_b32 v7, 4, %0\n"
                 "v_mov_b32 v26, 0\n v_mov_b32 v27, 0x3ff00000\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0x3ff00000\n"
                 :: "v"(threadIdx.x) : "v2", "v3", "v4", "v5", "v6", "v7", "v26", "v27", "v28", "v29");
    TIME(0, F8)
    TIME(1, F8 "ds_write_b64 v6, v[2:3]\n ds_write_b64 v6, v[4:5] offset:2048\n")
    TIME(2, F8 "ds_write_b128 v7, v[26:29]\n")
    TIME(3, F8 "ds_write2st64_b64 v6, v[2:3], v[4:5] offset1:4\n")
    TIME(4, F8 "ds_read_b64 v[30:31], v6\n ds_read_b64 v[32:33], v6 offset:2048\n")
    TIME(5, F8 "ds_read_b128 v[30:33], v7\n")
    TIME(6, F8 "ds_write_b64 v6, v[2:3]\n" F8 "ds_write_b64 v6, v[4:5] offset:2048\n")     // writes spaced by 8 fma (per 16 fma)
    TIME(7, F8 "ds_write_b128 v7, v[26:29]\n ds_write_b128 v7, v[26:29] offset:4096\n")  // 4 rows as two b128
    TIME(8, F8 "ds_write_b64 v6, v[2:3]\n ds_write_b64 v6, v[4:5] offset:2048\n ds_write_b64 v6, v[2:3] offset:4096\n ds_write_b64 v6, v[4:5] offset:6144\n")
    TIME(9, F8 "ds_write_b96 v7, v[26:28]\n")
    TIME(10, F8 "ds_write2_b64 v7, v[2:3], v[4:5] offset1:1\n")
    TIME(11, F8 "ds_write_b32 v6, v2\n ds_write_b32 v6, v3 offset:2048\n")
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < NT; k++) out[k] = r[k];
}
int main() {
    unsigned long long* out; CHK(hipMalloc(&out, NT * 8));
    const char* names[NT] = {"8 fma", "+ 2 ds_write_b64 (two rows)", "+ ds_write_b128 (pair layout)", "+ ds_write2st64_b64 (two rows)", "+ 2 ds_read_b64 (two rows)", "+ ds_read_b128 (pair layout)",
        "8 fma, write, 8 fma, write (16 fma)", "+ 2 ds_write_b128 (four rows)", "+ 4 ds_write_b64 (four rows)", "+ ds_write_b96", "+ ds_write2_b64 adjacent (pair layout)", "+ 2 ds_write_b32"};
    const int iters = 100;
    CHK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 131072, 0, out, iters); CHK(hipDeviceSynchronize()); }
    unsigned long long r[NT]; CHK(hipMemcpy(r, out, NT * 8, hipMemcpyDeviceToHost));
    const double base = (double)r[0] / iters / 64;
    for (int k = 0; k < NT; k++) { const double g = (double)r[k] / iters / 64; printf("%-44s %7.2f   over the fma stream %7.2f\n", names[k], g, g - base * (k == 6 ? 2 : 1)); }
    return 0;
}
