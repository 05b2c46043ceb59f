// Two waves per SIMD on gfx950: what does a wave's instruction stream cost while ANOTHER wave on the same SIMD runs a second stream?
// 512-thread workgroups (LDS-limited to one per CU, 256 registers per wave): waves 0-3 run stream X and time it, waves 4-7 run
// stream Y for longer than that. Wave w and wave w + 4 share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","s20","s21","s22","s23","s24","s25","vcc"
#define FMA2 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n"
#define SMOV2 "s_mov_b32 s20, 0x3ff00001\n s_mov_b32 s21, 0x3ff00002\n"
#define VMOV2 "v_mov_b32 v26, v6\n v_add_u32 v27, v6, v7\n"
#define LDSR2 "ds_read_b64 v[28:29], v6\n ds_read_b64 v[30:31], v6 offset:2048\n"
#define LDSW2 "ds_write_b64 v6, v[2:3]\n ds_write_b64 v6, v[4:5] offset:2048\n"
#define MIX4 "v_fma_f64 v[10:11], v[2:3], v[4:5], v[10:11]\n s_mov_b32 s20, 0x3ff00001\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[12:13]\n v_add_u32 v27, v6, v7\n"
#define RSQ2 "v_rsq_f64 v[10:11], v[2:3]\n v_rcp_f64 v[12:13], v[4:5]\n"
#define RUN(body) for (int k = 0; k < n; k++) asm volatile(body ::: CLOB, "memory");
__device__ void stream(int which, int n) {
    switch (which) {
        case 0: break;
        case 1: RUN(REP64(FMA2)) break;
        case 2: RUN(REP64(SMOV2)) break;
        case 3: RUN(REP64(VMOV2)) break;
        case 4: RUN(REP64(LDSR2) "s_waitcnt lgkmcnt(0)\n") break;
        case 5: RUN(REP64(LDSW2) "s_waitcnt lgkmcnt(0)\n") break;
        case 6: RUN(REP16(MIX4) REP16(MIX4)) break;
        case 7: RUN(REP64(RSQ2)) break;
    }
}
__global__ __launch_bounds__(512) void k_issue3(unsigned long long* out, int iters, int X, int Y) {
    __shared__ double lds[18000];   // 144 KB: one workgroup per CU
    for (int k = threadIdx.x; k < 18000; k += blockDim.x) lds[k] = 1.0;
    __syncthreads();
    asm volatile("v_mov_b32 v2, 0\n v_mov_b32 v3, 0x3ff00000\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0x3ff00000\n v_lshlrev_b32 v6, 3, %0\n v_mov_b32 v7, 0\n"
                 :: "v"(threadIdx.x & 255) : "v2", "v3", "v4", "v5", "v6", "v7");
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (role == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        stream(X, iters);
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
    } else {
        stream(Y, iters * 3);
    }
    if (lds[threadIdx.x] == 123.0) out[1] = 1;
}
int main() {
    unsigned long long* out; CHK(hipMalloc(&out, 16));
    const char* nm[8] = {"(idle)", "v_fma_f64", "s_mov_b32", "32-bit VALU", "ds_read_b64", "ds_write_b64", "fma,s_mov,fma,v_add mix", "v_rsq/v_rcp_f64"};
    const int iters = 100;
    printf("two waves per SIMD: cycles per instruction of stream X while the other wave of the SIMD runs stream Y\n");
    const int combos[][2] = {{1, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 7}, {2, 0}, {2, 2}, {2, 1}, {3, 0}, {3, 3}, {4, 0}, {4, 4}, {4, 1}, {5, 0}, {5, 5}, {5, 1}, {6, 0}, {6, 6}, {7, 0}, {7, 7}, {7, 1}};
    for (auto& c : combos) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_issue3, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1]); CHK(hipDeviceSynchronize()); }
        unsigned long long r[2]; CHK(hipMemcpy(r, out, 16, hipMemcpyDeviceToHost));
        printf("X = %-24s Y = %-24s %7.2f\n", nm[c[0]], nm[c[1]], (double)r[0] / iters / 128.0);
    }
    return 0;
}
