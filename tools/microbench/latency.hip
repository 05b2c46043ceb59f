// Latencies seen by ONE wave per SIMD on gfx950 (the stepping kernel's regime): dependent ds_read_b64, s_load (scalar cache hit),
// global_load (L2 hit), and the issue cost of fp64 VALU / v_readlane / s_mov. Build: hipcc --offload-arch=gfx950 -O3 latency.hip -o latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_lat(const double* g, const int* gi, unsigned long long* out, int iters) {
    __shared__ double lds[4096];
    for (int k = threadIdx.x; k < 4096; k += 256) lds[k] = (double)((k * 7 + 1) & 4095);
    __syncthreads();
    typedef __attribute__((address_space(3))) const double* lp;
    lp L = (lp)lds;
    const int t = threadIdx.x;
    unsigned long long r[8] = {0};
    // 1. dependent LDS chain: idx = (int)lds[idx]
    int idx = t;
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) idx = (int)L[idx & 4095];
    asm volatile("" :: "v"(idx));
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r[0] = t1 - t0;
    // 2. dependent scalar-load chain (uniform): j = gi[j]
    int j = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) j = __builtin_amdgcn_readfirstlane(((__attribute__((address_space(4))) const int*)gi)[j & 1023]);
    asm volatile("" :: "s"(j));
    t1 = __builtin_amdgcn_s_memtime();
    r[1] = t1 - t0;
    // 3. dependent global-load chain (per lane, L2 resident)
    int m = t;
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) m = gi[(m + 64) & 1023];
    asm volatile("" :: "v"(m));
    t1 = __builtin_amdgcn_s_memtime();
    r[2] = t1 - t0;
    // 4. dependent fp64 fma chain, 5. independent fp64 fma x4
    double a = g[t], b = g[t + 256], c0 = a, c1 = b, c2 = a + 1, c3 = b + 1;
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) c0 = __builtin_fma(c0, a, b);
    asm volatile("" :: "v"(c0));
    t1 = __builtin_amdgcn_s_memtime();
    r[3] = t1 - t0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) { c0 = __builtin_fma(c0, a, b); c1 = __builtin_fma(c1, a, b); c2 = __builtin_fma(c2, a, b); c3 = __builtin_fma(c3, a, b); }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    t1 = __builtin_amdgcn_s_memtime();
    r[4] = t1 - t0;
    // 6. LDS read -> dependent fma -> next address (read + convert + use chain typical of a table lookup): covered by 1
    // 7. independent ds_read_b64 x8 then one wait
    double s = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        double v0 = L[(t + k) & 4095], v1 = L[(t + k + 300) & 4095], v2 = L[(t + k + 600) & 4095], v3 = L[(t + k + 900) & 4095];
        double v4 = L[(t + k + 1200) & 4095], v5 = L[(t + k + 1500) & 4095], v6 = L[(t + k + 1800) & 4095], v7 = L[(t + k + 2100) & 4095];
        s += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
    }
    asm volatile("" :: "v"(s));
    t1 = __builtin_amdgcn_s_memtime();
    r[5] = t1 - t0;
    // 8. v_readlane pairs feeding an fp64 multiply
    double q = a;
    const int cv = gi[t & 63];
    t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < iters; k++) {
        const unsigned lo = __builtin_amdgcn_readlane(cv, (k * 2) & 63), hi = __builtin_amdgcn_readlane(cv, (k * 2 + 1) & 63);
        const double cst = __builtin_bit_cast(double, ((unsigned long long)(hi & 0xfffff | 0x3ff00000) << 32) | lo);
        q = q * cst;
    }
    asm volatile("" :: "v"(q));
    t1 = __builtin_amdgcn_s_memtime();
    r[6] = t1 - t0;
    if (blockIdx.x == 0 && t == 0) for (int k = 0; k < 8; k++) out[k] = r[k];
}

int main() {
    const int iters = 2000;
    std::vector<double> hg(512);
    for (int k = 0; k < 512; k++) hg[k] = 1.0 + 1e-9 * k;
    std::vector<int> hi(1024);
    for (int k = 0; k < 1024; k++) hi[k] = (k * 37 + 11) & 1023;
    double* g; int* gi; unsigned long long* out;
    CHK(hipMalloc(&g, 512 * 8)); CHK(hipMalloc(&gi, 1024 * 4)); CHK(hipMalloc(&out, 64));
    CHK(hipMemcpy(g, hg.data(), 512 * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(gi, hi.data(), 1024 * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(256), dim3(256), 0, 0, g, gi, out, iters);
        CHK(hipDeviceSynchronize());
    }
    unsigned long long r[8];
    CHK(hipMemcpy(r, out, 64, hipMemcpyDeviceToHost));
    const char* names[] = {"dependent ds_read_b64 -> cvt -> address (per link)", "dependent s_load_dword (K$ hit, per link)", "dependent global_load_dword (L2 hit, per link)",
                           "dependent v_fma_f64 (per op)", "independent v_fma_f64 (per op, 4 chains)", "8 independent ds_read_b64 + 1 wait + 8 adds (per group)",
                           "2 v_readlane + v_mul_f64 dependent chain (per link)"};
    const double div[] = {1, 1, 1, 1, 4, 1, 1};
    for (int k = 0; k < 7; k++) printf("%-62s %8.1f cycles\n", names[k], (double)r[k] / iters / div[k]);
    return 0;
}
