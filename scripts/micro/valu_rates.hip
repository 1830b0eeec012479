// Calibration: issue cost of the VALU instructions the path tracer is made of (f64 add/mul/fma, div, sqrt,
// 32x32->64 multiply), per SIMD, on gfx950.  Prints wave-instructions per SIMD-cycle at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double seed, uint32_t iseed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const double m = 1.0000001, c = 1e-9;
    uint32_t u0 = iseed + threadIdx.x, u1 = u0 * 3 + 1;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c); }
        if (OP == 1) { a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; }
        if (OP == 2) { a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; }
        if (OP == 3) { a0 = c / a0 + 1.0; a1 = c / a1 + 1.0; a2 = c / a2 + 1.0; a3 = c / a3 + 1.0; }
        if (OP == 4) { a0 = sqrt(a0) + 1.0; a1 = sqrt(a1) + 1.0; a2 = sqrt(a2) + 1.0; a3 = sqrt(a3) + 1.0; }
        if (OP == 5) { uint64_t p = (uint64_t)u0 * 0xD2511F53u; uint64_t q = (uint64_t)u1 * 0xCD9E8D57u; u0 = (uint32_t)(p >> 32) ^ (uint32_t)q; u1 = (uint32_t)(q >> 32) ^ (uint32_t)p; 
                       p = (uint64_t)u0 * 0xD2511F53u; q = (uint64_t)u1 * 0xCD9E8D57u; u0 = (uint32_t)(p >> 32) ^ (uint32_t)q; u1 = (uint32_t)(q >> 32) ^ (uint32_t)p; }
        if (OP == 6) { a0 = fmax(fmin(a0, a1), c); a1 = fmin(fmax(a1, a2), m); a2 = fmax(fmin(a2, a3), c); a3 = fmin(fmax(a3, a0), m); }
        if (OP == 7) { a0 = a0 < a1 ? a2 : a3; a1 = a1 < a2 ? a3 : a0; a2 = a2 < a3 ? a0 : a1; a3 = a3 < a0 ? a1 : a2; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (double)(u0 ^ u1);
}

template <int OP>
int run(const char* name, double ops_per_iter, double* out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, 12345u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double waves = blocks * 4.0;
    double wave_ops = waves * ITERS * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f G source-ops/s per wave-op => %.3f ns per op per SIMD (1024 SIMDs)\n", name, ms, wave_ops / ms * 1e-6,
           ms * 1e6 / (wave_ops / 1024.0));
    return 0;
}

int main() {
    double* out;
    int blocks = 256 * 8;  // 8 blocks per CU = 8 waves per SIMD
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
    run<0>("fma f64 (4 per iter)", 4, out, blocks);
    run<1>("mul f64 (4 per iter)", 4, out, blocks);
    run<2>("add f64 (4 per iter)", 4, out, blocks);
    run<3>("div+add f64 (4 per iter)", 4, out, blocks);
    run<4>("sqrt+add f64 (4 per iter)", 4, out, blocks);
    run<5>("philox-like half round x4", 4, out, blocks);
    run<6>("fmin/fmax f64 (8 per iter)", 8, out, blocks);
    run<7>("cmp+select f64 (4 per iter)", 4, out, blocks);
    return 0;
}
