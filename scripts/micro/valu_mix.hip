// Calibration for the VALU-issue roofline (bench.py, scripts/make_valu.py): what one wave64 instruction of each class
// the path tracer is made of costs a gfx950 SIMD, in cycles, and which rocprofv3 SQ_INSTS_VALU_* counter it lands in.
//
// One kernel per instruction class (its name is the class, so a --pmc pass over this program gives the counters per
// class), 8 waves per SIMD on every CU, 16 independent instructions per loop iteration written in inline asm so that
// the compiler can neither fuse nor drop them.  The clock the chip holds under each load is measured inside the
// kernel: delta s_memtime (shader cycles) / delta s_memrealtime (100 MHz), median over the waves.
//
//   hipcc -O2 --offload-arch=gfx950 -o bin/valu_mix scripts/micro/valu_mix.hip
//   bin/valu_mix                 -> table on stdout (profiles/<round>/valu_mix.txt)
//   rocprofv3 --pmc ... -- bin/valu_mix    -> counters per class (profiles/<round>/valu_mix_counters.txt)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 8192;
constexpr int PER_ITER = 16;

struct Stamp {
    unsigned long long cycles, ticks;
};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL_BEGIN(name, MASK)                                                                            \
    __global__ void __launch_bounds__(256) name(double* out, Stamp* stamps, double seed, uint32_t iseed) { \
        double d[16];                                                                                  \
        float f[16];                                                                                   \
        uint32_t u[16];                                                                                \
        uint64_t q[16];                                                                                \
        for (int k = 0; k < 16; k++) {                                                                 \
            d[k] = seed + 0.001 * (double)(threadIdx.x + 7 * k);                                       \
            f[k] = (float)d[k];                                                                        \
            u[k] = iseed + threadIdx.x * 16u + (uint32_t)k;                                            \
            q[k] = ((uint64_t)u[k] << 20) | 5u;                                                        \
        }                                                                                              \
        constexpr int mask_ = (MASK);                                                                  \
        const double dm = 1.0000001, dc = 1e-9;                                                        \
        const float fm = 1.0001f, fc = 1e-6f;                                                          \
        (void)dm; (void)dc; (void)fm; (void)fc;                                                        \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime(); \
        for (int i = 0; i < ITERS; i++) {
#define KERNEL_END                                                                                     \
        }                                                                                              \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + threadIdx.x / 64] = Stamp{c1 - c0, t1 - t0}; \
        double s = 0;                                                                                  \
        for (int k = 0; k < 16; k++)                                                                   \
            s += (mask_ & 1 ? d[k] : 0.0) + (mask_ & 2 ? (double)f[k] : 0.0) + (mask_ & 4 ? (double)u[k] : 0.0) + (mask_ & 8 ? (double)q[k] : 0.0); \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                       \
    }

// ---- f64 arithmetic
KERNEL_BEGIN(add_f64, 1)
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mul_f64, 1)
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dm));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(fma_f64, 1)
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dm), "v"(dc));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(max_f64, 1)
#define X(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[k]) : "v"(d[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cmp_f64, 1)
#define X(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[k]), "v"(d[(k + 1) & 15]) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(rcp_f64, 1)
#define X(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(rsq_f64, 1)
#define X(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(sqrt_f64, 1)
#define X(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(div_scale_f64, 1)
#define X(k) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[k]) : "v"(dm) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(div_fmas_f64, 1)
#define X(k) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dm), "v"(dc) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(div_fixup_f64, 1)
#define X(k) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(dm), "v"(dc));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(ldexp_f64, 5)
#define X(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(u[k] & 1u));
REP16(X)
#undef X
KERNEL_END
// ---- 32-bit floating point
KERNEL_BEGIN(add_f32, 2)
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(fc));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mul_f32, 2)
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(fm));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(fma_f32, 2)
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fm), "v"(fc));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(pk_fma_f32, 1)
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[k]) : "v"(dm));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(max_f32, 2)
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(rcp_f32, 2)
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
REP16(X)
#undef X
KERNEL_END
// ---- integer, moves, selects, conversions
KERNEL_BEGIN(xor_b32, 4)
#define X(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(add_u32, 4)
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mul_lo_u32, 4)
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mad_u64_u32, 12)
#define X(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(u[k]), "v"(u[(k + 1) & 15]) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(lshlrev_b64, 8)
#define X(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mov_b32, 4)
#define X(k) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
// (VOP2 form with a VCC that no VALU instruction has written: ~22 cycles each, a case compiled code never has - kept as a warning)
KERNEL_BEGIN(cndmask_b32, 4)
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cvt_f32_f64, 3)
#define X(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cvt_f64_u32, 5)
#define X(k) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[k]) : "v"(u[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cvt_f32_ubyte0, 6)
#define X(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[k]) : "v"(u[k]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(readlane_b32, 4)
#define X(k) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(u[k])); asm volatile("" :: "s"(s_)); }
REP16(X)
#undef X
KERNEL_END

KERNEL_BEGIN(cndmask_b32_sgpr, 4)
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : "s10", "s11");
REP16(X)
#undef X
KERNEL_END
// a select of a double as the compiler writes it: one compare, the wait state it must leave (s_nop 1), two VOP2
// v_cndmask_b32 that read the mask from VCC (3 VALU instructions per slot)
KERNEL_BEGIN(cmp_cndmask2_f64, 5)
#define X(k) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %0, %3, %0, vcc" : "+v"(u[k]) : "v"(d[k]), "v"(d[(k + 1) & 15]), "v"(u[(k + 1) & 15]) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(min_max_f32_pair, 2)
#define X(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[k]) : "v"(f[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(mul_hi_u32, 4)
#define X(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(writelane_b32, 4)
#define X(k) asm volatile("v_writelane_b32 %0, s10, 3" : "+v"(u[k]) : : "s10");
REP16(X)
#undef X
KERNEL_END
// ---- what makes a select dear: the mask operand (VCC of the VOP2 form against an SGPR pair of the VOP3 form)
KERNEL_BEGIN(cndmask_b32_e64_vcc, 4)
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cmp_cndmask2_f64_sgpr, 5)
#define X(k) asm volatile("v_cmp_lt_f64_e64 s[10:11], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[10:11]\n\tv_cndmask_b32_e64 %0, %3, %0, s[10:11]" : "+v"(u[k]) : "v"(d[k]), "v"(d[(k + 1) & 15]), "v"(u[(k + 1) & 15]) : "s10", "s11");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(addc_co_u32_vcc, 4)
#define X(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : "vcc");
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cndmask_b32_vcc_salu, 4)
asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
REP16(X)
#undef X
KERNEL_END
KERNEL_BEGIN(cmp_f32_cndmask_vcc, 6)
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[k]) : "v"(f[k]), "v"(f[(k + 1) & 15]), "v"(u[(k + 1) & 15]) : "vcc");
REP16(X)
#undef X
KERNEL_END

typedef void (*kernel_t)(double*, Stamp*, double, uint32_t);

static int run(const char* name, kernel_t kfn, double* out, Stamp* stamps, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 3; warm++) hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), 0, 0, out, stamps, 1.5, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), 0, 0, out, stamps, 1.5, 12345u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> host((size_t)blocks * 4);
    CHECK(hipMemcpy(host.data(), stamps, host.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz, wave_cycles;
    for (const Stamp& s : host)
        if (s.ticks > 0) {
            ghz.push_back((double)s.cycles / (double)s.ticks * 0.1);
            wave_cycles.push_back((double)s.cycles);
        }
    std::sort(ghz.begin(), ghz.end());
    std::sort(wave_cycles.begin(), wave_cycles.end());
    const double clock = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double insts_per_simd = waves_per_simd * ITERS * PER_ITER;
    const double ns_per_inst = ms * 1e6 / insts_per_simd;
    // from the waves' own cycle counters: a SIMD's waves run side by side for the loop's length
    const double cyc_in_kernel = wave_cycles.empty() ? 0.0 : wave_cycles[wave_cycles.size() / 2] / (ITERS * PER_ITER * waves_per_simd);
    printf("%-16s %8.3f ms  %7.3f ns/inst/SIMD  clock %5.3f GHz  %6.2f cycles/inst/SIMD (events x clock)  %6.2f (in-kernel cycles)\n", name, ms,
           ns_per_inst, clock, ns_per_inst * clock, cyc_in_kernel);
    return 0;
}

int main() {
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    double* out;
    Stamp* stamps;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
    CHECK(hipMalloc(&stamps, (size_t)blocks * 4 * sizeof(Stamp)));
    printf("# one wave64 instruction per class, %d per iteration x %d iterations, 8 waves per SIMD, %d SIMDs\n", PER_ITER, ITERS, 1024);
#define RUN(name) if (run(#name, name, out, stamps, blocks)) return 1;
    RUN(add_f64) RUN(mul_f64) RUN(fma_f64) RUN(max_f64) RUN(cmp_f64) RUN(rcp_f64) RUN(rsq_f64) RUN(sqrt_f64)
    RUN(div_scale_f64) RUN(div_fmas_f64) RUN(div_fixup_f64) RUN(ldexp_f64)
    RUN(add_f32) RUN(mul_f32) RUN(fma_f32) RUN(pk_fma_f32) RUN(max_f32) RUN(rcp_f32)
    RUN(xor_b32) RUN(add_u32) RUN(mul_lo_u32) RUN(mad_u64_u32) RUN(lshlrev_b64) RUN(mov_b32) RUN(cndmask_b32)
    RUN(cvt_f32_f64) RUN(cvt_f64_u32) RUN(cvt_f32_ubyte0) RUN(readlane_b32)
    RUN(cndmask_b32_sgpr) RUN(cmp_cndmask2_f64) RUN(min_max_f32_pair) RUN(mul_hi_u32) RUN(writelane_b32)
    RUN(cndmask_b32_e64_vcc) RUN(cmp_cndmask2_f64_sgpr) RUN(addc_co_u32_vcc) RUN(cndmask_b32_vcc_salu) RUN(cmp_f32_cndmask_vcc)
    return 0;
}
