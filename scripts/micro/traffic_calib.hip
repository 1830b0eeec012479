// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the path tracer's
// queues use (SoA fields of 8 B and 4 B per lane), as MI355X_MICROARCH.md asks ("other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel streams over a
// buffer far larger than the 256 MiB Infinity Cache and moves a known number of bytes, printed here; run
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./traffic_calib      (and again with WRITE_SIZE)
// and divide the counter (KB) by the printed bytes.  Results: profiles/r01/traffic_calibration.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double double2v __attribute__((ext_vector_type(2)));

// One element of type E per lane and iteration, a wave touches 64 consecutive elements; `shift` elements of
// offset make every wave's 64-element run straddle cache lines the way compacted queue output does.
template <typename E>
__global__ void __launch_bounds__(256) k_read(const E* __restrict__ in, size_t n, size_t shift, double* sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + shift < n; i += (size_t)gridDim.x * 256) {
        const E v = in[i + shift];
        acc += (double)(*(const uint32_t*)&v & 1u);
    }
    if (acc == 12345.678) sink[0] = acc;
}
template <typename E>
__global__ void __launch_bounds__(256) k_write(E* __restrict__ out, size_t n, size_t shift, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + shift < n; i += (size_t)gridDim.x * 256) {
        E v;
        uint32_t* w = (uint32_t*)&v;
        for (unsigned k = 0; k < sizeof(E) / 4; k++) w[k] = seed + (uint32_t)i;
        out[i + shift] = v;
    }
}
// The queues' pattern: 11 SoA fields (ten f64, one u32) of the same dense index read, and written at a
// wave-granular offset that is not a multiple of a cache line (compaction: survivors of a wave go to the
// next free slots of the output tile).
__global__ void __launch_bounds__(256) k_queue_copy(const double* __restrict__ in, double* __restrict__ out, const uint32_t* __restrict__ id_in,
                                                    uint32_t* __restrict__ id_out, size_t n_paths, size_t field_stride, size_t shift) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + shift < n_paths; i += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int f = 0; f < 10; f++) out[f * field_stride + i + shift] = in[f * field_stride + i];
        id_out[i + shift] = id_in[i];
    }
}

int main() {
    const size_t BYTES = (size_t)2 << 30;  // 2 GiB per buffer
    void *a, *b;
    double* sink;
    CHECK(hipMalloc(&a, BYTES));
    CHECK(hipMalloc(&b, BYTES));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, BYTES));
    CHECK(hipMemset(b, 0, BYTES));
    const int blocks = 256 * 16;
    auto report = [](const char* name, double rd, double wr) { printf("%-28s reads %.0f B  writes %.0f B\n", name, rd, wr); };

    hipLaunchKernelGGL(k_read<uint32_t>, dim3(blocks), dim3(256), 0, 0, (const uint32_t*)a, BYTES / 4, (size_t)0, sink);
    report("k_read<uint32_t> shift 0", (double)BYTES, 0);
    hipLaunchKernelGGL(k_read<double>, dim3(blocks), dim3(256), 0, 0, (const double*)a, BYTES / 8, (size_t)0, sink);
    report("k_read<double> shift 0", (double)BYTES, 0);
    hipLaunchKernelGGL(k_read<double2v>, dim3(blocks), dim3(256), 0, 0, (const double2v*)a, BYTES / 16, (size_t)0, sink);
    report("k_read<double2> shift 0", (double)BYTES, 0);
    hipLaunchKernelGGL(k_read<double>, dim3(blocks), dim3(256), 0, 0, (const double*)a, BYTES / 8, (size_t)5, sink);
    report("k_read<double> shift 5", (double)BYTES - 40, 0);
    CHECK(hipDeviceSynchronize());

    hipLaunchKernelGGL(k_write<uint32_t>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)b, BYTES / 4, (size_t)0, 7u);
    report("k_write<uint32_t> shift 0", 0, (double)BYTES);
    hipLaunchKernelGGL(k_write<double>, dim3(blocks), dim3(256), 0, 0, (double*)b, BYTES / 8, (size_t)0, 7u);
    report("k_write<double> shift 0", 0, (double)BYTES);
    hipLaunchKernelGGL(k_write<double2v>, dim3(blocks), dim3(256), 0, 0, (double2v*)b, BYTES / 16, (size_t)0, 7u);
    report("k_write<double2> shift 0", 0, (double)BYTES);
    hipLaunchKernelGGL(k_write<double>, dim3(blocks), dim3(256), 0, 0, (double*)b, BYTES / 8, (size_t)5, 7u);
    report("k_write<double> shift 5", 0, (double)BYTES - 40);
    CHECK(hipDeviceSynchronize());

    const size_t n_paths = (size_t)24 << 20;  // 24 Mi paths x 84 B = 2.0 GB each way
    const size_t stride = n_paths;
    hipLaunchKernelGGL(k_queue_copy, dim3(blocks), dim3(256), 0, 0, (const double*)a, (double*)b, (const uint32_t*)((const double*)a + 10 * stride),
                       (uint32_t*)((double*)b + 10 * stride), n_paths, stride, (size_t)0);
    report("k_queue_copy shift 0", 84.0 * n_paths, 84.0 * n_paths);
    hipLaunchKernelGGL(k_queue_copy, dim3(blocks), dim3(256), 0, 0, (const double*)a, (double*)b, (const uint32_t*)((const double*)a + 10 * stride),
                       (uint32_t*)((double*)b + 10 * stride), n_paths, stride, (size_t)5);
    report("k_queue_copy shift 5", 84.0 * (n_paths - 5), 84.0 * (n_paths - 5));
    CHECK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
