// rt_multi_*: one image over the GPUs of a node from ONE host process (include/rt_mi355x.h).
//
// Replaces the rayon row loop of the reference (src/main.rs:117-154) at node scale, behind the C ABI, so that
// bin/raytracer --gpus N and any FFI host get every GPU without Python or torch.  The scene is replicated; for
// every band of rows device g traces the sample indices split_samples(spp, N, g) of every pixel into its own f64
// accumulation buffer (rt_render_accum_device from a host thread per device), the buffers are summed onto device
// 0 by ONE RCCL ncclReduce(sum, f64) per band, and device 0 quantises.  The reduce and the quantisation of band
// k run on a second stream per device while band k + 1 is being traced (two accumulation buffers per device).
// With fewer samples than GPUs the rows of a band are split instead and nothing is reduced.
//
// RCCL is loaded with dlopen the first time a context with more than one distinct device is made (a process that
// has torch loaded already holds a copy; linking another one in would clash - Rccl::load says which copy and why).  RT_REDUCE=p2p in the environment
// (and any device list that names a GPU twice, which RCCL does not allow) takes the copy-and-add path instead:
// every peer's buffer is copied to device 0 over its own xGMI link (hipMemcpyPeerAsync) and added there.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "device/rt_device.h"

namespace {

struct MultiError {
    int code;
    std::string msg;
};
#define MHIP(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) throw MultiError{RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)}; \
    } while (0)

thread_local std::string g_multi_error;

// ---- RCCL through dlopen
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    std::string path;  // the file in use
    bool load() {
        if (handle) return true;
        // ONE copy of the ROCm stack per process.  PyTorch wheels bundle their own (torch/lib/libamdhip64.so, librccl.so, ...)
        // and ask for it by unversioned file names that match no SONAME of the system copies (libamdhip64.so.7,
        // librccl.so.1): a process that has /opt/rocm's RCCL - and with it /opt/rocm's HIP runtime, rocm_smi, ... - and then
        // imports torch holds every library twice and aborts at exit ("free(): invalid pointer"; reproduced without a GPU,
        // tests/test_host_io.py).  So RCCL is taken, in this order, from: a copy the process holds already (RTLD_NOLOAD);
        // the file RT_RCCL_LIB names; the directory of the HIP runtime THIS process runs on (torch's if torch's runtime is the
        // one loaded - the Python bindings load it first -, /opt/rocm's for a plain FFI host); the loader's search path.
        // And always RTLD_LOCAL: with torch's own librccl.so opened RTLD_GLOBAL before `import torch`, the process still
        // ended in "double free or corruption" at exit although it held one copy of every library - RCCL's global symbols
        // interposed on definitions of the torch libraries loaded after it.  Its entry points are looked up with dlsym.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);  // (never RTLD_GLOBAL: that would promote the copy)
            if (handle) break;
        }
        // (dlerror() hands its text out once and clears it: read once, right after the dlopen that failed)
        std::string last_failure;
        auto open = [&](const char* file) {
            handle = dlopen(file, RTLD_NOW | RTLD_LOCAL);  // (local: its symbols must not interpose on libraries loaded later, see below)
            if (!handle) {
                const char* e = dlerror();
                last_failure = e ? e : std::string(file) + " not found";
            }
        };
        if (const char* env = std::getenv("RT_RCCL_LIB"); !handle && env && env[0]) open(env);
        if (!handle) {
            Dl_info hip;
            void* sym = dlsym(RTLD_DEFAULT, "hipGetDeviceCount");
            if (sym && dladdr(sym, &hip) && hip.dli_fname) {
                std::string dir(hip.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) {
                    dir.resize(slash + 1);
                    for (const char* n : {"librccl.so.1", "librccl.so"}) {
                        if (handle) break;
                        if (access((dir + n).c_str(), R_OK) == 0) open((dir + n).c_str());
                    }
                }
            }
        }
        for (const char* n : names) {
            if (handle) break;
            open(n);
        }
        if (!handle) {
            why = "cannot load RCCL: " + (last_failure.empty() ? std::string("librccl.so.1 not found") : last_failure);
            return false;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(handle, "ncclReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Reduce || !GetErrorString) {
            why = "RCCL library lacks ncclCommInitAll / ncclReduce";
            handle = nullptr;
            return false;
        }
        Dl_info where;
        path = dladdr((void*)Reduce, &where) && where.dli_fname ? where.dli_fname : "";
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

__global__ void __launch_bounds__(256) k_add_f64(double* __restrict__ dst, const double* __restrict__ src, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// All device threads meet here after a band's tracing and agree on whether anyone failed: a thread that skipped
// its collective call would leave the others waiting on the GPU for ever.
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0, generation = 0;
    bool failed = false, round_failed = false;
    bool meet(bool ok) {  // returns true when every thread of this round was ok
        std::unique_lock<std::mutex> lock(m);
        if (!ok) failed = true;
        const int gen = generation;
        if (++arrived == n) {
            arrived = 0;
            round_failed = failed;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
        return !round_failed;
    }
};

struct Device {
    int id = 0;
    rt_context* ctx = nullptr;
    hipStream_t comm_stream = nullptr;   // reduce, quantise, copies to the host
    double* accum[2] = {nullptr, nullptr};
    double* staging = nullptr;           // root, copy-and-add path: a peer's buffer
    uint8_t* rgb8[2] = {nullptr, nullptr};
    size_t accum_elems = 0;
    hipEvent_t buffer_free[2] = {nullptr, nullptr};  // recorded on comm_stream when a band's buffer has been consumed
    ncclComm_t comm = nullptr;
};

}  // namespace

struct rt_multi {
    std::vector<Device> dev;
    bool use_rccl = false;
    uint64_t max_paths = 0;
    std::string error;
};

namespace {

void split_range(uint32_t total, uint32_t parts, uint32_t part, uint32_t first, uint32_t* begin, uint32_t* count) {
    const uint32_t base = total / parts, extra = total % parts;
    *count = base + (part < extra ? 1u : 0u);
    *begin = first + part * base + std::min(part, extra);
}

template <typename F>
int multi_guard(rt_multi* m, F&& f) {
    try {
        f();
        return RT_OK;
    } catch (const MultiError& e) {
        g_multi_error = e.msg;
        if (m) m->error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_multi_error = e.what();
        if (m) m->error = e.what();
        return RT_ERR_INVALID;
    }
}

void ensure_buffers(Device& d, size_t elems, bool root, bool p2p) {
    if (elems <= d.accum_elems) return;
    MHIP(hipSetDevice(d.id));
    for (int k = 0; k < 2; k++) {
        if (d.accum[k]) (void)hipFree(d.accum[k]);
        d.accum[k] = nullptr;
        MHIP(hipMalloc((void**)&d.accum[k], elems * sizeof(double)));
        if (d.rgb8[k]) (void)hipFree(d.rgb8[k]);
        d.rgb8[k] = nullptr;
        MHIP(hipMalloc((void**)&d.rgb8[k], elems));
    }
    if (root && p2p) {
        if (d.staging) (void)hipFree(d.staging);
        d.staging = nullptr;
        MHIP(hipMalloc((void**)&d.staging, elems * sizeof(double)));
    }
    d.accum_elems = elems;
}

void add_stats(rt_stats* total, const rt_stats& part, bool same_device) {
    total->rays += part.rays;
    total->paths += part.paths;
    // a device's bands run one after the other; the devices run side by side
    total->ms_total = same_device ? total->ms_total + part.ms_total : std::max(total->ms_total, part.ms_total);
    total->ms_extend += part.ms_extend;
    total->ms_shade += part.ms_shade;
    total->ms_tail += part.ms_tail;
    total->extend_launches += part.extend_launches;
    total->shade_launches += part.shade_launches;
    total->tail_launches += part.tail_launches;
    total->rays_tail += part.rays_tail;
    total->paths_tail += part.paths_tail;
    total->paths_first += part.paths_first;
    total->batches += part.batches;
    total->extend_kernel = part.extend_kernel;
    total->max_depth_reached = std::max(total->max_depth_reached, part.max_depth_reached);
}

}  // namespace

extern "C" {

void rt_split_samples(uint32_t total, uint32_t parts, uint32_t part, uint32_t first, uint32_t* begin, uint32_t* count) {
    uint32_t b = first, c = 0;
    if (parts > 0 && part < parts) split_range(total, parts, part, first, &b, &c);
    if (begin) *begin = b;
    if (count) *count = c;
}

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->error.c_str() : g_multi_error.c_str(); }

const char* rt_multi_rccl_library(void) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) {
        g_multi_error = g_rccl.why;
        return nullptr;
    }
    return g_rccl.path.c_str();
}

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    int before = -1;
    if (hipGetDevice(&before) != hipSuccess) before = -1;
    for (Device& d : m->dev) {
        if (!d.ctx) continue;  // rt_multi_create stopped before this device
        (void)hipSetDevice(d.id);
        if (d.comm_stream) (void)hipStreamSynchronize(d.comm_stream);
        if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
        for (int k = 0; k < 2; k++) {
            if (d.accum[k]) (void)hipFree(d.accum[k]);
            if (d.buffer_free[k]) (void)hipEventDestroy(d.buffer_free[k]);
        }
        if (d.staging) (void)hipFree(d.staging);
        for (int k = 0; k < 2; k++)
            if (d.rgb8[k]) (void)hipFree(d.rgb8[k]);
        if (d.comm_stream) (void)hipStreamDestroy(d.comm_stream);
        if (d.ctx) rt_destroy(d.ctx);
    }
    if (before >= 0) (void)hipSetDevice(before);
    (void)hipGetLastError();
    delete m;
}

int rt_multi_create(const int* device_ids, int n_devices, const rt_options* options, rt_multi** out) {
    if (!out) {
        g_multi_error = "rt_multi_create: null out pointer";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    std::unique_ptr<rt_multi, void (*)(rt_multi*)> m(new rt_multi, rt_multi_destroy);
    const int rc = multi_guard(nullptr, [&] {
        if (!device_ids || n_devices < 1 || n_devices > 64) throw MultiError{RT_ERR_INVALID, "rt_multi_create: 1 to 64 device ids are needed"};
        int before = -1;
        if (hipGetDevice(&before) != hipSuccess) before = -1;
        bool distinct = true;
        for (int a = 0; a < n_devices; a++)
            for (int b = 0; b < a; b++)
                if (device_ids[a] == device_ids[b]) distinct = false;
        m->dev.resize((size_t)n_devices);
        for (int g = 0; g < n_devices; g++) {
            Device& d = m->dev[(size_t)g];
            d.id = device_ids[g];
            const int c = rt_create(d.id, options, &d.ctx);
            if (c != RT_OK) throw MultiError{c, std::string("rt_multi_create: device ") + std::to_string(d.id) + ": " + rt_last_global_error()};
            MHIP(hipSetDevice(d.id));
            MHIP(hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking));
            for (int k = 0; k < 2; k++) MHIP(hipEventCreateWithFlags(&d.buffer_free[k], hipEventDisableTiming));
        }
        m->max_paths = options && options->max_paths ? options->max_paths : (1ull << 25);
        const char* mode = std::getenv("RT_REDUCE");
        // (RT_REDUCE=rccl: the collective also with one device - a one-rank ncclReduce - so that a one-GPU box can
        // exercise the loading of RCCL and the call itself)
        const bool force = mode && std::strcmp(mode, "rccl") == 0;
        m->use_rccl = (n_devices > 1 || force) && distinct && !(mode && std::strcmp(mode, "p2p") == 0);
        if (m->use_rccl) {
            std::lock_guard<std::mutex> lock(g_rccl_mutex);
            if (!g_rccl.load()) throw MultiError{RT_ERR_DEVICE, g_rccl.why + " (RT_REDUCE=p2p selects the copy-and-add reduce)"};
            std::vector<ncclComm_t> comms((size_t)n_devices);
            const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, device_ids);
            if (r != ncclSuccess) throw MultiError{RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)};
            for (int g = 0; g < n_devices; g++) m->dev[(size_t)g].comm = comms[(size_t)g];
        } else if (n_devices > 1) {
            // copy-and-add: let device 0 read its peers' memory where the hardware allows it (hipMemcpyPeerAsync works either way)
            MHIP(hipSetDevice(m->dev[0].id));
            for (int g = 1; g < n_devices; g++) {
                int can = 0;
                if (m->dev[(size_t)g].id != m->dev[0].id && hipDeviceCanAccessPeer(&can, m->dev[0].id, m->dev[(size_t)g].id) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(m->dev[(size_t)g].id, 0);
            }
            (void)hipGetLastError();
        }
        if (before >= 0) (void)hipSetDevice(before);
    });
    if (rc != RT_OK) return rc;
    *out = m.release();
    return RT_OK;
}

int rt_multi_device_count(const rt_multi* m) { return m ? (int)m->dev.size() : 0; }

int rt_multi_upload_scene(rt_multi* m, const rt_scene_desc* scene) {
    if (!m) return RT_ERR_INVALID;
    return multi_guard(m, [&] {
        // one host thread per device: the uploads (layout building included) run side by side
        std::vector<int> rcs(m->dev.size(), RT_OK);
        std::vector<std::thread> threads;
        for (size_t g = 0; g < m->dev.size(); g++)
            threads.emplace_back([&, g] { rcs[g] = rt_upload_scene(m->dev[g].ctx, scene); });
        for (auto& t : threads) t.join();
        for (size_t g = 0; g < m->dev.size(); g++)
            if (rcs[g] != RT_OK) throw MultiError{rcs[g], std::string("device ") + std::to_string(m->dev[g].id) + ": " + rt_last_error(m->dev[g].ctx)};
    });
}

int rt_multi_set_launch_timing(rt_multi* m, int on) {
    if (!m) return RT_ERR_INVALID;
    for (Device& d : m->dev) {
        const int rc = rt_set_launch_timing(d.ctx, on);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_multi_render_rows(rt_multi* m, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t spp, uint32_t max_bounces,
                         uint64_t seed, uint32_t band_rows, uint8_t* rgb8, double* sum_rgb, rt_stats* stats) {
    return rt_multi_render_rows_cb(m, width, height, j0, j1, spp, max_bounces, seed, band_rows, rgb8, sum_rgb, stats, nullptr, nullptr);
}

int rt_multi_render_rows_cb(rt_multi* m, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t spp, uint32_t max_bounces,
                            uint64_t seed, uint32_t band_rows, uint8_t* rgb8, double* sum_rgb, rt_stats* stats,
                            rt_band_callback on_band, void* user) {
    if (!m) return RT_ERR_INVALID;
    return multi_guard(m, [&] {
        if (width < 2 || height < 2 || j0 > j1 || j1 > height)
            throw MultiError{RT_ERR_INVALID, "rt_multi_render_rows: bad image geometry (width, height >= 2, j0 <= j1 <= height)"};
        if (stats) std::memset(stats, 0, sizeof(*stats));
        if (j1 == j0) return;
        const uint32_t N = (uint32_t)m->dev.size();
        const bool split_rows = spp < N;  // fewer samples than GPUs: every GPU takes rows of the band instead, nothing is reduced
        // bands: whole rows; a GPU's share of a band must fit its path budget
        const uint32_t fit = (uint32_t)std::max<uint64_t>(1, m->max_paths / width);
        uint32_t rows_per_band = band_rows ? std::min(band_rows, split_rows ? fit * N : fit) : std::min<uint32_t>(j1 - j0, split_rows ? fit * N : fit);
        rows_per_band = std::max(1u, rows_per_band);
        const uint32_t n_bands = (j1 - j0 + rows_per_band - 1) / rows_per_band;
        const size_t band_elems = (size_t)rows_per_band * width * 3;
        const bool p2p = !m->use_rccl;
        int before = -1;
        if (hipGetDevice(&before) != hipSuccess) before = -1;
        for (uint32_t g = 0; g < N; g++) ensure_buffers(m->dev[g], band_elems, g == 0, p2p);

        Rendezvous meet;
        meet.n = (int)N;
        std::vector<std::string> errors(N);
        std::vector<rt_stats> dev_stats(N);
        for (auto& s : dev_stats) std::memset(&s, 0, sizeof(s));
        std::mutex p2p_mutex;  // copy-and-add: the root's staging buffer takes one peer at a time

        auto device_thread = [&](uint32_t g) {
            Device& d = m->dev[g];
            Device& root = m->dev[0];
            bool ok = true;
            auto fail = [&](const std::string& msg) {
                if (ok) errors[g] = "device " + std::to_string(d.id) + ": " + msg;
                ok = false;
            };
            if (hipSetDevice(d.id) != hipSuccess) fail("hipSetDevice failed");
            for (uint32_t b = 0; b < n_bands; b++) {
                const uint32_t b0 = j0 + b * rows_per_band, b1 = std::min(j1, b0 + rows_per_band);
                const size_t elems = (size_t)(b1 - b0) * width * 3;
                double* acc = d.accum[b & 1];
                uint32_t r0 = b0, r1 = b1, s0 = 0, ns = spp;
                if (split_rows) {
                    uint32_t first, count;
                    split_range(b1 - b0, N, g, b0, &first, &count);
                    r0 = first;
                    r1 = first + count;
                } else {
                    split_range(spp, N, g, 0, &s0, &ns);
                }
                if (ok) {
                    try {
                        // the buffer was last used two bands ago: wait until its reduce / copies have run
                        if (b >= 2) MHIP(hipEventSynchronize(d.buffer_free[b & 1]));
                        // (not on comm_stream: the reduce of the band before may still be running there)
                        MHIP(hipMemset(acc, 0, elems * sizeof(double)));
                        rt_stats part;
                        double* dst = acc + (size_t)(r0 - b0) * width * 3;
                        if (r1 > r0 && ns > 0) {
                            const int rc = rt_render_accum_device(d.ctx, width, height, r0, r1, s0, ns, max_bounces, seed, dst, stats ? &part : nullptr);
                            if (rc != RT_OK) fail(rt_last_error(d.ctx));
                            else if (stats) add_stats(&dev_stats[g], part, true);
                        }
                    } catch (const MultiError& e) {
                        fail(e.msg);
                    }
                }
                // everyone has traced the band (or someone failed: then nobody enters the collective)
                if (!meet.meet(ok)) {
                    ok = false;
                    continue;
                }
                try {
                    if (split_rows) {
                        // no reduce: every GPU quantises its own rows and hands them to the host
                        const size_t off = (size_t)(r0 - b0) * width * 3, n = (size_t)(r1 - r0) * width * 3;
                        const size_t host_off = (size_t)(b0 - j0) * width * 3 + off;
                        if (n) {
                            if (rgb8) {
                                rtd::launch_quantize(acc + off, n / 3, spp, d.rgb8[b & 1] + off, d.comm_stream);
                                MHIP(hipGetLastError());
                                MHIP(hipMemcpyAsync(rgb8 + host_off, d.rgb8[b & 1] + off, n, hipMemcpyDeviceToHost, d.comm_stream));
                            }
                            if (sum_rgb) MHIP(hipMemcpyAsync(sum_rgb + host_off, acc + off, n * sizeof(double), hipMemcpyDeviceToHost, d.comm_stream));
                        }
                    } else {
                        if (m->use_rccl) {
                            const ncclResult_t r = g_rccl.Reduce(acc, acc, elems, ncclDouble, ncclSum, 0, d.comm, d.comm_stream);
                            if (r != ncclSuccess) throw MultiError{RT_ERR_DEVICE, std::string("ncclReduce: ") + g_rccl.GetErrorString(r)};
                        } else if (g > 0) {
                            // the peer's buffer over its own link into the root's staging buffer, added there
                            std::lock_guard<std::mutex> lock(p2p_mutex);
                            MHIP(hipSetDevice(root.id));
                            const double* src = acc;
                            if (d.id != root.id) {
                                MHIP(hipMemcpyPeerAsync(root.staging, root.id, acc, d.id, elems * sizeof(double), root.comm_stream));
                                src = root.staging;
                            }
                            hipLaunchKernelGGL(k_add_f64, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, root.comm_stream,
                                               root.accum[b & 1], src, (uint64_t)elems);
                            MHIP(hipGetLastError());
                            MHIP(hipStreamSynchronize(root.comm_stream));  // staging is free again; the peer's buffer has been read
                            MHIP(hipSetDevice(d.id));
                        }
                    }
                } catch (const MultiError& e) {
                    fail(e.msg);
                }
                // copy-and-add: the root quantises after every peer has added its part
                if (!split_rows && p2p && !meet.meet(ok)) {
                    ok = false;
                    continue;
                }
                try {
                    if (!split_rows && g == 0) {
                        const size_t host_off = (size_t)(b0 - j0) * width * 3;
                        if (rgb8) {
                            rtd::launch_quantize(acc, elems / 3, spp, d.rgb8[b & 1], d.comm_stream);
                            MHIP(hipGetLastError());
                            MHIP(hipMemcpyAsync(rgb8 + host_off, d.rgb8[b & 1], elems, hipMemcpyDeviceToHost, d.comm_stream));
                        }
                        if (sum_rgb) MHIP(hipMemcpyAsync(sum_rgb + host_off, acc, elems * sizeof(double), hipMemcpyDeviceToHost, d.comm_stream));
                    }
                    // the band after next reuses this band's buffers: it waits for this event first
                    MHIP(hipEventRecord(d.buffer_free[b & 1], d.comm_stream));
                    if (on_band && split_rows) MHIP(hipStreamSynchronize(d.comm_stream));  // every GPU holds rows of the band
                } catch (const MultiError& e) {
                    fail(e.msg);
                }
                // Finished rows to the caller.  Sample split: the root's copies of band b - 1 ran while band b was
                // traced, so its rows are handed over now, one band late, without waiting.  Row split: all meet first.
                if (on_band) {
                    if (split_rows) {
                        if (!meet.meet(ok)) {
                            ok = false;
                            continue;
                        }
                        if (g == 0) on_band(user, b0, b1);
                    } else if (g == 0 && ok && b >= 1) {
                        if (hipEventSynchronize(d.buffer_free[(b - 1) & 1]) != hipSuccess) fail("the reduce stream failed");
                        else on_band(user, b0 - rows_per_band, b0);
                    }
                }
            }
            if (hipStreamSynchronize(d.comm_stream) != hipSuccess) fail("the reduce stream failed");
            if (on_band && !split_rows && g == 0 && ok) {
                const uint32_t last0 = j0 + (n_bands - 1) * rows_per_band;
                on_band(user, last0, j1);
            }
        };

        std::vector<std::thread> threads;
        for (uint32_t g = 1; g < N; g++) threads.emplace_back(device_thread, g);
        device_thread(0);
        for (auto& t : threads) t.join();
        if (before >= 0) (void)hipSetDevice(before);
        for (uint32_t g = 0; g < N; g++)
            if (!errors[g].empty()) throw MultiError{RT_ERR_DEVICE, errors[g]};
        if (stats)
            for (uint32_t g = 0; g < N; g++) add_stats(stats, dev_stats[g], false);
    });
}

}  // extern "C"
