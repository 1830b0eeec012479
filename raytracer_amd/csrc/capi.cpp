// C ABI of librt_mi355x.so (include/rt_mi355x.h): scene files -> flattened arrays on the
// host, and the per-GPU render context that owns the HBM-resident scene, the ray queues
// and the HIP stream the wavefront kernels run on.
//
// There is deliberately no CPU fallback here: every compute entry point needs a HIP device
// and fails with RT_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "device/rt_device.h"
#include "host/flatten.hpp"
#include "host/json5.hpp"
#include "host/scene_config.hpp"
#include "tree_layout.hpp"

namespace {

using namespace rt_layout;

thread_local std::string g_error;

struct RtError {
    int code;
    std::string msg;
};

#define HIP_CHECK(expr)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw RtError{RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)};          \
    } while (0)

template <typename T>
struct DevBuf {  // owning device allocation
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void ensure(size_t count) {
        if (count <= n) return;
        release();
        HIP_CHECK(hipMalloc((void**)&p, (count ? count : 1) * sizeof(T)));
        n = count;
    }
    void upload(const std::vector<T>& v) {
        ensure(v.size() ? v.size() : 1);
        if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }
};

// ---- scene description validation: every index a kernel will follow is checked here, on
// the host, so that a malformed description is an error code and never a GPU fault.
void validate(const rt_scene_desc& d) {
    auto bad = [](const std::string& m) { throw RtError{RT_ERR_INVALID, "scene description: " + m}; };
    auto need = [&](const void* p, uint64_t n, const char* what) {
        if (n && !p) bad(std::string(what) + " is null");
    };
    need(d.nodes, d.n_nodes, "nodes");
    need(d.objects, d.n_objects, "objects");
    need(d.xforms, d.n_xforms, "xforms");
    need(d.spheres, d.n_spheres, "spheres");
    need(d.moving_spheres, d.n_moving_spheres, "moving_spheres");
    need(d.rects, d.n_rects, "rects");
    need(d.triangles, d.n_triangles, "triangles");
    need(d.planes, d.n_planes, "planes");
    need(d.media, d.n_media, "media");
    need(d.materials, d.n_materials, "materials");
    need(d.images, d.n_images, "images");
    need(d.texels, d.n_texel_bytes, "texels");
    need(d.top_list, d.n_top, "top_list");
    if (d.n_top == 0) bad("empty world");
    if (d.n_nodes > 0x7fffffffull || d.n_objects > 0x7fffffffull || d.n_triangles > 0x7fffffffull ||
        d.n_materials >= 0x7fffffffull)
        bad("too many elements");
    if (d.stack_depth < 0 || d.stack_depth > 64) bad("BVH too deep for the traversal stack (64 entries)");
    auto mat_ok = [&](int32_t m) { return m >= 0 && (uint64_t)m < d.n_materials; };
    for (uint64_t i = 0; i < d.n_materials; i++) {
        const rt_material& m = d.materials[i];
        if (m.type < RT_MAT_LAMBERTIAN || m.type > RT_MAT_CHECKERED) bad("material type");
        if (m.image >= 0 && (uint64_t)m.image >= d.n_images) bad("material image index");
        if (m.type == RT_MAT_CHECKERED) {
            // Parts must precede nothing in particular, but must not loop: follow the chain.
            if (!mat_ok(m.odd) || !mat_ok(m.even)) bad("checkered part index");
        }
    }
    {  // checkered nesting must be acyclic (the reference builds parts first, src/config.rs:143-159)
        std::vector<int> state(d.n_materials, 0);
        std::vector<std::pair<int32_t, int>> stack;
        for (uint64_t r = 0; r < d.n_materials; r++) {
            if (state[r]) continue;
            stack.push_back({(int32_t)r, 0});
            while (!stack.empty()) {
                auto& top = stack.back();
                const rt_material& m = d.materials[top.first];
                if (top.second == 0) state[top.first] = 1;
                if (m.type != RT_MAT_CHECKERED || top.second == 2) {
                    state[top.first] = 2;
                    stack.pop_back();
                    continue;
                }
                int32_t child = top.second == 0 ? m.odd : m.even;
                top.second++;
                if (state[child] == 1) bad("checkered materials form a cycle");
                if (state[child] == 0) stack.push_back({child, 0});
            }
        }
    }
    for (uint64_t i = 0; i < d.n_images; i++) {
        const rt_image& im = d.images[i];
        if (im.width <= 0 || im.height <= 0) bad("image size");
        if (im.texel_offset + (uint64_t)im.width * im.height * 3 > d.n_texel_bytes) bad("image texel range");
    }
    for (uint64_t i = 0; i < d.n_spheres; i++)
        if (!mat_ok(d.spheres[i].material)) bad("sphere material");
    for (uint64_t i = 0; i < d.n_moving_spheres; i++)
        if (!mat_ok(d.moving_spheres[i].material)) bad("moving sphere material");
    for (uint64_t i = 0; i < d.n_rects; i++) {
        if (!mat_ok(d.rects[i].material)) bad("rect material");
        if (d.rects[i].axis < 0 || d.rects[i].axis > 2) bad("rect axis");
    }
    for (uint64_t i = 0; i < d.n_triangles; i++)
        if (!mat_ok(d.triangles[i].material)) bad("triangle material");
    for (uint64_t i = 0; i < d.n_planes; i++)
        if (!mat_ok(d.planes[i].material)) bad("plane material");
    // BVH: children in range, every subtree reached once (no cycles), leaf payloads in range.
    std::vector<char> seen(d.n_nodes, 0);
    auto walk = [&](int32_t root, int leaf_kind) {
        if (root < 0 || (uint64_t)root >= d.n_nodes) bad("BVH root index");
        if (seen[root] == 2) return;  // a BLAS shared by several instances: checked already
        std::vector<int32_t> st{root};
        while (!st.empty()) {
            int32_t n = st.back();
            st.pop_back();
            if (seen[n]) bad("BVH node reached twice");
            seen[n] = 1;
            const rt_node& nd = d.nodes[n];
            if (nd.left < 0) {
                int64_t payload = -1 - (int64_t)nd.left;
                if (nd.right != leaf_kind) bad("BVH leaf kind");
                if (leaf_kind == RT_LEAF_OBJECT ? (uint64_t)payload >= d.n_objects : (uint64_t)payload >= d.n_triangles)
                    bad("BVH leaf payload");
            } else {
                if ((uint64_t)nd.left >= d.n_nodes || nd.right < 0 || (uint64_t)nd.right >= d.n_nodes)
                    bad("BVH child index");
                st.push_back(nd.left);
                st.push_back(nd.right);
            }
        }
        seen[root] = 2;
    };
    for (uint64_t i = 0; i < d.n_media; i++) {
        const rt_medium& m = d.media[i];
        if (!mat_ok(m.material)) bad("medium material");
        if (m.boundary_object < 0 || (uint64_t)m.boundary_object >= d.n_objects) bad("medium boundary index");
        if (d.objects[m.boundary_object].kind == RT_OBJ_MEDIUM) {  // a medium bounded by a medium: one level (kernels.hip, boundary_hit)
            const rt_object& inner = d.objects[m.boundary_object];
            if (inner.first < 0 || (uint64_t)inner.first >= d.n_media) bad("medium boundary index");
            const rt_medium& im = d.media[inner.first];
            if (im.boundary_object < 0 || (uint64_t)im.boundary_object >= d.n_objects) bad("medium boundary index");
            if (d.objects[im.boundary_object].kind == RT_OBJ_MEDIUM) bad("media nested more than two deep");
        }
    }
    for (uint64_t i = 0; i < d.n_objects; i++) {
        const rt_object& o = d.objects[i];
        if (o.xform < -1 || (o.xform >= 0 && (uint64_t)o.xform >= d.n_xforms)) bad("object xform index");
        uint64_t limit = 0;
        switch (o.kind) {
            case RT_OBJ_SPHERE: limit = d.n_spheres; break;
            case RT_OBJ_MOVING_SPHERE: limit = d.n_moving_spheres; break;
            case RT_OBJ_RECTS: limit = d.n_rects; break;
            case RT_OBJ_TRIANGLE: limit = d.n_triangles; break;
            case RT_OBJ_PLANE: limit = d.n_planes; break;
            case RT_OBJ_MEDIUM: limit = d.n_media; break;
            case RT_OBJ_BLAS: walk(o.first, RT_LEAF_TRIANGLE); continue;
            default: bad("object kind");
        }
        int64_t count = o.kind == RT_OBJ_RECTS ? o.count : 1;
        if (o.first < 0 || count < 1 || (uint64_t)o.first + (uint64_t)count > limit) bad("object primitive range");
    }
    bool uses_bvh = false;
    for (uint64_t i = 0; i < d.n_top; i++) {
        int32_t e = d.top_list[i];
        if (e == RT_TOP_BVH) uses_bvh = true;
        else if (e < 0 || (uint64_t)e >= d.n_objects) bad("top_list entry");
    }
    if (uses_bvh) walk(d.tlas_root, RT_LEAF_OBJECT);
}

template <typename T> void cvt3(T* o, const double* s) { o[0] = (T)s[0]; o[1] = (T)s[1]; o[2] = (T)s[2]; }

// ---- work buffers of one lane.  Raw device pointers: either views of hipMalloc'ed buffers that grow on
// demand (OwnedWork) or pieces of the library's static device array (carve_work: the allocation-free entry point).
template <typename T>
struct Work {
    T* qslab[2] = {nullptr, nullptr};  // two ray queues, 10 fields of `capacity` entries each
    uint32_t* qpid[2] = {nullptr, nullptr};
    T* hslab = nullptr;  // hit records, 6 fields
    uint32_t* hmat = nullptr;
    T* sample_rgb = nullptr;  // per-path radiance of a batch
    uint32_t* tile_counts[2] = {nullptr, nullptr};
    uint32_t* tile_offsets = nullptr;
    uint32_t* counts = nullptr;  // [0] live paths after the last scan, [1] k_extend_refill's cursor, [2] deepest bounce reached inside a tail launch, [3] the tail launch's cursor
    unsigned long long* ray_total = nullptr;  // [0] closest-hit queries counted by the scans, [1] those issued inside a tail launch beyond its first bounce, [2] paths that entered tail launches
    int32_t* hids = nullptr;  // rt_trace only
    T* tminmax = nullptr;     // rt_trace only
    size_t capacity = 0;         // paths the queues hold
    size_t sample_capacity = 0;  // paths sample_rgb holds
};

template <typename T>
struct OwnedWork {
    DevBuf<T> qslab[2], hslab, sample_rgb, tminmax;
    DevBuf<uint32_t> qpid[2], hmat;
    DevBuf<int32_t> hids;
    DevBuf<uint32_t> counts, tile_counts[2], tile_offsets;
    DevBuf<unsigned long long> ray_total;
    void ensure(Work<T>& w, size_t n, size_t n_samples, bool with_ids) {
        if (!counts.p) {
            counts.ensure(4);
            ray_total.ensure(3);
        }
        if (n > w.capacity) {
            for (int k = 0; k < 2; k++) {
                qslab[k].ensure(n * 10);
                qpid[k].ensure(n);
            }
            hslab.ensure(n * 6);
            hmat.ensure(n);
            for (int k = 0; k < 2; k++) tile_counts[k].ensure((n + RT_TILE - 1) / RT_TILE);
            tile_offsets.ensure((n + RT_TILE - 1) / RT_TILE + 1);
            w.capacity = n;
        }
        if (n_samples > w.sample_capacity) {
            sample_rgb.ensure(n_samples * 3);
            w.sample_capacity = n_samples;
        }
        if (with_ids) {
            hids.ensure(w.capacity * 4);     // owner, sub, world_index, inst
            tminmax.ensure(w.capacity * 3);  // tmin, tmax, second_t
        }
        for (int k = 0; k < 2; k++) {
            w.qslab[k] = qslab[k].p;
            w.qpid[k] = qpid[k].p;
            w.tile_counts[k] = tile_counts[k].p;
        }
        w.hslab = hslab.p; w.hmat = hmat.p; w.sample_rgb = sample_rgb.p; w.tile_offsets = tile_offsets.p;
        w.counts = counts.p; w.ray_total = ray_total.p; w.hids = hids.p; w.tminmax = tminmax.p;
    }
};

// Paths in flight of the allocation-free entry point, and the bytes of its buffers (f64: 244 B per path).
#define RT_DEFAULT_LANES 2
constexpr size_t work_bytes(size_t elem, size_t paths) {
    const size_t tiles = (paths + RT_TILE - 1) / RT_TILE;
    return 2 * (paths * 10 * elem + paths * 4) + paths * 6 * elem + paths * 4 + paths * 3 * elem  // queues, hits, samples
           + (3 * tiles + 1) * 4 + 4 * 4 + 3 * 8                                                  // tile counts / offsets, counters
           + 16 * 256;                                                                            // every piece starts on a 256-byte boundary
}
template <typename T>
Work<T> carve_work(unsigned char* base, size_t bytes, size_t paths) {
    Work<T> w;
    size_t off = 0;
    auto take = [&](size_t n) {
        unsigned char* p = base + off;
        off += (n + 255) & ~(size_t)255;
        return p;
    };
    const size_t tiles = (paths + RT_TILE - 1) / RT_TILE;
    for (int k = 0; k < 2; k++) {
        w.qslab[k] = (T*)take(paths * 10 * sizeof(T));
        w.qpid[k] = (uint32_t*)take(paths * 4);
        w.tile_counts[k] = (uint32_t*)take(tiles * 4);
    }
    w.hslab = (T*)take(paths * 6 * sizeof(T));
    w.hmat = (uint32_t*)take(paths * 4);
    w.sample_rgb = (T*)take(paths * 3 * sizeof(T));
    w.tile_offsets = (uint32_t*)take((tiles + 1) * 4);
    w.counts = (uint32_t*)take(4 * 4);
    w.ray_total = (unsigned long long*)take(3 * 8);
    if (off > bytes) throw RtError{RT_ERR_INVALID, "internal: the static work area is too small"};
    w.capacity = paths;
    w.sample_capacity = paths;
    return w;
}

// One lane: a stream, its work buffers and the pinned words through which the kernels tell the host how
// many paths are alive (k_scan_tiles stores the count of every bounce there; the host never waits for the
// stream to learn it).
// Two rings of per-bounce live counts per lane, indexed by depth % PEEK_SLOTS; a lane's batches alternate between them.
// A batch can be finished by the host while its last scans are still queued on the GPU (the tail launch needs no
// count): their late stores must not land in the ring the lane's NEXT batch is reading, which has re-marked its slots
// unknown meanwhile - a stale count there would be taken for the new batch's own.  By the time the batch after next
// reuses a ring, the host has read a count of the batch between (stored by a scan that ran after all of this one's).
constexpr uint32_t PEEK_SLOTS = 128;
constexpr uint32_t PEEK_UNKNOWN = 0xffffffffu;
template <typename T>
struct Lane {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool fixed = false;  // buffers are pieces of the static device array: never grown
    Work<T> w;
    OwnedWork<T> owned;
    uint32_t* peek = nullptr;              // pinned, device-visible: PEEK_SLOTS live counts
    unsigned long long* readback = nullptr;  // pinned: ray totals [0..2], deepest [3]
    hipEvent_t ev_acc = nullptr;   // recorded after a batch's k_accumulate (the next batch's accumulate waits for it)
    hipEvent_t ev_join = nullptr;  // recorded when the lane has enqueued all its work of a call
    std::vector<hipEvent_t> events;  // launch timing
    explicit Lane(hipStream_t st) {
        if (st) stream = st;
        else {
            HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            own_stream = true;
        }
        HIP_CHECK(hipHostMalloc((void**)&peek, 2 * PEEK_SLOTS * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_CHECK(hipHostMalloc((void**)&readback, 4 * sizeof(unsigned long long), hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&ev_acc, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    Lane(const Lane&) = delete;
    Lane& operator=(const Lane&) = delete;
    ~Lane() {
        if (peek) (void)hipHostFree(peek);
        if (readback) (void)hipHostFree(readback);
        if (ev_acc) (void)hipEventDestroy(ev_acc);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (auto e : events) (void)hipEventDestroy(e);
        if (own_stream) (void)hipStreamDestroy(stream);
    }
    void ensure(size_t n, size_t n_samples, bool with_ids) {
        if (fixed) {
            if (n > w.capacity || n_samples > w.sample_capacity || with_ids)
                throw RtError{RT_ERR_INVALID, "internal: request beyond the static work area"};
            return;
        }
        owned.ensure(w, n, n_samples, with_ids);
    }
    uint32_t* peek_dev() const {  // the pinned words as the kernels address them
        void* d = nullptr;
        HIP_CHECK(hipHostGetDevicePointer(&d, peek, 0));
        return (uint32_t*)d;
    }
    uint32_t ring = 0;  // which of the two rings the lane's current batch uses
    uint32_t peek_slot(uint32_t depth) const { return ring * PEEK_SLOTS + depth % PEEK_SLOTS; }
    uint32_t peek_value(uint32_t depth) const { return __atomic_load_n(&peek[peek_slot(depth)], __ATOMIC_ACQUIRE); }
    hipEvent_t event(size_t k) {
        while (events.size() <= k) {
            hipEvent_t e;
            HIP_CHECK(hipEventCreate(&e));
            events.push_back(e);
        }
        return events[k];
    }
};

// Experiment switches (A/B runs and tests; the results never depend on them - tests/test_gpu_parity.py renders every scene
// under each).  They are read from the environment ONCE, when a context is created: no entry point looks at the
// environment afterwards.  INTEGRATION.md lists them.
struct Tuning {
    int stack_cap = 0;          // RT_STACK_CAP: LDS stack entries a four-wide walk may use (0: the kernel's own)
    bool bvh4 = true;           // RT_BVH4=0: binary walks only
    int sah = 1;                // RT_SAH=0: the reference's grouping folded two levels into one; 2: SAH for meshes only
    bool ordered = true;        // RT_ORDERED=0: the reference's fixed left-first order
    int coop = 16;              // RT_COOP=n: lane groups answer a tail wave's queries while it has at most n live paths (0: never)
    bool all_features = false;  // RT_FEATURES=all: the general four-wide kernel whatever the scene holds
    long sweep_max = 12;        // RT_SWEEP_MAX: largest world (leaves) that is swept instead of walked
    int leaf_frac = 5;          // RT_LEAF_FRAC=d: the wave runs its exact triangle tests when d sixteenths of the lanes still in the mesh walk hold a leaf (0: RT_LEAF_VOTE alone)
    int obj_frac = 12;          // RT_OBJ_FRAC=d: the same rule for the leaves of the world tree in closest_hit (0: an object is tested as soon as a lane holds it)
    int leaf_vote = 64;         // RT_LEAF_VOTE=n: ... or n lanes do (1: every leaf is tested at once)
    int refill = -1;            // RT_REFILL=n: k_extend_refill's threshold (-1: by scene)
    size_t grid_mult = 0;       // RT_GRID_MULT: persistent blocks per CU (0: 32, or 8 per lane while two lanes run)
    int tail_blocks = 0;        // RT_TAIL_BLOCKS: resident blocks per CU of the tail launch (0: by kind of walk)
    uint64_t lane_min_paths = (uint64_t)8 << 20;  // RT_LANE_MIN_PATHS: paths per lane from which a second lane runs
    bool one_lane = false;      // RT_ONE_LANE=1: one lane at a time, batches cut as usual (profiles)
    int fuse = -1;              // RT_FUSE=0 / 1: fused per-bounce kernel off / on (-1: lean sweeps only)
    long tail = -1;             // RT_TAIL=n: live paths at or below which the tail launch takes over (-1: by scene, 0: never)
    int pixel_blocks = -1;      // RT_PIXEL_BLOCKS=0 rows, 1 blocks of 8 x 8 pixels, 2 forced pixels x samples (-1: by scene)
    uint32_t runahead = 2;      // RT_RUNAHEAD: bounces enqueued beyond the last live count the host has seen
    bool debug = false;         // RT_DEBUG: counts and layout choices on stderr
    bool fuse_raygen = true;    // RT_FUSE_RAYGEN=0: k_raygen writes the camera rays of a batch to its queue also where the first bounce could make them
    long gpu_build = -1;        // RT_GPU_BUILD=n: meshes of at least n triangles get their four-wide tree built on the GPU (0: never; -1: default)
    static Tuning from_environment() {
        Tuning t;
        auto num = [](const char* name, long fallback) {
            const char* e = std::getenv(name);
            return e && e[0] ? std::strtol(e, nullptr, 10) : fallback;
        };
        auto is = [](const char* name, char c) {
            const char* e = std::getenv(name);
            return e && e[0] == c;
        };
        t.stack_cap = (int)num("RT_STACK_CAP", 0);
        t.bvh4 = !is("RT_BVH4", '0');
        t.sah = is("RT_SAH", '0') ? 0 : is("RT_SAH", '2') ? 2 : 1;
        t.ordered = !is("RT_ORDERED", '0');
        t.coop = (int)std::max(0l, std::min(32l, num("RT_COOP", 16)));
        t.all_features = is("RT_FEATURES", 'a');
        t.sweep_max = num("RT_SWEEP_MAX", 12);
        t.refill = (int)num("RT_REFILL", -1);
        t.leaf_vote = (int)std::max(1l, std::min(64l, num("RT_LEAF_VOTE", 64)));
        t.obj_frac = (int)std::max(0l, std::min(16l, num("RT_OBJ_FRAC", 12)));
        t.leaf_frac = (int)std::max(0l, std::min(16l, num("RT_LEAF_FRAC", 5)));
        const long gm = num("RT_GRID_MULT", 0);
        t.grid_mult = gm > 0 && gm <= 65536 ? (size_t)gm : 0;
        t.tail_blocks = (int)num("RT_TAIL_BLOCKS", 0);
        if (const char* e = std::getenv("RT_LANE_MIN_PATHS")) t.lane_min_paths = (uint64_t)std::strtoull(e, nullptr, 10);
        t.one_lane = is("RT_ONE_LANE", '1');
        if (const char* e = std::getenv("RT_FUSE")) t.fuse = e[0] != '0' ? 1 : 0;
        t.tail = num("RT_TAIL", -1);
        if (const char* e = std::getenv("RT_PIXEL_BLOCKS")) t.pixel_blocks = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
        t.runahead = (uint32_t)std::max(1l, std::min(64l, num("RT_RUNAHEAD", 2)));
        t.debug = std::getenv("RT_DEBUG") != nullptr;
        t.fuse_raygen = !is("RT_FUSE_RAYGEN", '0');
        t.gpu_build = num("RT_GPU_BUILD", -1);
        return t;
    }
};

// Meshes of at least this many triangles get their four-wide tree from the GPU builder unless RT_GPU_BUILD says otherwise
// (0: never).  Measured in DESIGN.md section 6 ("GPU tree build").
#ifndef RT_GPU_BUILD_DEFAULT_MIN
#define RT_GPU_BUILD_DEFAULT_MIN 2000000
#endif

struct EngineBase {
    virtual ~EngineBase() {}
    virtual void upload(const rt_scene_desc& d) = 0;
    virtual void trace(const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) = 0;
    virtual void primary_rays(uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t sample, uint64_t seed, rt_ray* out) = 0;
    virtual void render_accum(uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t s0, uint32_t ns,
                              uint32_t max_bounces, uint64_t seed, double* accum_dev, rt_stats* stats, bool fixed) = 0;
    virtual void set_launch_timing(bool on) = 0;
    virtual void upload_stats(rt_upload_stats* out) = 0;
};

template <typename T>
struct Engine : EngineBase {
    hipStream_t stream;
    int num_cus;
    uint64_t max_paths;
    bool have_scene = false;
    int stack_depth = 16;
    int stack_need4 = 0;  // stack entries a four-wide mesh walk can need (0: no four-wide trees)
    bool sah_trees = false;  // the four-wide trees were regrouped by SAH (else: the reference's grouping, folded)
    bool nested_media = false;  // some medium's boundary is a medium: the general four-wide kernel runs the scene
    rtd::DScene<T> S;
    DevBuf<rtd::DNode2> nodes2;
    size_t n_nodes2 = 0;
    DevBuf<rtd::DNode4> nodes4;
    DevBuf<rtd::DNode<T>> nodes_own;  // only kept when the scene is small enough for LDS staging
    DevBuf<rtd::DObject> objects;
    DevBuf<rtd::DSky<T>> sky;
    DevBuf<rtd::DBlock<T>> blocks;
    DevBuf<rtd::DXform<T>> xforms;
    DevBuf<rtd::DSphere<T>> spheres;
    DevBuf<rtd::DMovingSphere<T>> moving_spheres;
    DevBuf<rtd::DRect<T>> rects;
    DevBuf<rtd::DTriangle<T>> triangles;
    DevBuf<rtd::DPlane<T>> planes;
    DevBuf<rtd::DMedium<T>> media;
    DevBuf<rtd::DMaterial<T>> materials;
    DevBuf<rtd::DImage> images;
    DevBuf<uint8_t> texels;
    DevBuf<int32_t> top_list, sweep;
    DevBuf<int4> sweep_items;
    DevBuf<unsigned char> small_blob;
    rtd::DSmallScene small;
    // work buffers: one lane = one stream + one set of queues (render_accum deals its batches over the lanes)
    std::vector<std::unique_ptr<Lane<T>>> lanes;
    std::unique_ptr<Lane<T>> fixed_lane;  // the lane of rt_render_accum_device_noalloc: buffers in the library's static device array
    DevBuf<double> io_f64;
    DevBuf<rt_hit> io_hits;
    bool time_launches = false;  // rt_set_launch_timing: HIP events around every launch (each costs the stream ~5 us)
    uint32_t tail_threshold_fused = 1048576u, tail_threshold_tree = 1048576u;  // live paths at or below which the tail launch takes over
    int tail_blocks_per_cu = 2;  // blocks of the tail launch that stay resident per CU (k_bounce<.., CAN_TAIL = true>: RT_TAIL_WAVES waves per SIMD)
    rt_upload_stats up_stats;
    const Tuning tune = Tuning::from_environment();  // (once per context, here: rt_create)
    void set_launch_timing(bool on) override { time_launches = on; }
    void upload_stats(rt_upload_stats* out) override { *out = up_stats; }

    Engine(hipStream_t st, int cus, uint64_t maxp, int n_lanes, void* fixed_base, size_t fixed_bytes)
        : stream(st), num_cus(cus), max_paths(maxp) {
        std::memset(&S, 0, sizeof(S));
        std::memset(&up_stats, 0, sizeof(up_stats));
        for (int k = 0; k < n_lanes; k++) lanes.emplace_back(new Lane<T>(k == 0 ? st : nullptr));
        if (fixed_base) {
            fixed_lane.reset(new Lane<T>(st));
            fixed_lane->w = carve_work<T>((unsigned char*)fixed_base, fixed_bytes, RT_NOALLOC_MAX_PATHS);
            fixed_lane->fixed = true;
        }
    }
    ~Engine() override {}

    void upload(const rt_scene_desc& d) override {
        using clock = std::chrono::steady_clock;
        auto ms_since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
        const clock::time_point t_begin = clock::now();
        std::memset(&up_stats, 0, sizeof(up_stats));
        validate(d);
        up_stats.ms_validate = ms_since(t_begin);
        const clock::time_point t_build = clock::now();
        have_scene = false;
        // Traversal layout: the reference-style tree (rt_node: every node holds its own f64 box) is
        // re-laid as DNode2 records that hold the boxes of both children (see rt_device.h).
        std::vector<int32_t> root2(d.n_nodes, -1);  // rt_node root index -> DNode2 super-root index
        LeafRanks ranks;
        ranks.object.assign(d.n_objects, 0);
        ranks.triangle.assign(d.n_triangles, 0);
        // (built by a task of its own while this thread regroups the four-wide trees below; uploaded after both)
        std::vector<rtd::DNode2> v2;
        std::future<void> nodes2_task = std::async(std::launch::async, [&] {
            v2.reserve(d.n_nodes / 2 + 8);
            const float pad = box_pad(d);
            if (d.tlas_root >= 0) root2[d.tlas_root] = build_nodes2(d, d.tlas_root, pad, v2, ranks);
            for (uint64_t i = 0; i < d.n_objects; i++)
                if (d.objects[i].kind == RT_OBJ_BLAS && root2[d.objects[i].first] < 0)
                    root2[d.objects[i].first] = build_nodes2(d, d.objects[i].first, pad, v2, ranks);
        });
        struct Join {  // (an exception below must not leave the task running on locals that are gone)
            std::future<void>& f;
            ~Join() {
                if (f.valid()) f.wait();
            }
        } join_nodes2{nodes2_task};
        // Four-wide mesh trees (RT_BVH4=0 in the environment keeps the binary walk for A/B runs).
        std::vector<int32_t> root4(d.n_nodes, -1);
        S.nodes4 = nullptr;
        // They are regrouped by SAH (RT_SAH=0: the reference's grouping, folded two levels into one); a
        // walk keeps up to 3 siblings per level on its stack, and a tree whose worst case would not fit
        // the stack (LDS class + overflow array) is not used: SAH -> folded reference tree -> binary walk.
        stack_depth = d.stack_depth;
        S.stack_cap = tune.stack_cap;  // (tests)
        sah_trees = false;
        {
            const int lds_class = stack_depth <= 16 ? 16 : stack_depth <= 32 ? 32 : 64;
            const int lds_part = S.stack_cap > 0 && S.stack_cap < lds_class ? S.stack_cap : lds_class;
            const float pad = box_pad(d);
            S.tlas_root4 = -1;
            stack_need4 = 0;
            // Meshes of at least `gpu_build_min` triangles get their four-wide tree from the GPU builder (bvh_build.hip:
            // Morton order, radix tree, fold, quantisation) instead of the host's binned-SAH regrouping; the host still
            // says which leaves the reference can reach (collect_leaves) and pads their boxes.  A tree that comes out too
            // deep for the walk's stack, or that the builder cannot store, is built by the host after all.
            struct GpuTree {
                int32_t first;  // rt_node root of the mesh
                rtd::DNode4* nodes;
                uint32_t n_nodes;
                int levels;
            };
            std::vector<GpuTree> gpu_trees;
            struct FreeTrees {
                std::vector<GpuTree>& t;
                ~FreeTrees() {
                    for (GpuTree& g : t)
                        if (g.nodes) (void)hipFree(g.nodes);
                }
            } free_trees{gpu_trees};
            const long gpu_build_min = tune.gpu_build >= 0 ? tune.gpu_build : RT_GPU_BUILD_DEFAULT_MIN;
            const clock::time_point t_gpu = clock::now();
            // (a world of one boxed object has no world tree - BVHNode::from_vec makes a plain list of it, src/hitting.rs:165-170 -
            // but a mesh in it is walked through its four-wide tree all the same)
            bool any_mesh = false;
            for (uint64_t i = 0; i < d.n_objects; i++)
                if (d.objects[i].kind == RT_OBJ_BLAS) any_mesh = true;
            const bool want4 = tune.bvh4 && (d.tlas_root >= 0 || any_mesh);
            if (want4 && gpu_build_min > 0) {
                std::vector<char> tried(d.n_nodes, 0);
                for (uint64_t i = 0; i < d.n_objects; i++) {
                    if (d.objects[i].kind != RT_OBJ_BLAS || tried[(size_t)d.objects[i].first]) continue;
                    const int32_t first = d.objects[i].first;
                    tried[(size_t)first] = 1;
                    std::vector<float> lo, hi;
                    std::vector<int32_t> refs;
                    if ((long)d.n_triangles < gpu_build_min) continue;  // (no mesh of the scene can be that large)
                    lo.reserve(3 * (size_t)d.n_triangles);
                    hi.reserve(3 * (size_t)d.n_triangles);
                    refs.reserve((size_t)d.n_triangles);
                    if (!collect_leaf_arrays(d, first, pad, lo, hi, refs) || (long)refs.size() < gpu_build_min || refs.empty()) continue;
                    GpuTree g{first, nullptr, 0, 0};
                    const int rc = rtd::gpu_build_tree4(lo.data(), hi.data(), refs.data(), (uint32_t)refs.size(), 0, stream, &g.nodes, &g.n_nodes, &g.levels);
                    if (rc > 0) throw RtError{RT_ERR_DEVICE, std::string("GPU tree build: ") + hipGetErrorString((hipError_t)rc)};
                    if (rc == 0 && 3 * g.levels + 8 <= lds_part + RT_STACK_OVERFLOW) gpu_trees.push_back(g);
                    else if (g.nodes) (void)hipFree(g.nodes);
                    if (tune.debug)
                        std::fprintf(stderr, "  GPU tree build: %zu leaves -> %u nodes, %d levels%s\n", refs.size(), g.n_nodes, g.levels,
                                     rc == 0 && 3 * g.levels + 8 <= lds_part + RT_STACK_OVERFLOW ? "" : " (not used: too deep or not storable)");
                }
            }
            up_stats.ms_gpu_build = gpu_trees.empty() ? 0.0 : ms_since(t_gpu);
            up_stats.gpu_built_trees = (uint32_t)gpu_trees.size();
            auto gpu_tree_of = [&](int32_t first) -> const GpuTree* {
                for (const GpuTree& g : gpu_trees)
                    if (g.first == first) return &g;
                return nullptr;
            };
            for (int use_sah = tune.sah == 0 ? 0 : 1; use_sah >= 0; use_sah--) {
                if (!want4) break;
                std::vector<rtd::WideNode4> v;
                int need = 0, need_world = 0;
                auto build = [&](int32_t root, int* need_out) {
                    int32_t r = -1;
                    if (use_sah && build_sah4(d, root, pad, v, &r, need_out)) return r;
                    return build_nodes4(d, root, pad, v, need_out);
                };
                std::fill(root4.begin(), root4.end(), -1);
                {
                    const int meshes_too = use_sah;
                    if (tune.sah == 2) use_sah = 0;  // RT_SAH=2: meshes only
                    S.tlas_root4 = d.tlas_root >= 0 ? build(d.tlas_root, &need_world) : -1;
                    use_sah = meshes_too;
                }
                for (uint64_t i = 0; i < d.n_objects; i++)
                    if (d.objects[i].kind == RT_OBJ_BLAS && root4[d.objects[i].first] < 0) {
                        int need_here = 0;
                        if (const GpuTree* g = gpu_tree_of(d.objects[i].first)) {
                            root4[d.objects[i].first] = 0;  // (its place follows the host's nodes: set below)
                            need_here = 3 * g->levels;
                        } else {
                            root4[d.objects[i].first] = build(d.objects[i].first, &need_here);
                        }
                        need = std::max(need, need_here);
                    }
                stack_need4 = need + need_world + 3;
                std::vector<rtd::DNode4> packed;
                if (stack_need4 <= lds_part + RT_STACK_OVERFLOW && quantize_nodes4(v, packed)) {
                    if (tune.debug && S.tlas_root4 >= 0 && v[(size_t)S.tlas_root4].child[0] >= 0) {
                        const rtd::WideNode4& w = v[(size_t)v[(size_t)S.tlas_root4].child[0]];
                        for (int k = 0; k < 4; k++)
                            std::fprintf(stderr, "  world slot %d: child %d  [%g %g %g] - [%g %g %g]\n", k, w.child[k], w.lo[0][k], w.lo[1][k],
                                         w.lo[2][k], w.hi[0][k], w.hi[1][k], w.hi[2][k]);
                    }
                    size_t total = packed.size();
                    for (const GpuTree& g : gpu_trees) total += g.n_nodes;
                    if (total > 0x7fffffffull) throw RtError{RT_ERR_INVALID, "scene description: too many four-wide nodes"};
                    nodes4.release();
                    nodes4.ensure(total);
                    if (!packed.empty()) HIP_CHECK(hipMemcpy(nodes4.p, packed.data(), packed.size() * sizeof(rtd::DNode4), hipMemcpyHostToDevice));
                    size_t at = packed.size();
                    for (const GpuTree& g : gpu_trees) {
                        rtd::gpu_place_tree4(g.nodes, g.n_nodes, (int32_t)at, nodes4.p + at, stream);
                        HIP_CHECK(hipGetLastError());
                        root4[(size_t)g.first] = (int32_t)at;
                        at += g.n_nodes;
                    }
                    HIP_CHECK(hipStreamSynchronize(stream));
                    S.nodes4 = nodes4.p;
                    sah_trees = use_sah != 0;
                    break;
                }
            }
        }
        nodes2_task.get();
        nodes2.upload(v2);
        n_nodes2 = v2.size();
        std::vector<rtd::DNode2>().swap(v2);
        up_stats.ms_build = ms_since(t_build);  // (includes the copies of the two node arrays)
        const clock::time_point t_copy = clock::now();
        {
            std::vector<rtd::DObject> v(d.n_objects);
            std::vector<rtd::DBlock<T>> block_corners;
            for (size_t i = 0; i < v.size(); i++) {
                const rt_object& o = d.objects[i];
                // A Block (Block::new: six rects on the planes of two corners, axes z, z, y, y, x, x) whose
                // sides really are the faces of one box is tested from the two corners alone: 48 bytes
                // fetched per test instead of six 64-byte rect records (the walks are bound by fetches).
                int32_t block6 = 0;
                if (o.kind == RT_OBJ_RECTS && o.count == 6) {
                    static const int32_t axes[6] = {2, 2, 1, 1, 0, 0};
                    const rt_rect* q = d.rects + o.first;
                    bool box = true;
                    for (int k = 0; k < 6; k++)
                        if (q[k].axis != axes[k]) box = false;
                    if (box) {
                        const double lo[3] = {q[4].k, q[2].k, q[0].k}, hi[3] = {q[5].k, q[3].k, q[1].k};
                        for (int k = 0; k < 6 && box; k++) {
                            const int ia = axes[k] == 0 ? 1 : 0, ib = axes[k] == 2 ? 1 : 2;  // as in rect_hit
                            box = q[k].a0 == lo[ia] && q[k].a1 == hi[ia] && q[k].b0 == lo[ib] && q[k].b1 == hi[ib];
                        }
                        if (box) {
                            rtd::DBlock<T> b;
                            cvt3(b.lo, lo);
                            cvt3(b.hi, hi);
                            block_corners.push_back(b);
                            block6 = (int32_t)block_corners.size();
                        }
                    }
                }
                v[i] = rtd::DObject{o.kind, o.xform, o.kind == RT_OBJ_BLAS ? root2[o.first] : o.first, o.count, o.world_index,
                                    ranks.object[i], block6, o.kind == RT_OBJ_BLAS ? root4[o.first] : -1};
            }
            objects.upload(v);
            blocks.upload(block_corners);
        }
        {
            std::vector<rtd::DXform<T>> v(d.n_xforms);
            for (size_t i = 0; i < v.size(); i++) {
                for (int k = 0; k < 9; k++) v[i].m[k] = (T)d.xforms[i].m[k];
                cvt3(v[i].b, d.xforms[i].b);
            }
            xforms.upload(v);
        }
        {
            std::vector<rtd::DSphere<T>> v(d.n_spheres);
            for (size_t i = 0; i < v.size(); i++) {
                cvt3(v[i].centre, d.spheres[i].centre);
                v[i].radius = (T)d.spheres[i].radius;
                v[i].material = d.spheres[i].material;
                v[i].owner = d.spheres[i].owner;
            }
            spheres.upload(v);
        }
        {
            std::vector<rtd::DMovingSphere<T>> v(d.n_moving_spheres);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_moving_sphere& s = d.moving_spheres[i];
                cvt3(v[i].centre0, s.centre0);
                cvt3(v[i].centre1, s.centre1);
                v[i].time0 = (T)s.time0; v[i].time1 = (T)s.time1; v[i].radius = (T)s.radius;
                v[i].material = s.material;
                v[i].owner = s.owner;
            }
            moving_spheres.upload(v);
        }
        {
            std::vector<rtd::DRect<T>> v(d.n_rects);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_rect& r = d.rects[i];
                v[i].a0 = (T)r.a0; v[i].a1 = (T)r.a1; v[i].b0 = (T)r.b0; v[i].b1 = (T)r.b1; v[i].k = (T)r.k;
                v[i].axis = r.axis; v[i].facing_positive = r.facing_positive;
                v[i].material = r.material; v[i].owner = r.owner; v[i].sub = r.sub; v[i].pad = 0;
            }
            rects.upload(v);
        }
        {
            std::vector<rtd::DTriangle<T>> v(d.n_triangles);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_triangle& t = d.triangles[i];
                cvt3(v[i].point, t.point); cvt3(v[i].vec1, t.vec1); cvt3(v[i].vec2, t.vec2); cvt3(v[i].normal, t.normal);
                v[i].material = t.material; v[i].owner = t.owner; v[i].sub = t.sub; v[i].dfs_rank = ranks.triangle[i];
            }
            triangles.upload(v);
        }
        {
            std::vector<rtd::DPlane<T>> v(d.n_planes);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_plane& p = d.planes[i];
                cvt3(v[i].point, p.point); cvt3(v[i].vec1, p.vec1); cvt3(v[i].vec2, p.vec2); cvt3(v[i].normal, p.normal);
                v[i].uv_repeat = (T)p.uv_repeat;
                v[i].material = p.material; v[i].owner = p.owner;
            }
            planes.upload(v);
        }
        {
            std::vector<rtd::DMedium<T>> v(d.n_media);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_medium& m = d.media[i];
                v[i].neg_inv_density = (T)m.neg_inv_density;
                v[i].boundary_object = m.boundary_object; v[i].material = m.material; v[i].owner = m.owner;
                v[i].rng_key = m.rng_key & 0x7fffffffu;
            }
            media.upload(v);
        }
        {
            std::vector<rtd::DMaterial<T>> v(d.n_materials);
            for (size_t i = 0; i < v.size(); i++) {
                const rt_material& m = d.materials[i];
                cvt3(v[i].colour, m.colour);
                v[i].param = (T)m.param;
                v[i].type = m.type; v[i].image = m.image; v[i].odd = m.odd; v[i].even = m.even;
            }
            bool any_checker_emits = false;
            // Checkered: `image` (unused by that type) says whether some part of it, through nested checkers,
            // is a light; emitted() of any other checker is black whatever part (u, v) selects.
            for (size_t i = 0; i < v.size(); i++) {
                if (d.materials[i].type != RT_MAT_CHECKERED) continue;
                bool emits = false;
                std::vector<int32_t> todo{(int32_t)i};
                std::vector<char> seen(v.size(), 0);
                while (!todo.empty() && !emits) {
                    const int32_t k = todo.back();
                    todo.pop_back();
                    if (seen[(size_t)k]) continue;
                    seen[(size_t)k] = 1;
                    const rt_material& mk = d.materials[k];
                    if (mk.type == RT_MAT_DIFFUSE_LIGHT) emits = true;
                    if (mk.type == RT_MAT_CHECKERED) {
                        todo.push_back(mk.odd);
                        todo.push_back(mk.even);
                    }
                }
                v[i].image = emits ? 1 : 0;
                if (emits) any_checker_emits = true;
            }
            materials.upload(v);
            // Light and sky are where paths end (DiffuseLight::scatter is None, src/materials.rs:126-128); only a
            // checker may emit through one part and scatter through another (the PI asymmetry, :166-183).
            S.emit_once = any_checker_emits ? 0 : 1;
        }
        {
            std::vector<rtd::DImage> v(d.n_images);
            for (size_t i = 0; i < v.size(); i++) v[i] = rtd::DImage{d.images[i].width, d.images[i].height, d.images[i].texel_offset};
            images.upload(v);
        }
        texels.upload(std::vector<uint8_t>(d.texels, d.texels + d.n_texel_bytes));
        top_list.upload(std::vector<int32_t>(d.top_list, d.top_list + d.n_top));
        S.nodes2 = nodes2.p; S.objects = objects.p; S.blocks = blocks.p; S.xforms = xforms.p; S.spheres = spheres.p;
        S.moving_spheres = moving_spheres.p; S.rects = rects.p; S.triangles = triangles.p; S.planes = planes.p;
        S.media = media.p; S.materials = materials.p; S.images = images.p; S.texels = texels.p;
        S.top_list = top_list.p;
        S.n_top = (int32_t)d.n_top;
        S.tlas_root = d.tlas_root >= 0 ? root2[d.tlas_root] : -1;
        S.tlas_root_own = d.tlas_root;
        S.nodes = nullptr;
        const rt_camera& c = d.camera;
        cvt3(S.camera.origin, c.origin); cvt3(S.camera.lower_left_corner, c.lower_left_corner);
        cvt3(S.camera.horizontal, c.horizontal); cvt3(S.camera.vertical, c.vertical);
        cvt3(S.camera.u, c.u); cvt3(S.camera.v, c.v);
        S.camera.lens_radius = (T)c.lens_radius; S.camera.start_time = (T)c.start_time; S.camera.end_time = (T)c.end_time;
        {
            rtd::DSky<T> sky_host;
            std::memset(&sky_host, 0, sizeof(sky_host));
            sky_host.gradient = d.sky.gradient;
            cvt3(sky_host.colour, d.sky.colour); cvt3(sky_host.unit_dir, d.sky.unit_dir);
            cvt3(sky_host.colour0, d.sky.colour0); cvt3(sky_host.colour1, d.sky.colour1);
            sky.upload(std::vector<rtd::DSky<T>>(1, sky_host));
            S.sky = sky.p;
        }
        // bounding sphere of the world BVH (every box of every tree lies inside it) for far_shift
        S.far_origin = INFINITY;
        S.bound_centre[0] = S.bound_centre[1] = S.bound_centre[2] = (T)0;
        S.bound_radius = (T)0;
        if (d.tlas_root >= 0) {
            const rt_node& root = d.nodes[d.tlas_root];
            double r2 = 0;
            bool finite = true;
            for (int k = 0; k < 3; k++) {
                if (!std::isfinite(root.lo[k]) || !std::isfinite(root.hi[k])) finite = false;
                S.bound_centre[k] = (T)(0.5 * (root.lo[k] + root.hi[k]));
                r2 += 0.25 * (root.hi[k] - root.lo[k]) * (root.hi[k] - root.lo[k]);
            }
            if (finite) {
                S.bound_radius = (T)(std::sqrt(r2) * 1.001 + 1e-3 * scene_scale(d));
                S.far_origin = far_origin_limit(d);
            }
        }
        if (d.tlas_root < 0) {
            // No world tree (at most one boxed object, src/hitting.rs:165-170): the only f32 box tests are those of that
            // object's mesh tree, in ITS space.  Its transform is rigid, so the sphere around the mesh's root box, moved to
            // world space, bounds what far_shift has to bring a far ray near to.
            for (uint64_t e = 0; e < d.n_top; e++) {
                const int32_t oi = d.top_list[e];
                if (oi < 0 || d.objects[oi].kind != RT_OBJ_BLAS) continue;
                const rt_node& root = d.nodes[d.objects[oi].first];
                double c[3], r2 = 0;
                bool finite = true;
                for (int k = 0; k < 3; k++) {
                    if (!std::isfinite(root.lo[k]) || !std::isfinite(root.hi[k])) finite = false;
                    c[k] = 0.5 * (root.lo[k] + root.hi[k]);
                    r2 += 0.25 * (root.hi[k] - root.lo[k]) * (root.hi[k] - root.lo[k]);
                }
                if (!finite) continue;
                if (d.objects[oi].xform >= 0) {  // p' = m p + b  ->  p = m^T (p' - b)
                    const rt_xform& X = d.xforms[d.objects[oi].xform];
                    const double q[3] = {c[0] - X.b[0], c[1] - X.b[1], c[2] - X.b[2]};
                    for (int k = 0; k < 3; k++) c[k] = X.m[k] * q[0] + X.m[3 + k] * q[1] + X.m[6 + k] * q[2];
                }
                for (int k = 0; k < 3; k++) S.bound_centre[k] = (T)c[k];
                S.bound_radius = (T)(std::sqrt(r2) * 1.001 + 1e-3 * scene_scale(d));
                S.far_origin = far_origin_limit(d);
            }
        }
        // RT_ORDERED=0 (environment) walks trees in the reference's fixed left-first order (A/B runs).
        S.ordered = tune.ordered ? 1 : 0;
        // RT_COOP=0: the tail launch walks meshes with one lane per path throughout (A/B runs; the image is the same).
        // RT_COOP=n: groups of lanes answer a query in the tail launch while a wave has at most n live paths (0: never).
        S.coop = tune.coop;
        S.leaf_vote = tune.leaf_vote;
        S.leaf_frac = tune.leaf_frac;
        S.obj_frac = tune.obj_frac;
        S.obj_frac_pad = 0;
        S.coop_reserve = stack_need4;
        S.has_meshes = S.has_media = 0;
        for (uint64_t i = 0; i < d.n_objects; i++) {
            if (d.objects[i].kind == RT_OBJ_BLAS) S.has_meshes = 1;
            if (d.objects[i].kind == RT_OBJ_MEDIUM) S.has_media = 1;
        }
        if (tune.all_features) S.has_meshes = S.has_media = 1;  // RT_FEATURES=all: the general kernel (A/B runs, tests)
        // A medium bounded by a medium (src/objects.rs:404-409 takes any Hittable): only the general four-wide kernel
        // compiles the nested boundary query in (FEAT_NESTED_MEDIA), so such a scene takes it whatever else it holds.
        nested_media = false;
        for (uint64_t i = 0; i < d.n_media; i++)
            if (d.objects[d.media[i].boundary_object].kind == RT_OBJ_MEDIUM) nested_media = true;
        if (nested_media) {
            if (!S.nodes4) throw RtError{RT_ERR_UNSUPPORTED, "a medium bounded by a medium needs the four-wide trees (world tree too deep, or RT_BVH4=0)"};
            S.has_meshes = S.has_media = 1;
        }
        S.needs_uv = 0;
        for (uint64_t i = 0; i < d.n_materials; i++)
            if (d.materials[i].image >= 0 || d.materials[i].type == RT_MAT_CHECKERED) S.needs_uv = 1;
        build_small_scene(d);
        build_sweep(d);
        if (nested_media) {  // neither the sweep nor the LDS-staged walk carries the nested query
            S.sweep = nullptr;
            S.n_sweep = 0;
            S.lean = 0;
            S.sweep_items = nullptr;
            S.n_sweep_items = 0;
            std::memset(&small, 0, sizeof(small));
        }
        if (tune.debug)
            std::fprintf(stderr, "rt_upload_scene: objects %llu, stack_depth %d, four-wide %s (stack need %d, lds cap %d), sweep %d%s\n",
                         (unsigned long long)d.n_objects, stack_depth, S.nodes4 ? (sah_trees ? "sah" : "folded") : "no", stack_need4,
                         S.stack_cap, S.n_sweep, S.lean ? " lean" : "");
        HIP_CHECK(hipDeviceSynchronize());
        up_stats.ms_copy = ms_since(t_copy);
        up_stats.ms_total = ms_since(t_begin);
        up_stats.n_nodes2 = (uint32_t)n_nodes2;
        up_stats.n_nodes4 = (uint32_t)(S.nodes4 ? nodes4.n : 0);
        up_stats.sah_trees = sah_trees ? 1u : 0u;
        up_stats.needs_uv = S.needs_uv ? 1u : 0u;
        up_stats.walk = (S.n_sweep > 0 ? (S.lean == 1 ? 1u : S.lean == 2 ? 2u : 3u) : 0u) | ((uint32_t)S.n_sweep_items << 8);
        up_stats.bytes_device = nodes2.n * sizeof(rtd::DNode2) + (S.nodes4 ? nodes4.n * sizeof(rtd::DNode4) : 0) + objects.n * sizeof(rtd::DObject) +
                                blocks.n * sizeof(rtd::DBlock<T>) + xforms.n * sizeof(rtd::DXform<T>) + spheres.n * sizeof(rtd::DSphere<T>) +
                                moving_spheres.n * sizeof(rtd::DMovingSphere<T>) + rects.n * sizeof(rtd::DRect<T>) +
                                triangles.n * sizeof(rtd::DTriangle<T>) + planes.n * sizeof(rtd::DPlane<T>) + media.n * sizeof(rtd::DMedium<T>) +
                                materials.n * sizeof(rtd::DMaterial<T>) + images.n * sizeof(rtd::DImage) + texels.n + top_list.n * 4;
        have_scene = true;
    }

    // Worlds of a few objects are not walked as a tree: the leaves of the world BVH are listed in
    // the reference's depth-first order (left child first, src/hitting.rs:211-229) and every ray
    // tests all of them in that order (kernels.hip, WALK_SWEEP).  Subtrees under a zero-thickness
    // interior box are left out, as in build_nodes2: the reference culls them for every ray.
    // RT_SWEEP_MAX (environment, default 12) is the largest leaf count handled this way; 0 disables.
    void build_sweep(const rt_scene_desc& d) {
        S.sweep = nullptr;
        S.n_sweep = 0;
        S.lean = 0;
        S.sweep_items = nullptr;
        S.n_sweep_items = 0;
        S.sweep_items_pad = 0;
        if (d.tlas_root < 0) return;
        const long limit = tune.sweep_max;
        std::vector<int32_t> order, todo{d.tlas_root};
        while (!todo.empty()) {
            const rt_node& nd = d.nodes[todo.back()];
            todo.pop_back();
            if (nd.left < 0) {
                // a mesh is better entered through its own box in the tree walk (measured: knight 85 -> 114 ms)
                if (d.objects[-1 - nd.left].kind == RT_OBJ_BLAS) return;
                order.push_back(-1 - nd.left);
                if ((long)order.size() > limit) return;
                continue;
            }
            bool well_formed = true, flat = false;
            for (int k = 0; k < 3; k++) {
                if (!(nd.lo[k] <= nd.hi[k]) || !std::isfinite(nd.lo[k]) || !std::isfinite(nd.hi[k])) well_formed = false;
                if (nd.lo[k] == nd.hi[k]) flat = true;
            }
            if (well_formed && flat) continue;
            todo.push_back(nd.right);  // popped after the whole left subtree
            todo.push_back(nd.left);
        }
        if (order.empty()) return;
        // which sweep kernel: 1 = rect lists and spheres only (cornell, globe), 2 = no meshes and no media anywhere
        // (two_spheres: planes too), 0 = the general one (blue_ball's medium)
        S.lean = 1;
        for (uint64_t i = 0; i < d.n_objects; i++) {
            const int32_t k = d.objects[i].kind;
            if (k == RT_OBJ_BLAS || k == RT_OBJ_MEDIUM) S.lean = 0;
            else if (k != RT_OBJ_RECTS && k != RT_OBJ_SPHERE && k != RT_OBJ_MOVING_SPHERE && S.lean == 1) S.lean = 2;
        }
        sweep.upload(order);
        S.sweep = sweep.p;
        S.n_sweep = (int32_t)order.size();
        // A world that is nothing but swept rect lists (cornell): its rects one by one, for the lane groups at the end
        // of a tail launch (kernels.hip, sweep_hit_group).
        if (S.lean == 1 && d.n_top == 1 && d.top_list[0] == RT_TOP_BVH) {
            std::vector<int4> items;
            for (size_t k = 0; k < order.size(); k++) {
                const rt_object& ob = d.objects[order[k]];
                if (ob.kind != RT_OBJ_RECTS) {
                    items.clear();
                    break;
                }
                for (int32_t j = 0; j < ob.count; j++) items.push_back(int4{order[k], ob.first + j, ob.xform, (int32_t)k});
            }
            if (!items.empty() && items.size() <= 256) {
                sweep_items.upload(items);
                S.sweep_items = sweep_items.p;
                S.n_sweep_items = (int32_t)items.size();
            }
        }
    }

    // Packs the traversal records of a small scene into one blob for LDS staging (k_extend<.., true>).
    void build_small_scene(const rt_scene_desc& d) {
        std::memset(&small, 0, sizeof(small));
        if (d.n_nodes * sizeof(rtd::DNode<T>) > RT_LDS_SCENE_BYTES) return;
        {
            std::vector<rtd::DNode<T>> v(d.n_nodes);
            for (size_t i = 0; i < v.size(); i++) {
                cvt3(v[i].lo, d.nodes[i].lo);
                cvt3(v[i].hi, d.nodes[i].hi);
                if (sizeof(T) == 4) {  // f32 mode: round boxes outwards so that narrowing never culls a hit
                    for (int k = 0; k < 3; k++) {
                        if ((double)v[i].lo[k] > d.nodes[i].lo[k]) v[i].lo[k] = (T)std::nextafterf((float)v[i].lo[k], -INFINITY);
                        if ((double)v[i].hi[k] < d.nodes[i].hi[k]) v[i].hi[k] = (T)std::nextafterf((float)v[i].hi[k], INFINITY);
                    }
                }
                v[i].left = d.nodes[i].left;
                v[i].right = d.nodes[i].right;
            }
            nodes_own.upload(v);
        }
        struct Part { const void* dev; size_t bytes; };
        const Part parts[9] = {
            {nodes_own.p, d.n_nodes * sizeof(rtd::DNode<T>)},      {objects.p, d.n_objects * sizeof(rtd::DObject)},
            {xforms.p, d.n_xforms * sizeof(rtd::DXform<T>)},    {spheres.p, d.n_spheres * sizeof(rtd::DSphere<T>)},
            {moving_spheres.p, d.n_moving_spheres * sizeof(rtd::DMovingSphere<T>)},
            {rects.p, d.n_rects * sizeof(rtd::DRect<T>)},       {planes.p, d.n_planes * sizeof(rtd::DPlane<T>)},
            {media.p, d.n_media * sizeof(rtd::DMedium<T>)},     {top_list.p, d.n_top * sizeof(int32_t)}};
        size_t total = 0;
        uint32_t off[9];
        for (int k = 0; k < 9; k++) {
            off[k] = (uint32_t)total;
            total += (parts[k].bytes + 15) & ~(size_t)15;
        }
        if (total == 0 || total > RT_LDS_SCENE_BYTES || d.stack_depth > 16) return;
        small_blob.ensure(total);
        HIP_CHECK(hipMemset(small_blob.p, 0, total));
        for (int k = 0; k < 9; k++)
            if (parts[k].bytes)
                HIP_CHECK(hipMemcpy(small_blob.p + off[k], parts[k].dev, parts[k].bytes, hipMemcpyDeviceToDevice));
        small.blob = small_blob.p;
        small.bytes = (uint32_t)total;
        std::memcpy(small.off, off, sizeof(off));
    }

    static rtd::DQueue<T> queue(const Work<T>& w, int k) {
        T* b = w.qslab[k];
        size_t n = w.capacity;
        return rtd::DQueue<T>{b, (uint64_t)n, w.qpid[k]};
    }
    static rtd::DHits<T> hits(const Work<T>& w, bool with_ids) {
        T* b = w.hslab;
        size_t n = w.capacity;
        rtd::DHits<T> h{b, (uint64_t)n, w.hmat, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (with_ids) {
            h.owner = w.hids;
            h.sub = w.hids + n;
            h.world_index = w.hids + 2 * n;
            h.inst = w.hids + 3 * n;
            h.second = w.tminmax + 2 * n;
        }
        return h;
    }
    // k_extend.  RT_REFILL=64 sends world trees to k_extend_refill instead: node steps and object tests in separate
    // phases of the whole wave, rays handed out from a launch-wide cursor (counts[1], zeroed by the k_scan_tiles before);
    // n < 64 = lanes also take new rays while others of their wave still walk, as soon as n lanes wait.  It was the
    // kernel of world trees without meshes (blocks_ball 29.4 -> 23.0 ms when it came) until k_extend's own walk took
    // one step per round with a vote on the object tests (DScene::obj_frac): blocks_ball 17.3 -> 16.8 ms, 13 spp
    // 3.63 -> 3.25 with k_extend.
    int refill_lanes() const {
        int refill = 0;
        if (S.nodes4 && S.tlas_root4 >= 0 && S.n_sweep == 0 && !small.blob) {  // (Walk4 starts at the world tree's root)
            if (tune.refill >= 0) refill = tune.refill;
        }
        return refill;
    }
    void launch_extend(const Work<T>& w, hipStream_t st, const rtd::DBatch& B, rtd::DQueue<T> q, rtd::DHits<T> h, const T* tmin,
                       const T* tmax, uint32_t depth, uint32_t n_tiles, int grid) {
        const int refill = refill_lanes();
        if (refill > 0) grid = std::min(grid, num_cus * 4);
        rtd::Launch<T>::extend(S, small, B, q, h, tmin, tmax, depth, w.tile_offsets, n_tiles, stack_depth, grid, w.counts + 1, refill, st);
    }
    static uint32_t tiles_for(size_t n) { return (uint32_t)((n + RT_TILE - 1) / RT_TILE); }
    int grid_for(size_t n, int lanes_running = 1) {  // persistent blocks, grid-stride over the dense work index
        size_t blocks = (n + 255) / 256;
        // Blocks per CU that stride over the dense work index.  4 are resident (128 VGPRs); handing the work out in
        // 32 pieces per CU instead of 8 lets the dispatcher even out blocks that finish at different times:
        // cornell 22.3 -> 21.1 ms, knight 48.6 -> 47.2, blocks_ball 20.2 -> 20.0 (4: 23.0 / 49.4 / 20.3; 16: 21.4 /
        // 47.5 / 20.1; 64: 21.1 / 48.3 / 19.9; 128: 21.2 / 49.6 / 19.9; one block per 256 rays: 22.1 / 51.4 / 19.9).
        // RT_GRID_MULT=n (environment) overrides for A/B runs.
        const bool grid_mult_forced = tune.grid_mult != 0;
        const size_t per_cu = grid_mult_forced ? tune.grid_mult : 32;
        // (two lanes share the CUs: 8 per CU each measured best, knight 45.2 -> 43.8 ms, cornell 19.7 -> 19.3)
        size_t cap = (size_t)num_cus * (lanes_running > 1 && !grid_mult_forced ? 8 : per_cu);
        blocks = blocks < 1 ? 1 : blocks > cap ? cap : blocks;
        if (blocks > 8) blocks = (blocks + 7) & ~(size_t)7;  // whole groups of eight: k_extend's XCD-aware chunk order (a block without work leaves at once)
        return (int)blocks;
    }
    int tail_grid_for(size_t n) {  // the tail launch hands paths out from a cursor: no more blocks than stay resident (RT_TAIL_WAVES per SIMD)
        size_t blocks = (n + 255) / 256;
        const int env_blocks = tune.tail_blocks;  // (A/B runs of builds with other tail_waves())
        // blocks that stay resident per CU = the waves per SIMD the scene's tail kernel is compiled for (kernels.hip, tail_waves)
        const int resident = S.n_sweep > 0 && S.lean != 0 ? 3 : S.nodes4 && S.has_meshes ? 4 : tail_blocks_per_cu;
        const size_t cap = (size_t)num_cus * (env_blocks > 0 ? env_blocks : resident);
        return (int)(blocks < 1 ? 1 : blocks > cap ? cap : blocks);
    }
    void need_scene() {
        if (!have_scene) throw RtError{RT_ERR_NO_SCENE, "no scene uploaded (call rt_upload_scene first)"};
    }

    void trace(const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) override {
        need_scene();
        static_assert(sizeof(rt_ray) == 9 * sizeof(double), "rt_ray layout");
        Lane<T>& L = *lanes[0];
        size_t chunk_max = (size_t)std::min<uint64_t>(max_paths, 1u << 24);
        for (size_t base = 0; base < n; base += chunk_max) {
            size_t m = std::min(chunk_max, n - base);
            L.ensure(m, 0, true);
            const Work<T>& w = L.w;
            io_f64.ensure(m * 9);
            io_hits.ensure(m);
            HIP_CHECK(hipMemcpyAsync(io_f64.p, rays + base, m * sizeof(rt_ray), hipMemcpyHostToDevice, stream));
            rtd::DBatch B{0, 0, 0, 1, (uint32_t)base, 1u, seed};
            rtd::DQueue<T> q = queue(w, 0);
            rtd::DHits<T> h = hits(w, true);
            T* tmin = w.tminmax;
            T* tmax = w.tminmax + w.capacity;
            rtd::Launch<T>::load_rays(io_f64.p, (uint32_t)m, q, tmin, tmax, w.tile_counts[0], stream);
            rtd::launch_scan_tiles(w.tile_counts[0], tiles_for(m), w.tile_offsets, nullptr, w.counts, nullptr, nullptr, stream);
            launch_extend(w, stream, B, q, h, tmin, tmax, 0, tiles_for(m), grid_for(m));
            rtd::Launch<T>::second(S, B, q, h, tmin, (uint32_t)m, stream);  // rt_hit::second_t: a second, exhaustive walk
            rtd::Launch<T>::store_hits(q, h, (uint32_t)m, io_hits.p, stream);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(out + base, io_hits.p, m * sizeof(rt_hit), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
        }
    }

    void primary_rays(uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t sample, uint64_t seed, rt_ray* out) override {
        need_scene();
        size_t n = (size_t)W * (j1 - j0);
        if (n == 0) return;
        if (n > max_paths) throw RtError{RT_ERR_INVALID, "rt_primary_rays: band larger than max_paths"};
        Lane<T>& L = *lanes[0];
        L.ensure(n, 0, false);
        io_f64.ensure(n * 9);
        rtd::DBatch B{W, H, j0, (uint32_t)n, sample, 0u, seed};
        rtd::DQueue<T> q = queue(L.w, 0);
        rtd::Launch<T>::raygen(S, B, q, (uint32_t)n, L.w.tile_counts[0], stream);
        rtd::Launch<T>::store_rays(q, (uint32_t)n, io_f64.p, stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(out, io_f64.p, n * sizeof(rt_ray), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
    }

    // ---- the render loop.  A call's samples are cut into batches (sample ranges, in order); batch i runs on lane
    // i % K: raygen -> [extend -> shade | bounce] x bounces -> (tail launch) -> accumulate.  The host never waits
    // for a stream inside a batch: every k_scan_tiles stores the live count of its bounce in pinned memory, the host
    // reads what has arrived (RUNAHEAD bounces late at most) and uses it to stop when the paths are dead, to size the
    // grids and to start the tail launch; launches that find their queue empty leave at once.  k_accumulate of batch
    // i waits for that of batch i - 1 (an event), so that the sums are added in sample order whatever the lane count:
    // the image does not depend on K bit for bit.
    struct Span {
        size_t first;  // event index
        int kind;      // 0 = extend / 1 = shade / 2 = tail launch next to k_extend
    };
    struct BatchRun {
        bool active = false;
        uint32_t index = 0, bs0 = 0, bns = 0, n_paths = 0, n_tiles = 0, depth = 0, known_depth = 0, known_alive = 0;
        bool first_fused = false;  // the first bounce makes its own camera rays (k_bounce<.., FIRST = true>): no k_raygen, queue 0 never written
        rtd::DBatch B;
    };
    struct LaneRun {
        Lane<T>* lane = nullptr;
        BatchRun b;
        uint32_t next_batch = 0;  // next batch index this lane takes
        bool done = false;
        size_t ev = 0;
        std::vector<Span> spans;
        uint32_t n_extend = 0, n_shade = 0, n_tail = 0;  // launches (counted whether or not they are timed)
        uint64_t paths_first = 0;                          // paths of batches whose first bounce made the camera rays
        uint32_t deepest = 0;
    };
    enum StepResult { STEP_PROGRESS, STEP_BLOCKED, STEP_DONE };

    void render_accum(uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t s0, uint32_t ns, uint32_t max_bounces,
                      uint64_t seed, double* accum_dev, rt_stats* stats, bool fixed) override {
        need_scene();
        if (stats) std::memset(stats, 0, sizeof(*stats));
        const uint64_t npix = (uint64_t)W * (j1 - j0);
        if (npix == 0 || ns == 0) return;
        if (fixed && !fixed_lane) throw RtError{RT_ERR_UNSUPPORTED, "the static work area is not available on this device"};
        // Lanes: a second stream pays when each lane still gets a large batch (600 px, 100 spp: cornell 20.65 -> 19.3 ms,
        // knight 46.1 -> 43.8, blocks_ball 19.5 -> 18.7; at 13 spp the halves are too small and it loses 3 - 20 %), and
        // a band that fills the path budget leaves no room for a second lane's queues.  RT_LANE_MIN_PATHS overrides.
        const uint64_t lane_min_paths = tune.lane_min_paths;
        // While launches are timed one lane runs: a launch's duration is then its own, not stretched by the other lane's
        // kernels on the same CUs (what the roofline of a kernel is priced with).
        const int K_untimed = (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(lanes.size(), max_paths / npix),
                                                                            npix * ns / std::max<uint64_t>(1, lane_min_paths)));
        // RT_ONE_LANE=1: one lane at a time with the batches of the usual call (what launch timing does: for profiles
        // whose per-launch durations are to be a kernel's own)
        const int K = fixed || time_launches || tune.one_lane ? 1 : K_untimed;
        // (a call with launch timing runs one lane, but cuts its batches as the same call without timing would: the
        // lane's buffers then need not grow - freeing and allocating 9 GB cost the timed step of bench.py 3 ms)
        const uint64_t lane_paths = fixed ? (uint64_t)RT_NOALLOC_MAX_PATHS : std::max<uint64_t>(1, max_paths / (uint64_t)K_untimed);
        if (npix > lane_paths || npix > 0xffffffffull) {
            if (fixed)
                throw RtError{RT_ERR_INVALID, "rt_render_accum_device_noalloc: the row band holds more than " +
                                                  std::to_string(RT_NOALLOC_MAX_PATHS) + " pixels (the static work area's size); render fewer rows per call"};
            throw RtError{RT_ERR_INVALID, "row band holds more pixels than max_paths; render fewer rows per call"};
        }
        // batches: whole samples of every pixel of the band; at least one per lane when there are samples enough
        const uint64_t spb_cap = std::max<uint64_t>(1, lane_paths / npix);
        uint32_t n_batches = (uint32_t)((ns + spb_cap - 1) / spb_cap);
        n_batches = std::max(n_batches, std::min<uint32_t>((uint32_t)K, ns));
        const uint32_t spb = (ns + n_batches - 1) / n_batches;
        n_batches = (ns + spb - 1) / spb;
        const bool timing = stats != nullptr && time_launches && !fixed;
        // One fused launch per bounce pays off when the traversal is a sweep (no divergent walk: cornell 30.5 -> 25.2 ms
        // when it came; the sweeps that are not lean followed at the end of round 3, once the shading had lost its
        // per-lane rejection loop and its scratch memory: two_spheres 12.45 -> 10.0 ms, blue_ball 25.7 -> 22.0); with
        // a divergent tree walk the shading half runs at the walk's low occupancy and loses (knight 95 -> 107 ms).
        // RT_FUSE=0 / 1 (environment) overrides for A/B runs.
        const bool debug_counts = tune.debug;
        // World trees without meshes (blocks_ball) are fused too since the end of round 3: 16.8 -> 15.9 ms, 13 spp 3.22 -> 3.09
        // through a k_bounce compiled without the bounce loop; with meshes it is a tie (knight 30.2 / 29.7, 1 M triangles
        // 38.7 / 39.0) and the two kernels stay.
        const bool tree_without_meshes = S.n_sweep == 0 && !(small.blob && stack_depth <= 16) && S.nodes4 && !S.has_meshes;
        const bool fused = tune.fuse >= 0 ? tune.fuse != 0 : (S.n_sweep > 0 || tree_without_meshes);
        // Paths alive at or below which the tail launch takes over (RT_TAIL=n in the environment overrides, 0: never).
        // The tail launch hands its paths out from a cursor, so it stays efficient with many more paths than are
        // resident at a time; what it saves is the wait for the slowest ray of every bounce.
        uint32_t tail_threshold = fused ? tail_threshold_fused : tail_threshold_tree;
        // Meshes: with the queue order of DBatch::blocked the tail launch (walk and shading in one kernel, every path at
        // its own pace) beats a launch per bounce from 4 M paths down - but not for the first two bounces, whose rays
        // are many and alike (knight 35.8 -> 35.0 ms, 1 M triangles 48.6 -> 46.5; 13 spp, where the whole batch is
        // 4.7 M paths: 7.8 ms either way, 9.3 if the tail took bounce 0 too).
        uint32_t tail_threshold_late = tail_threshold;  // from the third bounce on
        if (!fused && S.nodes4 && S.has_meshes) tail_threshold_late = 4194304u;
        if (tune.tail >= 0) tail_threshold = tail_threshold_late = (uint32_t)tune.tail;
        // RT_PIXEL_BLOCKS=0: paths enter the queue along the pixel rows; =1: through 8 x 8 blocks of pixels, sample by
        // sample; unset: a wave holds several samples of a few neighbouring pixels (DBatch::blocked; same image always)
        const bool pixel_blocks = tune.pixel_blocks != 0;
        const uint32_t runahead = tune.runahead;  // bounces enqueued beyond the last one whose live count the host has seen

        std::vector<LaneRun> runs((size_t)K);
        for (int k = 0; k < K; k++) {
            runs[k].lane = fixed ? fixed_lane.get() : lanes[(size_t)k].get();
            runs[k].next_batch = (uint32_t)k;
            if ((uint32_t)k >= n_batches) runs[k].done = true;
            else runs[k].lane->ensure((size_t)(npix * spb), (size_t)(npix * spb), false);
        }
        Lane<T>& L0 = *runs[0].lane;
        hipEvent_t ev_begin = nullptr, ev_end = nullptr;
        if (stats && !fixed) {
            ev_begin = L0.event(runs[0].ev++);
            ev_end = L0.event(runs[0].ev++);
            HIP_CHECK(hipEventRecord(ev_begin, L0.stream));
        }
        // the other lanes start after what the caller has enqueued on the context's stream (e.g. the clearing of accum)
        if (K > 1) {
            HIP_CHECK(hipEventRecord(L0.ev_join, L0.stream));
            for (int k = 1; k < K; k++)
                if (!runs[k].done) HIP_CHECK(hipStreamWaitEvent(runs[k].lane->stream, L0.ev_join, 0));
        }
        for (int k = 0; k < K; k++) {
            if (runs[k].done) continue;
            const Work<T>& w = runs[k].lane->w;
            HIP_CHECK(hipMemsetAsync(w.ray_total, 0, 3 * sizeof(unsigned long long), runs[k].lane->stream));
            HIP_CHECK(hipMemsetAsync(w.counts + 2, 0, sizeof(uint32_t), runs[k].lane->stream));
        }
        uint32_t acc_recorded = 0;  // batches whose k_accumulate has been enqueued (they are enqueued in batch order)

        auto finish_batch = [&](LaneRun& R) -> bool {
            if (acc_recorded != R.b.index) return false;  // the batch before has not enqueued its k_accumulate yet
            Lane<T>& L = *R.lane;
            if (R.b.index > 0 && K > 1) {
                Lane<T>& prev = *runs[(size_t)((R.b.index - 1) % (uint32_t)K)].lane;
                HIP_CHECK(hipStreamWaitEvent(L.stream, prev.ev_acc, 0));
            }
            rtd::Launch<T>::accumulate(L.w.sample_rgb, (uint32_t)npix, R.b.bns, accum_dev, L.stream);
            HIP_CHECK(hipGetLastError());
            if (K > 1) HIP_CHECK(hipEventRecord(L.ev_acc, L.stream));
            acc_recorded++;
            R.b.active = false;
            R.next_batch += (uint32_t)K;
            return true;
        };

        auto step = [&](LaneRun& R) -> StepResult {
            if (R.done) return STEP_DONE;
            Lane<T>& L = *R.lane;
            const Work<T>& w = L.w;
            hipStream_t st = L.stream;
            BatchRun& b = R.b;
            if (!b.active) {
                if (R.next_batch >= n_batches) {
                    R.done = true;
                    return STEP_DONE;
                }
                b.index = R.next_batch;
                b.bs0 = s0 + b.index * spb;
                b.bns = std::min(spb, s0 + ns - b.bs0);
                b.n_paths = (uint32_t)(npix * b.bns);
                b.n_tiles = tiles_for(b.n_paths);
                b.B = rtd::DBatch{W, H, j0, (uint32_t)npix, b.bs0, 0u, seed, (uint32_t)(npix / W), pixel_blocks && W % 8 == 0 && (S.n_sweep == 0 || tune.pixel_blocks == 2) ? 1u : 0u};  // (tree walks only: a sweep has no lanes to keep in step; =2 forces it, A/B runs)
                if (b.B.blocked && (tune.pixel_blocks < 0 || tune.pixel_blocks == 2)) b.B.blocked = 2u;  // pixels x samples per wave
                b.depth = 0;
                b.known_depth = 0;
                b.known_alive = b.n_paths;
                b.active = true;
                L.ring ^= 1u;
                for (uint32_t k = 0; k < PEEK_SLOTS; k++) L.peek[L.ring * PEEK_SLOTS + k] = PEEK_UNKNOWN;
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
                // (every path stores its sample exactly once, when it ends, where DScene::emit_once holds: nothing to clear)
                if (!S.emit_once || max_bounces == 0) HIP_CHECK(hipMemsetAsync(w.sample_rgb, 0, (size_t)b.n_paths * 3 * sizeof(T), st));
                // Sweep scenes (one fused launch per bounce): the first bounce makes the camera rays itself - 84 B per path
                // neither written nor read, one launch fewer (cornell 0.65 ms of 19.5, globe of 3.0) - unless the batch is
                // small enough to start in the tail launch, which reads a queue.
                b.first_fused = fused && tune.fuse_raygen && max_bounces > 0 && b.n_paths > tail_threshold && S.n_sweep > 0;
                if (b.first_fused) {
                    R.paths_first += b.n_paths;
                    // the totals of the (unwritten) dense queue 0, clear the counts of queue 1, rays += n_paths
                    rtd::launch_scan_tiles(nullptr, b.n_tiles, w.tile_offsets, w.tile_counts[1], w.counts, w.ray_total, nullptr, st, b.n_paths);
                    return STEP_PROGRESS;
                }
                rtd::Launch<T>::raygen(S, b.B, queue(w, 0), b.n_paths, w.tile_counts[0], st);
                // offsets of queue 0, clear the counts of queue 1, rays += n_paths (if any bounce runs)
                rtd::launch_scan_tiles(w.tile_counts[0], b.n_tiles, w.tile_offsets, w.tile_counts[1], w.counts,
                                       max_bounces ? w.ray_total : nullptr, nullptr, st);
                return STEP_PROGRESS;
            }
            // what the kernels have told us so far
            while (b.known_depth < b.depth) {
                const uint32_t v = L.peek_value(b.known_depth + 1);
                if (v == PEEK_UNKNOWN) break;
                b.known_depth++;
                b.known_alive = v;
                if (debug_counts) std::fprintf(stderr, "  batch %u after bounce %u: %u paths alive\n", b.index, b.known_depth, v);
            }
            if (b.known_alive == 0 || b.depth >= max_bounces) return finish_batch(R) ? STEP_PROGRESS : STEP_BLOCKED;
            if (b.depth - b.known_depth > runahead) return STEP_BLOCKED;
            const uint32_t depth = b.depth;
            const int in = (int)(depth & 1u), out = in ^ 1;
            if (b.known_alive <= (depth >= 2 ? tail_threshold_late : tail_threshold)) {
                // The paths that are left go through all their remaining bounces in ONE launch, each at its own
                // pace, instead of one launch per bounce that waits for the slowest ray (k_bounce, DTail).  The
                // count is RUNAHEAD bounces old at most, i.e. an upper bound of what the launch will find.
                const rtd::DTail tail{w.ray_total + 1, w.counts + 2, w.counts + 3};
                if (fused) R.n_extend++;
                else R.n_tail++;
                if (timing) {
                    R.spans.push_back({R.ev, fused ? 0 : 2});
                    HIP_CHECK(hipEventRecord(L.event(R.ev), st));
                }
                rtd::Launch<T>::bounce(S, small, b.B, queue(w, in), queue(w, out), w.sample_rgb, depth, max_bounces, w.tile_offsets,
                                       w.tile_counts[out], b.n_tiles, stack_depth, tail_grid_for(b.known_alive), &tail, 0u, st);
                if (timing) HIP_CHECK(hipEventRecord(L.event(R.ev + 1), st));
                R.ev += 3;
                b.depth = max_bounces;  // nothing is left after it
                b.known_alive = 0;
                return STEP_PROGRESS;
            }
            const int grid = grid_for(b.known_alive, K);
            R.n_extend++;
            if (!fused) R.n_shade++;
            if (timing) {
                R.spans.push_back({R.ev, 0});
                HIP_CHECK(hipEventRecord(L.event(R.ev), st));
            }
            if (fused) {  // one launch per bounce; its time is reported as ms_extend
                rtd::Launch<T>::bounce(S, small, b.B, queue(w, in), queue(w, out), w.sample_rgb, depth, max_bounces, w.tile_offsets,
                                       w.tile_counts[out], b.n_tiles, stack_depth, grid, nullptr, depth == 0 && b.first_fused ? b.n_paths : 0u, st);
                if (timing) HIP_CHECK(hipEventRecord(L.event(R.ev + 1), st));
            } else {
                launch_extend(w, st, b.B, queue(w, in), hits(w, false), nullptr, nullptr, depth, b.n_tiles, grid);
                if (timing) {
                    HIP_CHECK(hipEventRecord(L.event(R.ev + 1), st));
                    R.spans.push_back({R.ev + 1, 1});
                }
                rtd::Launch<T>::shade(S, b.B, queue(w, in), hits(w, false), queue(w, out), w.sample_rgb, depth, max_bounces,
                                      w.tile_offsets, w.tile_counts[out], b.n_tiles, grid, st);
            }
            // offsets of the queue just written, clear the counts of the other one, rays += survivors, and the
            // live count to the host
            L.peek[L.peek_slot(depth + 1)] = PEEK_UNKNOWN;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            rtd::launch_scan_tiles(w.tile_counts[out], b.n_tiles, w.tile_offsets, w.tile_counts[in], w.counts, w.ray_total,
                                   L.peek_dev() + L.peek_slot(depth + 1), st);
            if (timing) HIP_CHECK(hipEventRecord(L.event(R.ev + 2), st));
            if (timing) R.ev += 3;
            R.deepest = std::max(R.deepest, depth + 1);
            b.depth = depth + 1;
            return STEP_PROGRESS;
        };

        uint64_t idle_spins = 0;
        uint32_t idle_checks = 0;
        for (;;) {
            bool all_done = true, progressed = false;
            for (int k = 0; k < K; k++) {
                const StepResult r = step(runs[(size_t)k]);
                if (r != STEP_DONE) all_done = false;
                if (r == STEP_PROGRESS) progressed = true;
            }
            if (all_done) break;
            if (!progressed) {
                // every lane waits for a live count: the streams are RUNAHEAD bounces ahead of what we know.  Now and
                // then make sure that they are still running (a count that never arrives must not hang the host).
                if (++idle_spins % 4096 == 0) {
                    bool busy = false;
                    for (int k = 0; k < K; k++) {
                        const hipError_t e = hipStreamQuery(runs[(size_t)k].lane->stream);
                        if (e == hipErrorNotReady) {
                            busy = true;
                            (void)hipGetLastError();
                        } else if (e != hipSuccess) {
                            throw RtError{RT_ERR_DEVICE, std::string("render stream failed: ") + hipGetErrorString(e)};
                        }
                    }
                    if (!busy && ++idle_checks > 64) throw RtError{RT_ERR_DEVICE, "internal: a live count never arrived from the device"};
                }
                __builtin_ia32_pause();
            } else {
                idle_checks = 0;
            }
        }
        // join: the context's stream continues after every lane
        for (int k = 1; k < K; k++) {
            if (runs[(size_t)k].next_batch == (uint32_t)k) continue;  // never ran
            HIP_CHECK(hipEventRecord(runs[(size_t)k].lane->ev_join, runs[(size_t)k].lane->stream));
            HIP_CHECK(hipStreamWaitEvent(L0.stream, runs[(size_t)k].lane->ev_join, 0));
        }
        if (ev_end) HIP_CHECK(hipEventRecord(ev_end, L0.stream));
        for (int k = 0; k < K; k++) {
            LaneRun& R = runs[(size_t)k];
            if (R.next_batch == (uint32_t)k) continue;
            const Work<T>& w = R.lane->w;
            R.lane->readback[3] = 0;
            HIP_CHECK(hipMemcpyAsync(R.lane->readback, w.ray_total, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.lane->stream));
            HIP_CHECK(hipMemcpyAsync(R.lane->readback + 3, w.counts + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, R.lane->stream));
        }
        for (int k = K - 1; k >= 0; k--) HIP_CHECK(hipStreamSynchronize(runs[(size_t)k].lane->stream));
        if (!stats) return;
        uint32_t deepest = 0;
        for (int k = 0; k < K; k++) {
            LaneRun& R = runs[(size_t)k];
            if (R.next_batch == (uint32_t)k) continue;
            stats->rays += R.lane->readback[0] + R.lane->readback[1];
            if (!fused) {
                stats->rays_tail += R.lane->readback[1] + R.lane->readback[2];
                stats->paths_tail += R.lane->readback[2];
            }
            deepest = std::max(deepest, std::min(std::max(R.deepest, (uint32_t)R.lane->readback[3]), max_bounces));
            stats->extend_launches += R.n_extend;
            stats->shade_launches += R.n_shade;
            stats->tail_launches += R.n_tail;
            stats->paths_first += R.paths_first;
        }
        stats->extend_kernel = fused ? 2u : refill_lanes() > 0 ? 1u : 0u;
        stats->paths = npix * ns;
        stats->batches = n_batches;
        stats->max_depth_reached = deepest;
        if (ev_begin) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_begin, ev_end));
            stats->ms_total = ms;
        }
        for (int k = 0; k < K; k++) {
            LaneRun& R = runs[(size_t)k];
            for (auto& sp : R.spans) {
                float ms = 0;
                HIP_CHECK(hipEventElapsedTime(&ms, R.lane->events[sp.first], R.lane->events[sp.first + 1]));
                if (sp.kind == 0) stats->ms_extend += ms;
                else if (sp.kind == 2) stats->ms_tail += ms;
                else stats->ms_shade += ms;
            }
        }
    }
};

}  // namespace

// ---- the work area of rt_render_accum_device_noalloc: device memory that exists from the moment the library's
// code object is loaded on a GPU until the process ends, so that the entry point never allocates.  Each GPU holds
// its own copy at its own address: host code never uses the array's name as an address, only the pointer
// hipGetSymbolAddress returned for that device (scratch_for_device).
constexpr size_t RT_NOALLOC_BYTES = work_bytes(sizeof(double), RT_NOALLOC_MAX_PATHS);
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];
std::mutex g_scratch_mutex;
// Every context on a GPU shares that GPU's one array: a call holds it for its whole duration, and a second call that
// arrives meanwhile (another context, another thread) is refused with RT_ERR_BUSY instead of racing on the queues.
std::atomic<int> g_scratch_busy[MAX_DEVICES];
struct ScratchHold {
    int dev;
    explicit ScratchHold(int d) : dev(d) {
        int expected = 0;
        if (dev < 0 || dev >= MAX_DEVICES || !g_scratch_busy[dev].compare_exchange_strong(expected, 1))
            throw RtError{RT_ERR_BUSY, "rt_render_accum_device_noalloc: the static work area of this GPU is in use by another call "
                                       "(every context on a GPU shares it; make the calls one after the other)"};
    }
    ~ScratchHold() { g_scratch_busy[dev].store(0); }
};
// The address of `scratch` on device `dev`, which must be the current device.
void* scratch_for_device(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        HIP_CHECK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[dev] = ptr;
    }
    return g_scratch_ptr[dev];
}
}  // namespace

struct rt_scene {
    rt::Scene scene;
    rt::FlatScene flat;
    rt_scene_desc desc;
};

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::unique_ptr<EngineBase> engine;
    std::string error;
    uint64_t max_paths = 0;
    DevBuf<double> tmp_accum;
    DevBuf<uint8_t> tmp_rgb8;
};

namespace rt {
// host_capi.cpp reports its errors through the same thread-local text.
void set_global_error(const std::string& msg) { g_error = msg; }
}  // namespace rt

// Makes `device` current for the calling thread and puts the caller's device back on the way out.
struct DeviceScope {
    int before = -1;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&before) != hipSuccess) before = -1;
        if (before != device) HIP_CHECK(hipSetDevice(device));
        else before = -1;
    }
    ~DeviceScope() {
        if (before >= 0) (void)hipSetDevice(before);
    }
};

template <typename F>
static int guard_global(F&& f) {
    try {
        f();
        return RT_OK;
    } catch (const RtError& e) {
        g_error = e.msg;
        return e.code;
    } catch (const rt::SceneError& e) {
        g_error = e.what();
        return RT_ERR_SCENE;
    } catch (const rt::Json5Error& e) {
        g_error = e.what();
        return RT_ERR_SCENE;
    } catch (const std::exception& e) {
        g_error = e.what();
        return RT_ERR_INVALID;
    }
}

template <typename F>
static int guard(rt_context* ctx, F&& f) {
    if (!ctx) {
        g_error = "null context";
        return RT_ERR_INVALID;
    }
    int rc = guard_global([&] {
        (void)hipGetLastError();  // an error left behind by the caller's own HIP calls is not ours to report
        DeviceScope scope(ctx->device);
        f();
    });
    if (rc != RT_OK) ctx->error = g_error;
    return rc;
}

extern "C" {

const char* rt_last_global_error(void) { return g_error.c_str(); }

static int load_scene_impl(rt::Scene&& sc, rt_scene** out) {
    std::unique_ptr<rt_scene> s(new rt_scene);
    s->scene = std::move(sc);
    s->flat = rt::flatten_scene(s->scene);
    s->desc = s->flat.desc();
    *out = s.release();
    return RT_OK;
}

int rt_scene_load_file(const char* path, const char* asset_root, rt_scene** out) {
    if (!path || !out) {
        g_error = "rt_scene_load_file: null argument";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    return guard_global([&] { load_scene_impl(rt::load_scene_file(path, asset_root ? asset_root : ""), out); });
}

int rt_scene_load_text(const char* text, const char* asset_root, rt_scene** out) {
    if (!text || !out) {
        g_error = "rt_scene_load_text: null argument";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    return guard_global([&] { load_scene_impl(rt::load_scene_text(text, asset_root ? asset_root : ""), out); });
}

void rt_scene_free(rt_scene* scene) { delete scene; }

const rt_scene_desc* rt_scene_get_desc(const rt_scene* scene) { return scene ? &scene->desc : nullptr; }

uint32_t rt_scene_image_height(const rt_scene* scene, uint32_t width) {
    return scene ? scene->scene.image_height(width) : 0;
}

int rt_create(int device_id, const rt_options* options, rt_context** out) {
    if (!out) {
        g_error = "rt_create: null out pointer";
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    return guard_global([&] {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count == 0)
            throw RtError{RT_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e)};
        if (device_id < 0 || device_id >= count) throw RtError{RT_ERR_INVALID, "rt_create: device id out of range"};
        DeviceScope scope(device_id);
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
        std::unique_ptr<rt_context> ctx(new rt_context);
        ctx->device = device_id;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        int precision = options ? options->precision : RT_PRECISION_F64;
        uint64_t max_paths = options && options->max_paths ? options->max_paths : (1ull << 25);
        if (max_paths > 0xffffffffull) max_paths = 0xffffffffull;
        // lanes: streams (each with its own queues) that share a call's batches; RT_LANES in the environment overrides
        int n_lanes = options && options->lanes > 0 ? options->lanes : RT_DEFAULT_LANES;
        if (const char* e = std::getenv("RT_LANES")) n_lanes = (int)std::strtol(e, nullptr, 10);
        n_lanes = std::max(1, std::min(8, n_lanes));
        void* fixed_base = scratch_for_device(device_id);  // (this device is current: DeviceScope above)
        if (precision == RT_PRECISION_F64)
            ctx->engine.reset(new Engine<double>(ctx->stream, prop.multiProcessorCount, max_paths, n_lanes, fixed_base, RT_NOALLOC_BYTES));
        else if (precision == RT_PRECISION_F32)
            ctx->engine.reset(new Engine<float>(ctx->stream, prop.multiProcessorCount, max_paths, n_lanes, fixed_base, RT_NOALLOC_BYTES));
        else
            throw RtError{RT_ERR_INVALID, "rt_create: unknown precision"};
        ctx->max_paths = max_paths;
        *out = ctx.release();
    });
}

void rt_destroy(rt_context* ctx) {
    if (!ctx) return;
    int before = -1;
    if (hipGetDevice(&before) != hipSuccess) before = -1;
    (void)hipSetDevice(ctx->device);
    ctx->engine.reset();
    ctx->tmp_accum.release();
    ctx->tmp_rgb8.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (before >= 0) (void)hipSetDevice(before);
    delete ctx;
}

int rt_set_launch_timing(rt_context* ctx, int on) {
    return guard(ctx, [&] { ctx->engine->set_launch_timing(on != 0); });
}

int rt_get_upload_stats(rt_context* ctx, rt_upload_stats* out) {
    return guard(ctx, [&] {
        if (!out) throw RtError{RT_ERR_INVALID, "rt_get_upload_stats: null out pointer"};
        ctx->engine->upload_stats(out);
    });
}

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

int rt_upload_scene(rt_context* ctx, const rt_scene_desc* scene) {
    return guard(ctx, [&] {
        if (!scene) throw RtError{RT_ERR_INVALID, "rt_upload_scene: null scene"};
        ctx->engine->upload(*scene);
    });
}

int rt_trace(rt_context* ctx, const rt_ray* rays, size_t n, uint64_t seed, rt_hit* out) {
    return guard(ctx, [&] {
        if (n == 0) return;
        if (!rays || !out) throw RtError{RT_ERR_INVALID, "rt_trace: null buffer"};
        ctx->engine->trace(rays, n, seed, out);
    });
}

int rt_primary_rays(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t sample,
                    uint64_t seed, rt_ray* out) {
    return guard(ctx, [&] {
        if (width < 2 || height < 2 || j0 > j1 || j1 > height) throw RtError{RT_ERR_INVALID, "rt_primary_rays: bad image geometry"};
        if (!out) throw RtError{RT_ERR_INVALID, "rt_primary_rays: null buffer"};
        ctx->engine->primary_rays(width, height, j0, j1, sample, seed, out);
    });
}

int rt_render_accum_device(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t s0,
                           uint32_t n_samples, uint32_t max_bounces, uint64_t seed, double* accum_dev, rt_stats* stats) {
    return guard(ctx, [&] {
        if (width < 2 || height < 2 || j0 > j1 || j1 > height)
            throw RtError{RT_ERR_INVALID, "rt_render_accum_device: bad image geometry (width, height >= 2, j0 <= j1 <= height)"};
        if (!accum_dev && j1 > j0 && n_samples) throw RtError{RT_ERR_INVALID, "rt_render_accum_device: null accum buffer"};
        if ((uint64_t)s0 + n_samples > 0xffffffffull) throw RtError{RT_ERR_INVALID, "sample index overflow"};
        ctx->engine->render_accum(width, height, j0, j1, s0, n_samples, max_bounces, seed, accum_dev, stats, false);
    });
}

int rt_render_accum_device_noalloc(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t s0,
                                   uint32_t n_samples, uint32_t max_bounces, uint64_t seed, double* accum_dev) {
    return guard(ctx, [&] {
        if (width < 2 || height < 2 || j0 > j1 || j1 > height)
            throw RtError{RT_ERR_INVALID, "rt_render_accum_device_noalloc: bad image geometry (width, height >= 2, j0 <= j1 <= height)"};
        if (!accum_dev && j1 > j0 && n_samples) throw RtError{RT_ERR_INVALID, "rt_render_accum_device_noalloc: null accum buffer"};
        if ((uint64_t)s0 + n_samples > 0xffffffffull) throw RtError{RT_ERR_INVALID, "sample index overflow"};
        if ((uint64_t)width * (j1 - j0) > RT_NOALLOC_MAX_PATHS)
            throw RtError{RT_ERR_INVALID, "rt_render_accum_device_noalloc: rows [j0, j1) hold " + std::to_string((uint64_t)width * (j1 - j0)) +
                                              " pixels, more than the " + std::to_string(RT_NOALLOC_MAX_PATHS) +
                                              " paths the library's static work area holds; render fewer rows per call"};
        ScratchHold hold(ctx->device);
        ctx->engine->render_accum(width, height, j0, j1, s0, n_samples, max_bounces, seed, accum_dev, nullptr, true);
    });
}

int rt_quantize_device(rt_context* ctx, const double* accum_dev, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb8_dev) {
    return guard(ctx, [&] {
        if (n_pixels == 0) return;
        if (!accum_dev || !rgb8_dev) throw RtError{RT_ERR_INVALID, "rt_quantize_device: null buffer"};
        rtd::launch_quantize(accum_dev, n_pixels, spp_total, rgb8_dev, ctx->stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int rt_render_rows(rt_context* ctx, uint32_t width, uint32_t height, uint32_t j0, uint32_t j1, uint32_t spp,
                   uint32_t max_bounces, uint64_t seed, uint8_t* rgb8, double* sum_rgb, rt_stats* stats) {
    return guard(ctx, [&] {
        if (width < 2 || height < 2 || j0 > j1 || j1 > height)
            throw RtError{RT_ERR_INVALID, "rt_render_rows: bad image geometry (width, height >= 2, j0 <= j1 <= height)"};
        if (stats) std::memset(stats, 0, sizeof(*stats));
        uint64_t npix = (uint64_t)width * (j1 - j0);
        if (npix == 0) return;
        ctx->tmp_accum.ensure(npix * 3);
        ctx->tmp_rgb8.ensure(npix * 3);
        HIP_CHECK(hipMemsetAsync(ctx->tmp_accum.p, 0, npix * 3 * sizeof(double), ctx->stream));
        // Bands of whole rows that fit the path budget (each pixel's samples stay together).
        uint32_t rows_per_band = (uint32_t)std::max<uint64_t>(1, ctx->max_paths / width);
        for (uint32_t b0 = j0; b0 < j1; b0 += rows_per_band) {
            uint32_t b1 = std::min(j1, b0 + rows_per_band);
            rt_stats part;
            ctx->engine->render_accum(width, height, b0, b1, 0, spp, max_bounces, seed,
                                      ctx->tmp_accum.p + (uint64_t)(b0 - j0) * width * 3, stats ? &part : nullptr, false);
            if (stats) {
                stats->rays += part.rays; stats->paths += part.paths;
                stats->ms_total += part.ms_total; stats->ms_extend += part.ms_extend; stats->ms_shade += part.ms_shade;
                stats->extend_launches += part.extend_launches; stats->shade_launches += part.shade_launches;
                stats->ms_tail += part.ms_tail; stats->rays_tail += part.rays_tail; stats->tail_launches += part.tail_launches;
                stats->paths_tail += part.paths_tail;
                stats->extend_kernel = part.extend_kernel;
                stats->batches += part.batches;
                stats->max_depth_reached = std::max(stats->max_depth_reached, part.max_depth_reached);
            }
        }
        if (rgb8) {
            rtd::launch_quantize(ctx->tmp_accum.p, npix, spp, ctx->tmp_rgb8.p, ctx->stream);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipMemcpyAsync(rgb8, ctx->tmp_rgb8.p, npix * 3, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (sum_rgb)
            HIP_CHECK(hipMemcpyAsync(sum_rgb, ctx->tmp_accum.p, npix * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

}  // extern "C"
