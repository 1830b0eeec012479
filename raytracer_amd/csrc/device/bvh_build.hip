// GPU build of the four-wide traversal tree of a triangle mesh ("GPU (LBVH/PLOC) build for
// >= 1 M triangles").  What it replaces is the tree of the reference's BVHNode::from_vec (src/hitting.rs:147-197) as
// a culling structure only: the answer of a walk - nearest leaf, exact ties to the leaf that is later in the
// reference's depth-first order - does not depend on the tree that culls (kernels.hip, blas_hit4_lane; DTriangle::
// dfs_rank comes from the reference-shaped tree the flattener builds), so any tree whose boxes contain their leaves
// gives the same image bit for bit.  The host's binned-SAH regrouping (tree_layout.hpp, build_sah4) is the default; this
// is the same job done on the GPU for meshes large enough that the build time matters:
//
//   1. centroid bounds of the leaves' (padded f32) boxes             k_centroid_bounds   (wave reduce + ordered-int atomics)
//   2. 63-bit Morton code of every centroid                          k_morton
//   3. radix sort of (code, leaf)                                    rocprim::radix_sort_pairs
//   4. binary radix tree over the sorted codes (Karras 2012)         k_hierarchy         (one thread per internal node)
//   5. boxes bottom-up (second arrival at a node continues upwards)  k_fit
//   6. fold to four children per node level by level, the child with the largest box opened first (as build_nodes4
//      folds the reference's tree), slots ordered largest box first (as sah4_node), 8-bit quantisation of the child
//      boxes exactly as tree_layout.hpp::quantize_range (same fmaf, planes only ever move outwards)   k_fold_level
//
// The nodes come out in breadth-first order (the four children of a node are neighbours in memory).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>  // (rocprim's texture_cache_iterator.hpp calls memset without including it)

#include <rocprim/rocprim.hpp>

#include "rt_device.h"

namespace rtd {
namespace {

constexpr int BB = 256;

__device__ __forceinline__ uint32_t ordered_bits(float f) {  // monotone map float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// bounds[0..2] = min, bounds[3..5] = max of the centroids, as ordered bits (initialised to ~0 / 0)
__global__ void __launch_bounds__(BB) k_centroid_bounds(const float* __restrict__ lo, const float* __restrict__ hi, uint32_t n,
                                                         uint32_t* __restrict__ bounds) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * BB + threadIdx.x; i < n; i += gridDim.x * BB)
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * lo[3 * i + a] + 0.5f * hi[3 * i + a];
            mn[a] = fminf(mn[a], c);
            mx[a] = fmaxf(mx[a], c);
        }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    // one set of atomics per block (atomics on one address are served one by one)
    __shared__ float part[BB / 64][6];
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) {
            part[threadIdx.x >> 6][a] = mn[a];
            part[threadIdx.x >> 6][3 + a] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x < 3) {
        float lo_all = part[0][threadIdx.x], hi_all = part[0][3 + threadIdx.x];
        for (int w = 1; w < BB / 64; w++) {
            lo_all = fminf(lo_all, part[w][threadIdx.x]);
            hi_all = fmaxf(hi_all, part[w][3 + threadIdx.x]);
        }
        atomicMin(&bounds[threadIdx.x], ordered_bits(lo_all));
        atomicMax(&bounds[3 + threadIdx.x], ordered_bits(hi_all));
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t x) {  // bit k of x -> bit 3 k
    uint64_t v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(BB) k_morton(const float* __restrict__ lo, const float* __restrict__ hi, uint32_t n,
                                                const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys,
                                                uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float mn = from_ordered_bits(bounds[a]), mx = from_ordered_bits(bounds[3 + a]);
        const float c = 0.5f * lo[3 * i + a] + 0.5f * hi[3 * i + a];
        const float extent = mx - mn;
        float t = extent > 0.0f ? (c - mn) / extent : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        q[a] = min(2097151u, (uint32_t)(t * 2097152.0f));
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    order[i] = i;
}

// Common leading bits of the codes at sorted positions i and j (ties broken by the positions: every key is distinct
// then, and the tree is a proper binary tree over the sorted order); -1 outside the array.
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees" (2012): internal node i covers
// the run of sorted leaves that starts or ends at i.  Children: k >= 0 internal node k, ~k leaf at sorted position k.
__global__ void __launch_bounds__(BB) k_hierarchy(const uint64_t* __restrict__ keys, int n, int32_t* __restrict__ left,
                                                   int32_t* __restrict__ right, int32_t* __restrict__ parent,
                                                   int32_t* __restrict__ leaf_parent) {
    const int i = (int)(blockIdx.x * BB + threadIdx.x);
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    if (first == gamma) {
        left[i] = ~gamma;
        leaf_parent[gamma] = i;
    } else {
        left[i] = gamma;
        parent[gamma] = i;
    }
    if (last == gamma + 1) {
        right[i] = ~(gamma + 1);
        leaf_parent[gamma + 1] = i;
    } else {
        right[i] = gamma + 1;
        parent[gamma + 1] = i;
    }
    if (i == 0) parent[0] = -1;
}

struct Box3 {
    float lo[3], hi[3];
};
__device__ __forceinline__ Box3 child_box(int32_t c, const float* __restrict__ leaf_lo, const float* __restrict__ leaf_hi,
                                          const uint32_t* __restrict__ order, const Box3* __restrict__ boxes) {
    if (c >= 0) return boxes[c];
    const uint32_t leaf = order[~c];
    Box3 b;
    for (int a = 0; a < 3; a++) {
        b.lo[a] = leaf_lo[3 * leaf + a];
        b.hi[a] = leaf_hi[3 * leaf + a];
    }
    return b;
}

// The same for a box another thread of the SAME launch has written: loads that bypass the CU's vector cache (a line of
// it may hold a stale copy of a neighbouring box read earlier).
__device__ __forceinline__ Box3 child_box_coherent(int32_t c, const float* __restrict__ leaf_lo, const float* __restrict__ leaf_hi,
                                                   const uint32_t* __restrict__ order, Box3* boxes) {
    if (c < 0) return child_box(c, leaf_lo, leaf_hi, order, boxes);
    Box3 b;
    for (int a = 0; a < 3; a++) {
        b.lo[a] = __hip_atomic_load(&boxes[c].lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.hi[a] = __hip_atomic_load(&boxes[c].hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}

// One thread per leaf walks towards the root; the first to arrive at a node stops, the second - whose sibling's box is
// complete and visible (fence before the counter) - writes the node's box and goes on.  Every wave ends.
__global__ void __launch_bounds__(BB) k_fit(int n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                             const int32_t* __restrict__ parent, const int32_t* __restrict__ leaf_parent,
                                             const float* __restrict__ leaf_lo, const float* __restrict__ leaf_hi,
                                             const uint32_t* __restrict__ order, Box3* boxes, uint32_t* arrivals) {
    const int i = (int)(blockIdx.x * BB + threadIdx.x);
    if (i >= n) return;
    int cur = leaf_parent[i];
    while (cur >= 0) {
        __threadfence();
        if (atomicAdd(&arrivals[cur], 1u) == 0u) return;
        __threadfence();
        const Box3 a = child_box_coherent(left[cur], leaf_lo, leaf_hi, order, boxes), b = child_box_coherent(right[cur], leaf_lo, leaf_hi, order, boxes);
        Box3 u;
        for (int k = 0; k < 3; k++) {
            u.lo[k] = fminf(a.lo[k], b.lo[k]);
            u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
        }
        // (volatile-like store: the reader is another thread, after its own fence and atomic)
        for (int k = 0; k < 3; k++) {
            __hip_atomic_store(&boxes[cur].lo[k], u.lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&boxes[cur].hi[k], u.hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cur = parent[cur];
    }
}

__device__ __forceinline__ float half_area(const Box3& b) {
    const float e0 = b.hi[0] - b.lo[0], e1 = b.hi[1] - b.lo[1], e2 = b.hi[2] - b.lo[2];
    return e0 * e1 + e1 * e2 + e2 * e0;
}

// tree_layout.hpp::quantize_range for one node, on the device: origin = the smallest lower plane, step = extent / 255
// raised until fmaf(255, step, origin) reaches the largest upper plane, every plane rounded outwards to a multiple and
// checked with the fmaf the kernels decode with.  false: not representable (non-finite bounds).
__device__ bool quantize_node(const Box3* box, const int32_t* child, int n_slots, DNode4* out) {
    DNode4 nd;
    for (int k = 0; k < 4; k++) nd.child[k] = k < n_slots ? child[k] : RT_EMPTY;
    bool ok = true;
    for (int a = 0; a < 3; a++) {
        float lo_min = INFINITY, hi_max = -INFINITY;
        for (int k = 0; k < n_slots; k++) {
            if (child[k] == RT_EMPTY) continue;
            if (!isfinite(box[k].lo[a]) || !isfinite(box[k].hi[a])) ok = false;
            lo_min = fminf(lo_min, box[k].lo[a]);
            hi_max = fmaxf(hi_max, box[k].hi[a]);
        }
        if (!(lo_min <= hi_max)) {
            lo_min = 0.0f;
            hi_max = 0.0f;
        }
        float step = fmaxf((hi_max - lo_min) / 255.0f, 1e-30f);
        for (int guard = 0; guard < 64 && fmaf(255.0f, step, lo_min) < hi_max; guard++) step = nextafterf(step, INFINITY);
        if (!isfinite(step) || fmaf(255.0f, step, lo_min) < hi_max) ok = false;
        nd.origin[a] = lo_min;
        nd.step[a] = step;
        for (int k = 0; k < 4; k++) {
            nd.qlo[a][k] = 0;
            nd.qhi[a][k] = 0;
            if (k >= n_slots || child[k] == RT_EMPTY || !ok) continue;
            int qlo = (int)floorf((box[k].lo[a] - lo_min) / step);
            qlo = min(255, max(0, qlo));
            while (qlo > 0 && fmaf((float)qlo, step, lo_min) > box[k].lo[a]) qlo--;
            int qhi = (int)ceilf((box[k].hi[a] - lo_min) / step);
            qhi = min(255, max(0, qhi));
            while (qhi < 255 && fmaf((float)qhi, step, lo_min) < box[k].hi[a]) qhi++;
            if (fmaf((float)qlo, step, lo_min) > box[k].lo[a] || fmaf((float)qhi, step, lo_min) < box[k].hi[a]) ok = false;
            nd.qlo[a][k] = (uint8_t)qlo;
            nd.qhi[a][k] = (uint8_t)qhi;
        }
    }
    *out = nd;
    return ok;
}

struct FoldItem {
    int32_t bnode;  // internal node of the binary tree
    uint32_t wide;  // index of the four-wide node it becomes (local to this tree)
};

// One level of the fold: item t turns binary node b into wide node w.  Its children that are internal nodes become the
// items of the next level, numbered in the order of the output queue (one atomic per wave on counters[0]).
// counters: [0] items written to `next`, [1] error flag.
__global__ void __launch_bounds__(BB) k_fold_level(const FoldItem* __restrict__ items, uint32_t n_items, FoldItem* __restrict__ next,
                                                    uint32_t next_wide_base, uint32_t* __restrict__ counters,
                                                    const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                    const float* __restrict__ leaf_lo, const float* __restrict__ leaf_hi,
                                                    const uint32_t* __restrict__ order, const int32_t* __restrict__ refs,
                                                    const Box3* __restrict__ boxes, DNode4* __restrict__ out, int32_t index_base) {
    const uint32_t t = blockIdx.x * BB + threadIdx.x;
    const bool valid = t < n_items;
    int32_t slot[4];
    Box3 box[4];
    int n_slots = 0;
    uint32_t wide = 0;
    if (valid) {
        const int32_t b = items[t].bnode;
        wide = items[t].wide;
        slot[0] = left[b];
        slot[1] = right[b];
        n_slots = 2;
        box[0] = child_box(slot[0], leaf_lo, leaf_hi, order, boxes);
        box[1] = child_box(slot[1], leaf_lo, leaf_hi, order, boxes);
        while (n_slots < 4) {  // open the internal child with the largest box
            int pick = -1;
            float best = -1.0f;
            for (int k = 0; k < n_slots; k++)
                if (slot[k] >= 0) {
                    const float a = half_area(box[k]);
                    if (a > best || pick < 0) {
                        best = a;
                        pick = k;
                    }
                }
            if (pick < 0) break;
            const int32_t c = slot[pick];
            for (int k = n_slots; k > pick + 1; k--) {
                slot[k] = slot[k - 1];
                box[k] = box[k - 1];
            }
            slot[pick] = left[c];
            slot[pick + 1] = right[c];
            box[pick] = child_box(slot[pick], leaf_lo, leaf_hi, order, boxes);
            box[pick + 1] = child_box(slot[pick + 1], leaf_lo, leaf_hi, order, boxes);
            n_slots++;
        }
        // largest box first (slot order only matters to rays that start inside several children): insertion sort
        for (int k = 1; k < n_slots; k++) {
            const int32_t s = slot[k];
            const Box3 bx = box[k];
            const float a = half_area(bx);
            int m = k;
            while (m > 0 && half_area(box[m - 1]) < a) {
                slot[m] = slot[m - 1];
                box[m] = box[m - 1];
                m--;
            }
            slot[m] = s;
            box[m] = bx;
        }
    }
    // the internal children of the wave's items take consecutive places in the next level's queue
    uint32_t mine = 0;
    for (int k = 0; k < n_slots; k++) mine += slot[k] >= 0 ? 1u : 0u;
    uint32_t prefix = mine;
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(prefix, off);
        if (lane >= (uint32_t)off) prefix += v;
    }
    const uint32_t wave_total = __shfl(prefix, 63);
    uint32_t start = 0;
    if (lane == 63 && wave_total) start = atomicAdd(&counters[0], wave_total);
    start = __shfl(start, 63) + (prefix - mine);
    if (!valid) return;
    int32_t child[4];
    for (int k = 0; k < n_slots; k++) {
        if (slot[k] >= 0) {
            next[start] = FoldItem{slot[k], next_wide_base + start};
            child[k] = index_base + (int32_t)(next_wide_base + start);
            start++;
        } else {
            child[k] = refs[order[~slot[k]]];
        }
    }
    if (!quantize_node(box, child, n_slots, &out[wide])) atomicOr(&counters[1], 1u);
}

// Node 0: the super-root (child 0 = the tree, slot 0 = its box), as build_sah4 makes it.
__global__ void k_super_root(int n, const float* __restrict__ leaf_lo, const float* __restrict__ leaf_hi, const int32_t* __restrict__ refs,
                             const Box3* __restrict__ boxes, DNode4* __restrict__ out, int32_t index_base, uint32_t* __restrict__ counters) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Box3 b;
    int32_t child;
    if (n == 1) {
        for (int a = 0; a < 3; a++) {
            b.lo[a] = leaf_lo[a];
            b.hi[a] = leaf_hi[a];
        }
        child = refs[0];
    } else {
        b = boxes[0];
        child = index_base + 1;
    }
    if (!quantize_node(&b, &child, 1, &out[0])) atomicOr(&counters[1], 1u);
}

// A tree built with index_base = 0 copied to its place in the scene's node array: child references move by `offset`.
__global__ void __launch_bounds__(BB) k_place(const DNode4* __restrict__ src, uint32_t n, int32_t offset, DNode4* __restrict__ dst) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    DNode4 nd = src[i];
    for (int k = 0; k < 4; k++)
        if (nd.child[k] >= 0) nd.child[k] += offset;
    dst[i] = nd;
}

struct Scratch {  // the device memory of one build: one allocation, carved in pieces of 256-byte alignment, freed when it ends
    unsigned char* base = nullptr;
    size_t size = 0, used = 0;
    static size_t round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    hipError_t reserve(size_t bytes) {
        size = bytes;
        return hipMalloc((void**)&base, bytes ? bytes : 256);
    }
    template <typename U>
    hipError_t get(U** out, size_t count) {
        const size_t bytes = round((count ? count : 1) * sizeof(U));
        if (used + bytes > size) return hipErrorOutOfMemory;
        *out = (U*)(base + used);
        used += bytes;
        return hipSuccess;
    }
    ~Scratch() {
        if (base) (void)hipFree(base);
    }
};

}  // namespace

#define RT_TRY(expr)                   \
    do {                               \
        const hipError_t e_ = (expr);  \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

// Builds the tree over n >= 1 leaves.  lo / hi: the leaves' padded f32 boxes, [n][3], host; refs: what a leaf's slot
// holds (-1 - triangle index), host.  *nodes_out: a device array the caller frees (hipFree) of *n_nodes DNode4,
// node 0 the super-root; child indices are index_base + the position in that array.  *levels: four-wide levels below
// the super-root (a walk holds at most 3 entries per level on its stack).
// Returns 0, a hipError_t (> 0), or -1 when a box cannot be stored in a DNode4 (the caller keeps the host's tree).
int gpu_build_tree4(const float* lo, const float* hi, const int32_t* refs, uint32_t n, int32_t index_base, hipStream_t st,
                    DNode4** nodes_out, uint32_t* n_nodes, int* levels) {
    *nodes_out = nullptr;
    *n_nodes = 0;
    *levels = 0;
    if (n == 0 || n > 0x3fffffffu) return -1;
    Scratch sc;
    {
        size_t temp_bytes = 0;
        if (n > 1) RT_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0u, 63u, st));
        const size_t N = n;
        RT_TRY(sc.reserve(2 * Scratch::round(3 * N * 4) + Scratch::round(N * 4) + 3 * 256 /* refs, counters, bounds */ + 2 * Scratch::round(N * 8) +
                          2 * Scratch::round(N * 4) /* order */ + 4 * Scratch::round(N * 4) /* left, right, parent, leaf_parent */ +
                          Scratch::round(N * sizeof(Box3)) + Scratch::round(N * 4) /* arrivals */ + 2 * Scratch::round(N * sizeof(FoldItem)) +
                          Scratch::round(temp_bytes) + 4096));
    }
    float *d_lo, *d_hi;
    int32_t* d_refs;
    uint32_t *d_bounds, *d_order, *d_order_sorted, *d_arrivals, *d_counters;
    uint64_t *d_keys, *d_keys_sorted;
    int32_t *d_left, *d_right, *d_parent, *d_leaf_parent;
    Box3* d_boxes;
    FoldItem *d_items[2];
    RT_TRY(sc.get(&d_lo, 3 * (size_t)n));
    RT_TRY(sc.get(&d_hi, 3 * (size_t)n));
    RT_TRY(sc.get(&d_refs, n));
    RT_TRY(sc.get(&d_counters, 2));
    RT_TRY(hipMemcpyAsync(d_lo, lo, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    RT_TRY(hipMemcpyAsync(d_hi, hi, 3 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    RT_TRY(hipMemcpyAsync(d_refs, refs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RT_TRY(hipMemsetAsync(d_counters, 0, 2 * sizeof(uint32_t), st));
    DNode4* d_out = nullptr;
    RT_TRY(hipMalloc((void**)&d_out, ((size_t)n + 1) * sizeof(DNode4)));  // (a fold of n leaves has at most n - 1 nodes, + the super-root)
    struct Guard {
        DNode4*& p;
        bool keep = false;
        ~Guard() {
            if (!keep && p) {
                (void)hipFree(p);
                p = nullptr;
            }
        }
    } guard{d_out};
    const int blocks_n = (int)((n + BB - 1) / BB);
    if (n == 1) {
        RT_TRY(sc.get(&d_boxes, 1));
        hipLaunchKernelGGL(k_super_root, dim3(1), dim3(64), 0, st, 1, d_lo, d_hi, d_refs, d_boxes, d_out, index_base, d_counters);
        RT_TRY(hipGetLastError());
        uint32_t flags[2];
        RT_TRY(hipMemcpyAsync(flags, d_counters, sizeof(flags), hipMemcpyDeviceToHost, st));
        RT_TRY(hipStreamSynchronize(st));
        if (flags[1]) return -1;
        guard.keep = true;
        *nodes_out = d_out;
        *n_nodes = 1;
        return 0;
    }
    RT_TRY(sc.get(&d_bounds, 6));
    RT_TRY(sc.get(&d_keys, n));
    RT_TRY(sc.get(&d_keys_sorted, n));
    RT_TRY(sc.get(&d_order, n));
    RT_TRY(sc.get(&d_order_sorted, n));
    RT_TRY(sc.get(&d_left, n - 1));
    RT_TRY(sc.get(&d_right, n - 1));
    RT_TRY(sc.get(&d_parent, n - 1));
    RT_TRY(sc.get(&d_leaf_parent, n));
    RT_TRY(sc.get(&d_boxes, n - 1));
    RT_TRY(sc.get(&d_arrivals, n - 1));
    RT_TRY(sc.get(&d_items[0], n));
    RT_TRY(sc.get(&d_items[1], n));
    RT_TRY(hipMemsetAsync(d_bounds, 0xff, 3 * sizeof(uint32_t), st));
    RT_TRY(hipMemsetAsync(d_bounds + 3, 0, 3 * sizeof(uint32_t), st));
    RT_TRY(hipMemsetAsync(d_arrivals, 0, (size_t)(n - 1) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(blocks_n < 512 ? blocks_n : 512), dim3(BB), 0, st, d_lo, d_hi, n, d_bounds);
    hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(BB), 0, st, d_lo, d_hi, n, d_bounds, d_keys, d_order);
    RT_TRY(hipGetLastError());
    {
        size_t temp_bytes = 0;
        RT_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_order, d_order_sorted, (size_t)n, 0u, 63u, st));
        unsigned char* d_temp;
        RT_TRY(sc.get(&d_temp, temp_bytes));
        RT_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_order, d_order_sorted, (size_t)n, 0u, 63u, st));
    }
    hipLaunchKernelGGL(k_hierarchy, dim3((int)((n - 1 + BB - 1) / BB)), dim3(BB), 0, st, d_keys_sorted, (int)n, d_left, d_right, d_parent, d_leaf_parent);
    hipLaunchKernelGGL(k_fit, dim3(blocks_n), dim3(BB), 0, st, (int)n, d_left, d_right, d_parent, d_leaf_parent, d_lo, d_hi, d_order_sorted, d_boxes, d_arrivals);
    hipLaunchKernelGGL(k_super_root, dim3(1), dim3(64), 0, st, (int)n, d_lo, d_hi, d_refs, d_boxes, d_out, index_base, d_counters);
    RT_TRY(hipGetLastError());
    // the fold, level by level: the host reads one count per level (a tree of a million leaves has ~12-25 levels)
    const FoldItem first{0, 1u};
    RT_TRY(hipMemcpyAsync(d_items[0], &first, sizeof(first), hipMemcpyHostToDevice, st));
    uint32_t n_items = 1, wide_total = 2;  // super-root + the root's node
    int cur = 0, depth = 0;
    while (n_items > 0) {
        if (++depth > 4096) return -1;
        RT_TRY(hipMemsetAsync(d_counters, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_fold_level, dim3((int)((n_items + BB - 1) / BB)), dim3(BB), 0, st, d_items[cur], n_items, d_items[cur ^ 1], wide_total,
                           d_counters, d_left, d_right, d_lo, d_hi, d_order_sorted, d_refs, d_boxes, d_out, index_base);
        RT_TRY(hipGetLastError());
        uint32_t flags[2] = {0, 0};
        RT_TRY(hipMemcpyAsync(flags, d_counters, sizeof(flags), hipMemcpyDeviceToHost, st));
        RT_TRY(hipStreamSynchronize(st));
        if (flags[1]) return -1;
        if ((uint64_t)wide_total + flags[0] > (uint64_t)n + 1) return -1;  // (cannot happen for a binary tree over n leaves)
        wide_total += flags[0];
        n_items = flags[0];
        cur ^= 1;
    }
    guard.keep = true;
    *nodes_out = d_out;
    *n_nodes = wide_total;
    *levels = depth;
    return 0;
}

void gpu_place_tree4(const DNode4* src, uint32_t n, int32_t offset, DNode4* dst, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_place, dim3((int)((n + BB - 1) / BB)), dim3(BB), 0, st, src, n, offset, dst);
}

}  // namespace rtd
