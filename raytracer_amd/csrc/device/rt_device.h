// Device-side scene layout (HBM-resident SoA) and launch interface of the gfx950 kernels.
//
// Everything here is templated on the arithmetic type T: double = the reference's
// arithmetic (src/math.rs:9-13, every scalar is f64), float = the declared fast mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

template <typename T>
struct alignas(16) DNode {  // reference-shaped node with its own box (f64: 64 B); small worlds only
    T lo[3], hi[3];
    int32_t left, right;  // interior: children; leaf: left = -1 - payload, right = RT_LEAF_*
};

// Traversal node: the (padded, outward-rounded f32) boxes of BOTH children live in the parent, so
// one 64-byte fetch decides a whole level.  child >= 0: interior DNode2 index; child < 0: leaf
// payload -1 - child (triangle index in a mesh BVH, object index in the world BVH); RT_EMPTY: no
// child / a child the reference always culls (zero-thickness box).
struct alignas(16) DNode2 {
    float lo[2][3], hi[2][3];
    int32_t child[2];
    int32_t block6;    // RT_OBJ_RECTS with the six sides of a Block in Block::new's order (axes z, z, y, y, x, x)
    int32_t first4;    // RT_OBJ_BLAS: root in DScene::nodes4
};
static constexpr int32_t RT_EMPTY = (int32_t)0x80000000;

// Four-wide traversal node as the trees are built on the host (tree_layout.hpp): the reference's
// binary tree with every other level folded away, or regrouped by SAH.  Child references as in
// DNode2; bounds are the padded, outward-rounded f32 boxes, per axis across the four slots.
struct WideNode4 {
    float lo[3][4], hi[3][4];
    int32_t child[4];
    void set_box(int slot, const float* blo, const float* bhi) {
        for (int a = 0; a < 3; a++) {
            lo[a][slot] = blo[a];
            hi[a][slot] = bhi[a];
        }
    }
};

// The same node as the kernels fetch it (64 B, four 16-byte loads - the walks are bound by the number
// of cache-line look-ups their scattered per-lane fetches cost): the child boxes are
// 8-bit offsets from the node's own corner, plane = origin[axis] + q * step[axis], rounded outwards
// when the node is made, so a decoded box contains the padded f32 box it stands for.
struct alignas(16) DNode4 {
    float origin[3], step[3];
    uint8_t qlo[3][4], qhi[3][4];  // [axis][slot]
    int32_t child[4];
};
static_assert(sizeof(DNode4) == 64, "DNode4 is four 16-byte loads");

template <typename T>
struct alignas(16) DXform {  // world -> object: p' = m p + b
    T m[9], b[3];
};

template <typename T>
struct alignas(16) DSphere {
    T centre[3], radius;
    int32_t material, owner;
};

template <typename T>
struct alignas(16) DMovingSphere {
    T centre0[3], centre1[3], time0, time1, radius;
    int32_t material, owner;
};

template <typename T>
struct alignas(16) DRect {
    T a0, a1, b0, b1, k;
    int32_t axis, facing_positive, material, owner, sub, pad;
};

#ifndef RT_TRI_ALIGN
#define RT_TRI_ALIGN 16
#endif
template <typename T>
struct alignas(sizeof(T) == 8 ? RT_TRI_ALIGN : 16) DTriangle {
    T point[3], vec1[3], vec2[3], normal[3];
    int32_t material, owner, sub;
    int32_t dfs_rank;  // position of the triangle's leaf in the depth-first order of its mesh BVH (tie rule)
};

template <typename T>
struct alignas(16) DPlane {
    T point[3], vec1[3], vec2[3], normal[3], uv_repeat;
    int32_t material, owner;
};

template <typename T>
struct alignas(16) DMedium {
    T neg_inv_density;
    int32_t boundary_object, material, owner;
    uint32_t rng_key;
};

struct alignas(16) DObject {
    int32_t kind, xform, first, count;
    int32_t world_index;
    int32_t dfs_rank;  // position of the object's leaf in the depth-first order of the world BVH (tie rule)
    int32_t block6;    // RT_OBJ_RECTS that are the six faces of one box (Block::new's order: axes z, z, y, y, x, x): 1 + index in DScene::blocks
    int32_t first4;    // RT_OBJ_BLAS: root in DScene::nodes4
};

template <typename T>
struct alignas(16) DBlock {  // the two corners the six rects of a Block were made from
    T lo[3], hi[3];
};

template <typename T>
struct alignas(16) DMaterial {
    T colour[3], param;
    int32_t type, image, odd, even;
};

struct DImage {
    int32_t width, height;
    uint64_t texel_offset;
};

template <typename T>
struct DCamera {
    T origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    T lens_radius, start_time, end_time;
};

template <typename T>
struct DSky {
    int32_t gradient;
    T colour[3], unit_dir[3], colour0[3], colour1[3];
};

template <typename T>
struct DScene {
    const DNode<T>* nodes;  // own-box nodes of the world BVH; valid only inside k_extend<.., LDS_SCENE = true>
    const DNode2* nodes2;
    const DNode4* nodes4;  // world and mesh BVHs, four-wide (nullptr: walk nodes2)
    const DObject* objects;
    const DBlock<T>* blocks;
    const DXform<T>* xforms;
    const DSphere<T>* spheres;
    const DMovingSphere<T>* moving_spheres;
    const DRect<T>* rects;
    const DTriangle<T>* triangles;
    const DPlane<T>* planes;
    const DMedium<T>* media;
    const DMaterial<T>* materials;
    const DImage* images;
    const uint8_t* texels;
    const int32_t* top_list;
    const int32_t* sweep;  // leaves (object indices) of a small world BVH in depth-first order; n_sweep == 0: walk the tree
    // the rects of a swept world of rect lists only, one by one in the sweep's order: {object, rect, transform, place of the
    // object in the sweep} - what a group of lanes shares out when it answers one ray's query (sweep_hit_group); nullptr: no groups
    const int4* sweep_items;
    int32_t n_sweep_items, sweep_items_pad;
    int32_t n_top, tlas_root /* DNode2 super-root */, tlas_root_own /* DNode index */, n_sweep;
    int32_t lean;     // 1: every object is a rect list or a sphere (no meshes, media, planes, lone triangles); 2: no meshes and no media (WALK_SWEEP_FLAT)
    int32_t ordered;  // tree walks enter the nearer child first (results are the same either way)
    int32_t needs_uv;  // some material reads (u, v): image textures, checkers
    int32_t tlas_root4;  // world BVH in nodes4 (valid when nodes4 != nullptr)
    int32_t stack_cap;   // > 0: entries of the LDS stack a four-wide walk may use before it overflows (tests)
    int32_t has_meshes, has_media;  // object kinds present: pick the four-wide kernel compiled for them
    int32_t emit_once;  // no checker has a light among its parts: a path adds to its sample at most once, when it ends (plain store)
    // The tail launch answers the queries of a wave that has at most `coop` live paths left by groups of lanes
    // (blas_hit4_group); coop_reserve: stack entries a one-lane walk of the scene's trees can need.
    int32_t coop, coop_reserve;
    // blas_hit4_lane: the wave runs its exact triangle tests when leaf_frac sixteenths of the lanes still walking the mesh
    // hold a leaf (at most leaf_vote lanes; leaf_frac = 0: leaf_vote lanes); the others wait with their leaf in hand
    int32_t leaf_vote, leaf_frac;
    // closest_hit's walk of the world tree: the same rule for its leaves, the exact object tests (0: tested at once)
    int32_t obj_frac, obj_frac_pad;
    // Rays that start further than far_origin from the origin of coordinates (paths scattered off an unbounded Plane)
    // are moved along their line to the bounding sphere of the world BVH for the f32 box tests (kernels.hip, far_shift):
    // the padding of the boxes only covers the f32 rounding of origins near the scene.
    float far_origin;
    T bound_centre[3], bound_radius;
    DCamera<T> camera;
    // The sky's colours are read where a ray misses, through this pointer to a device copy: as kernel arguments the 25
    // words were fetched at every kernel's entry and kept in spilled SGPRs by kernels whose rays rarely miss.
    const DSky<T>* sky;
};

// Small scenes: one contiguous copy of nodes (own-box form) | objects | xforms | spheres | moving_spheres | rects |
// planes | media | top_list (byte offsets in that order, each 16-byte aligned) that every block
// of k_extend stages into LDS.  blob == nullptr when the scene is too large.
#define RT_LDS_SCENE_BYTES 16384
struct DSmallScene {
    const unsigned char* blob;
    uint32_t bytes;  // multiple of 16, <= RT_LDS_SCENE_BYTES
    uint32_t off[9];
};

// Ray / path queue: one SoA slab of `capacity` entries per field (coalesced per field).
template <typename T>
struct DQueue {
    // Field k of entry i is slab[k * cap + i]: one pointer and one stride as kernel arguments instead of ten pointers
    // (the lean k_bounce restored its 22 queue pointers from spilled SGPRs with ~190 v_readlane per wave and bounce).
    T* slab;
    uint64_t cap;
    uint32_t* pid;  // path index inside the batch
#ifdef __HIPCC__
    // (the empty asm keeps the compiler from computing all ten field pointers of both queues at the kernel's entry and
    // holding them in spilled SGPRs: a field's address is two scalar instructions where it is used)
    __device__ __forceinline__ T* field(uint64_t k) const {
        uint64_t stride = cap;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(stride));
#endif
        return slab + k * stride;
    }
#define RT_QUEUE_FIELD(name, k) __device__ __forceinline__ T* name() const { return field(k); }
#else
#define RT_QUEUE_FIELD(name, k) T* name() const { return slab + (k) * cap; }
#endif
    RT_QUEUE_FIELD(ox, 0) RT_QUEUE_FIELD(oy, 1) RT_QUEUE_FIELD(oz, 2)  // Ray (src/math.rs:201-205)
    RT_QUEUE_FIELD(dx, 3) RT_QUEUE_FIELD(dy, 4) RT_QUEUE_FIELD(dz, 5) RT_QUEUE_FIELD(time, 6)
    RT_QUEUE_FIELD(tr, 7) RT_QUEUE_FIELD(tg, 8) RT_QUEUE_FIELD(tb, 9)  // path throughput (product of attenuations)
#undef RT_QUEUE_FIELD
};

// Hit records written by extend, read by shade (HitRecord, src/hitting.rs:33-41).
template <typename T>
struct DHits {
    T* slab;  // t, nx, ny, nz, u, v: field k of entry i is slab[k * cap + i] (as DQueue)
    uint64_t cap;
#ifdef __HIPCC__
    __device__ __forceinline__ T* field(uint64_t k) const {
        uint64_t stride = cap;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(stride));
#endif
        return slab + k * stride;
    }
    __device__ __forceinline__ T* t() const { return field(0); }
    __device__ __forceinline__ T* nx() const { return field(1); }
    __device__ __forceinline__ T* ny() const { return field(2); }
    __device__ __forceinline__ T* nz() const { return field(3); }
    __device__ __forceinline__ T* u() const { return field(4); }
    __device__ __forceinline__ T* v() const { return field(5); }
#endif
    uint32_t* mat_front;                    // material | front_face << 31; RT_MISS for a miss
    int32_t *owner, *sub, *world_index;     // only written when non-null (rt_trace)
    int32_t* inst;                          // rt_trace: the DObject index of the instance that was hit
    T* second;                              // rt_trace: rt_hit::second_t (k_second)
};
static constexpr uint32_t RT_MISS = 0x7fffffffu;
#define RT_STACK_OVERFLOW 40  // per-lane traversal-stack entries beyond the LDS part (private memory, rarely touched)
#define RT_TILE 4096u  // queue slots per tile (kernels.hip: TILE)

struct DBatch {      // which pixels / samples the paths of a batch stand for
    uint32_t width, height, j0, npix;  // npix = width * rows in the band
    uint32_t s0;                       // first sample index of the batch
    uint32_t trace_mode;               // 1: rt_trace - pixel = s0 + pid (global ray index), sample = 0
    uint64_t seed;
    // blocked != 0 (width % 8 == 0): k_raygen puts the paths of a sample into the queue through 8 x 8 blocks of
    // pixels (strips of 8 rows, runs of 8 columns; a last strip of fewer rows along its rows) instead of along the
    // rows: a wave's 64 paths then start in a square of the image, not on a 64-pixel line, and stay neighbours through
    // the bounces (the queues keep the order).  A path is still named by pid = sample * npix + y * W + x, which the
    // queue carries: nothing after k_raygen knows of the order.  Measured on the knight's primary rays: 32 % fewer
    // wave-steps in k_extend (scripts/exp_regroup.py).
    // blocked == 2: a wave holds n samples each of 64 / n neighbouring pixels, n as large as
    // the batch's sample count allows (64, else 32, 16, ...): the 64 paths of a wave start as nearly the same ray.
    // Knight, 100 spp (two batches of 50: n = 32, 16, 2): 39.4 -> 36.4 ms; 128 spp: 49.7 -> 43.6.
    uint32_t rows, blocked;            // rows = npix / width
};

// k_bounce as the tail of a batch: with ray_total != nullptr the kernel carries every path of its
// input queue through all remaining bounces (nothing is written to the output queue) and adds the
// closest-hit queries beyond the first / the deepest bounce reached to these counters.
struct DTail {
    unsigned long long* ray_total;  // [0] += queries beyond a path's first bounce here, [1] += paths taken
    uint32_t* deepest;
    uint32_t* cursor;               // zeroed (k_scan_tiles): the launch hands its paths out from here
};

// Launch wrappers (kernels.hip).  `stack` selects the LDS traversal-stack depth (16/32/64).
template <typename T>
struct Launch {
    static void raygen(const DScene<T>& S, const DBatch& B, DQueue<T> q, uint32_t n_paths, uint32_t* tile_counts,
                       hipStream_t st);
    static void extend(const DScene<T>& S, const DSmallScene& small, const DBatch& B, DQueue<T> q, DHits<T> h, const T* tmin, const T* tmax,
                       uint32_t depth, const uint32_t* tile_offsets, uint32_t n_tiles, int stack, int grid_blocks,
                       uint32_t* cursor /* zeroed */, int refill /* 0: one ray per lane and grid stride */, hipStream_t st);
    // extend + shade of one bounce in one launch (render path).
    static void bounce(const DScene<T>& S, const DSmallScene& small, const DBatch& B, DQueue<T> qin, DQueue<T> qout,
                       T* sample_rgb, uint32_t depth, uint32_t max_bounces, const uint32_t* tile_offsets,
                       uint32_t* tile_counts_out, uint32_t n_tiles, int stack, int grid_blocks, const DTail* tail,
                       uint32_t n_first /* > 0: first bounce of a dense batch of n_first paths, camera rays made in the kernel */, hipStream_t st);
    static void shade(const DScene<T>& S, const DBatch& B, DQueue<T> qin, DHits<T> h, DQueue<T> qout, T* sample_rgb,
                      uint32_t depth, uint32_t max_bounces, const uint32_t* tile_offsets, uint32_t* tile_counts_out,
                      uint32_t n_tiles, int grid_blocks, hipStream_t st);
    static void accumulate(const T* sample_rgb, uint32_t npix, uint32_t n_samples, double* accum, hipStream_t st);
    static void load_rays(const double* rays9, uint32_t n, DQueue<T> q, T* tmin, T* tmax, uint32_t* tile_counts,
                          hipStream_t st);
    // rt_trace: rt_hit::second_t of rays 0 .. n-1 (slot i = ray i) into h.second, after extend has filled h
    static void second(const DScene<T>& S, const DBatch& B, DQueue<T> q, DHits<T> h, const T* tmin, uint32_t n, hipStream_t st);
    static void store_hits(DQueue<T> q, DHits<T> h, uint32_t n, void* out_rt_hit, hipStream_t st);
    static void store_rays(DQueue<T> q, uint32_t n, double* rays9, hipStream_t st);
};

// Exclusive scan of per-tile counts -> tile_offsets[0..n_tiles]; zeroes counts_to_zero[0..n_tiles) if
// non-null; counts[0] = total, counts[1] = counts[3] = 0 (the cursors of the next launch); *ray_total += total
// if non-null; *host_total = total (pinned host memory, system-scope store) if non-null.
// tile_counts == nullptr: the queue is a dense one of dense_total entries that has not been written (no offsets are made).
void launch_scan_tiles(const uint32_t* tile_counts, uint32_t n_tiles, uint32_t* tile_offsets, uint32_t* counts_to_zero,
                       uint32_t* counts, unsigned long long* ray_total, uint32_t* host_total, hipStream_t st, uint32_t dense_total = 0);
void launch_quantize(const double* accum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb8, hipStream_t st);

// bvh_build.hip: the four-wide tree over n >= 1 leaves built on the GPU (Morton order, binary radix tree, bottom-up
// boxes, fold to four children, 8-bit quantisation).  lo / hi: the leaves' padded f32 boxes [n][3] (host), refs: a
// leaf's child reference (host).  *nodes_out: device array of *n_nodes DNode4 that the caller frees with hipFree,
// node 0 the super-root, child references = index_base + position in the array; *levels: four-wide levels.
// Returns 0, a hipError_t (> 0), or -1 when the boxes cannot be stored (the caller keeps the host's tree).
int gpu_build_tree4(const float* lo, const float* hi, const int32_t* refs, uint32_t n, int32_t index_base, hipStream_t st,
                    DNode4** nodes_out, uint32_t* n_nodes, int* levels);
// Copies such a tree (built with index_base = 0) to dst, its child references moved by `offset` (= dst's index in the scene's array).
void gpu_place_tree4(const DNode4* src, uint32_t n, int32_t offset, DNode4* dst, hipStream_t st);

}  // namespace rtd
