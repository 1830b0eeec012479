// gfx950 (MI355X, wave64) kernels of the wavefront path tracer:
//   raygen  -> [extend -> shade(+compaction)] x bounces -> accumulate -> quantize
//
// What each stage replaces in the reference (FBemf/raytracer @ v0):
//   raygen      src/main.rs:132-135 + Camera::find_ray (src/camera.rs:67-75)
//   extend      world.hit(ray, 0.001, inf) (src/hitting.rs:21): BVHNode::hit (:211-229),
//               AABB::intersects (:249-268), Vec<..>::hit (:88-104), the primitives of
//               src/objects.rs, the instance transforms of src/transforms.rs
//   shade       Material::emitted / scatter (src/materials.rs), textures (src/textures.rs),
//               the sky closure (src/config.rs:405-421) and the recursion of cast_ray
//               (src/hitting.rs:11-31) unrolled into a throughput product
//   accumulate  `colour += cast_ray(..)` (src/main.rs:136)
//   quantize    src/main.rs:138-142 + colour_to_raw (:172-177)
//
// Compiled with -ffp-contract=off: rustc never contracts a*b+c, and the f64 instantiation
// follows the reference's operation order so that it walks the same paths as the oracle.
//
// The file is compiled once per RT_PART (0-5: extend / bounce / the rest, for double and for float; Makefile)
// so that the six groups of kernel instantiations build in parallel; without RT_PART everything is one unit.
#include <type_traits>
#include "rt_device.h"

#include "../../../include/rt_mi355x.h"

#ifdef RT_PART
#define RT_PART_HAS(n) (RT_PART == (n))
#else
#define RT_PART_HAS(n) 1
#endif

namespace rtd {

// ------------------------------------------------------------------ small vector algebra
template <typename T>
struct V3 {
    T x, y, z;
};
template <typename T> __device__ __forceinline__ V3<T> mk(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3<T> ld3(const T* p) { return V3<T>{p[0], p[1], p[2]}; }
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return V3<T>{a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return V3<T>{a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a) { return V3<T>{-a.x, -a.y, -a.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(V3<T> a, T s) { return V3<T>{a.x * s, a.y * s, a.z * s}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return V3<T>{a.x * s, a.y * s, a.z * s}; }
template <typename T> __device__ __forceinline__ V3<T> operator/(V3<T> a, T s) { return V3<T>{a.x / s, a.y / s, a.z / s}; }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> l, V3<T> r) {
    return V3<T>{l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x};
}
template <typename T> __device__ __forceinline__ V3<T> coeff(V3<T> a, V3<T> b) { return V3<T>{a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename T> __device__ __forceinline__ T length_squared(V3<T> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <typename T> __device__ __forceinline__ T length(V3<T> a) { return sqrt(length_squared(a)); }
template <typename T> __device__ __forceinline__ V3<T> unit_vector(V3<T> a) {  // src/math.rs:34-40
    T l = length(a);
    return V3<T>{a.x / l, a.y / l, a.z / l};
}
template <typename T> __device__ __forceinline__ T comp(V3<T> a, int i) { return i == 0 ? a.x : i == 1 ? a.y : a.z; }
template <typename T> __device__ __forceinline__ bool near_zero(V3<T> a) {  // src/math.rs:41-44
    const T d = (T)1e-8;
    return fabs(a.x) < d && fabs(a.y) < d && fabs(a.z) < d;
}
template <typename T> __device__ __forceinline__ V3<T> reflect(V3<T> d, V3<T> n) {  // src/math.rs:67-69
    return d - ((T)2.0 * dot(d, n)) * n;
}
template <typename T> __device__ __forceinline__ V3<T> refract(V3<T> d, V3<T> n, T eta) {  // src/math.rs:71-78
    d = unit_vector(d);
    T cos_theta = fmin(dot(-d, n), (T)1.0);
    V3<T> perp = eta * (d + cos_theta * n);
    V3<T> par = (-(sqrt(fabs((T)1.0 - length_squared(perp))))) * n;
    return perp + par;
}
template <typename T> struct Lim;
template <> struct Lim<double> { static __device__ __forceinline__ double inf() { return __longlong_as_double(0x7ff0000000000000LL); } };
template <> struct Lim<float> { static __device__ __forceinline__ float inf() { return __int_as_float(0x7f800000); } };

template <typename T>
struct Ray {
    V3<T> o, d;
    T time;
    __device__ __forceinline__ V3<T> at(T t) const { return o + t * d; }  // src/math.rs:215-217
};

// ------------------------------------------------------------------ random streams
// Philox4x32-10, keyed as documented in DESIGN.md ("RNG") - the same specification the
// oracle implements in oracle/philox.hpp (written independently here).
enum : uint32_t { DOMAIN_CAMERA = 0u, DOMAIN_MATERIAL = 1u, DOMAIN_MEDIUM = 0x80000000u };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one v_mad_u64_u32 per product (hi and lo together) instead of v_mul_hi + v_mul_lo
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename T> __device__ __forceinline__ T bits_to_unit(uint64_t x);
template <> __device__ __forceinline__ double bits_to_unit<double>(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }
template <> __device__ __forceinline__ float bits_to_unit<float>(uint64_t x) { return (float)(x >> 40) * 0x1.0p-24f; }

template <typename T>
struct Rng {
    uint32_t k0, k1, pixel, sample, bounce, block;
    T spare;
    int have;
    __device__ __forceinline__ Rng(uint64_t seed, uint32_t domain, uint32_t px, uint32_t smp, uint32_t b)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32) ^ domain), pixel(px), sample(smp), bounce(b), block(0),
          spare(0), have(0) {}
    __device__ __forceinline__ T next() {  // uniform in [0, 1)
        if (have) {
            have = 0;
            return spare;
        }
        uint32_t r[4];
        philox4x32_10(pixel, sample, bounce, block, k0, k1, r);
        block++;
        spare = bits_to_unit<T>((uint64_t)r[2] | ((uint64_t)r[3] << 32));
        have = 1;
        return bits_to_unit<T>((uint64_t)r[0] | ((uint64_t)r[1] << 32));
    }
    __device__ __forceinline__ T range(T lo, T hi) { return lo + (hi - lo) * next(); }
};

template <typename T> __device__ V3<T> random_in_unit_sphere(Rng<T>& rng) {  // src/math.rs:234-246
    for (;;) {
        T px = rng.range((T)-1.0, (T)1.0);
        T py = rng.range((T)-1.0, (T)1.0);
        T pz = rng.range((T)-1.0, (T)1.0);
        V3<T> p = mk(px, py, pz);
        if (length_squared(p) <= (T)1.0) return p;
    }
}
template <typename T> __device__ V3<T> random_in_unit_disc(Rng<T>& rng) {  // src/math.rs:248-256
    for (;;) {
        T px = rng.range((T)-1.0, (T)1.0);
        T py = rng.range((T)-1.0, (T)1.0);
        V3<T> p = mk(px, py, (T)0.0);
        if (length_squared(p) <= (T)1.0) return p;
    }
}

// Try number `tr` of random_in_unit_sphere on the stream (key, pixel, sample, bounce) taken from its start: the
// stream's numbers 3 tr, 3 tr + 1, 3 tr + 2 (number i is half i & 1 of block i >> 1), each through range(-1, 1).
template <typename T>
__device__ __forceinline__ V3<T> unit_sphere_try(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t bounce,
                                                 uint32_t tr, bool* inside) {
    const uint32_t first = 3u * tr, block = first >> 1;
    uint32_t a[4], b[4];
    philox4x32_10(pixel, sample, bounce, block, k0, k1, a);
    philox4x32_10(pixel, sample, bounce, block + 1u, k0, k1, b);
    const uint64_t a0 = (uint64_t)a[0] | ((uint64_t)a[1] << 32), a1 = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    const uint64_t b0 = (uint64_t)b[0] | ((uint64_t)b[1] << 32), b1 = (uint64_t)b[2] | ((uint64_t)b[3] << 32);
    const bool odd = (first & 1u) != 0;
    const T u0 = bits_to_unit<T>(odd ? a1 : a0), u1 = bits_to_unit<T>(odd ? b0 : a1), u2 = bits_to_unit<T>(odd ? b1 : b0);
    const T lo = (T)-1.0, hi = (T)1.0;
    const V3<T> p = mk(lo + (hi - lo) * u0, lo + (hi - lo) * u1, lo + (hi - lo) * u2);
    *inside = length_squared(p) <= (T)1.0;
    return p;
}

// random_in_unit_sphere for the lanes of a wave that `want` one, each on its own fresh stream (key, pixel, sample,
// bounce): the same point as the loop above returns - the first try that falls inside - found by the whole wave.
// A loop per lane runs as long as the unluckiest of 64 lanes (6.4 tries on average where every lane draws, 1.9 are
// what a lane needs; cornell: 16 % of its k_bounce).  Here every round shares the wave out among the lanes still
// without a point: R of them get 64 / R lanes each (a power of two), which evaluate that many consecutive tries at
// once; the first one inside wins.  Tries are independent of each other (counter-based streams), so this is the
// same arithmetic on the same numbers.  Rounds: ~3.3 on average where every lane draws, 1 for a lone lane.
// EVERY lane of the wave must call this (want = false for those that need nothing).
// UNIFORM_BOUNCE: `bounce` is the same in every lane (a launch per bounce).
template <typename T, bool UNIFORM_BOUNCE>
__device__ __forceinline__ V3<T> wave_first_in_unit_sphere(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample,
                                                           uint32_t bounce, bool want) {
    const int lane = (int)(threadIdx.x & 63u);
    V3<T> res = mk((T)0.0, (T)0.0, (T)0.0);
    bool pending = want;
    uint32_t next_try = 0;  // (wave-uniform: every pending lane has failed the same tries)
    for (;;) {
        const uint64_t P = __ballot(pending);
        if (P == 0) break;
        const int R = __popcll(P);
        const int lg = 6 - (32 - __clz(R - 1));  // 64 >> ceil(log2 R) lanes per pending path
        if (lg == 0) {                           // more than 32 pending: a try each, by themselves
            if (pending) {
                bool in;
                const V3<T> p = unit_sphere_try<T>(k0, k1, pixel, sample, bounce, next_try, &in);
                if (in) {
                    res = p;
                    pending = false;
                }
            }
            next_try += 1u;
            continue;
        }
        const int g = 1 << lg;
        // slot j (lanes j g .. j g + g - 1) serves the j-th pending lane: a permutation of the lanes that puts the
        // pending ones first (one push through the LDS crossbar) tells slot j who that is
        const int below = (int)__popcll(P & ((1ull << lane) - 1ull));
        const int place = pending ? below : R + (lane - below);
        const int by_rank = __builtin_amdgcn_ds_permute(place << 2, lane);
        const int slot = lane >> lg, sub = lane & (g - 1);
        const bool serve = slot < R;
        const int owner = __shfl(by_rank, serve ? slot : 0);
        const uint32_t o_pixel = (uint32_t)__shfl((int)pixel, owner), o_sample = (uint32_t)__shfl((int)sample, owner);
        const uint32_t o_bounce = UNIFORM_BOUNCE ? bounce : (uint32_t)__shfl((int)bounce, owner);
        bool in = false;
        V3<T> p = mk((T)0.0, (T)0.0, (T)0.0);
        if (serve) p = unit_sphere_try<T>(k0, k1, o_pixel, o_sample, o_bounce, next_try + (uint32_t)sub, &in);
        const uint64_t A = __ballot(in);
        // the first try inside among the owner's slot
        const uint64_t mine = (A >> ((below << lg) & 63)) & (g == 64 ? ~0ull : ((1ull << g) - 1ull));
        const bool found = pending && mine != 0;
        const int from = found ? (below << lg) + (__ffsll((long long)mine) - 1) : lane;
        const V3<T> got = mk(__shfl(p.x, from), __shfl(p.y, from), __shfl(p.z, from));
        if (found) {
            res = got;
            pending = false;
        }
        next_try += (uint32_t)g;
    }
    return res;
}

__device__ __forceinline__ void path_ids(const DBatch& B, uint32_t pid, uint32_t* pixel, uint32_t* sample) {
    if (B.trace_mode) {
        *pixel = B.s0 + pid;  // rt_trace: the ray's index in the caller's array
        *sample = 0;
    } else {
        uint32_t pl = pid % B.npix, sl = pid / B.npix;
        *pixel = B.j0 * B.width + pl;  // (j0 + pl / W) * W + pl % W
        *sample = B.s0 + sl;
    }
}

// ------------------------------------------------------------------ primitives
// Record loads.  UNIFORM = the index is the same for every lane of the wave (sweep traversal): the
// load goes through the constant address space, which the compiler turns into scalar loads
// (s_load_dwordx8/x16 through the scalar cache) - the record lands in SGPRs, costs no VGPRs and no
// vector-memory issue slot.  Valid because the scene is immutable while kernels run.
template <bool UNIFORM, typename P>
__device__ __forceinline__ P rec(const P* p) {
    if constexpr (UNIFORM) {
        static_assert(sizeof(P) % 4 == 0, "records are whole dwords");
        typedef const __attribute__((address_space(4))) uint32_t* const_dwords;
        const_dwords src = (const_dwords)(uintptr_t)p;
        P out;
        uint32_t* dst = (uint32_t*)&out;
#pragma unroll
        for (unsigned k = 0; k < sizeof(P) / 4; k++) dst[k] = src[k];
        return out;
    } else {
        return *p;
    }
}

template <typename T>
__device__ __forceinline__ Ray<T> to_object_space(const DXform<T> X, const Ray<T>& r) {
    // One rigid map for a whole chain of Translate/Rotate*::hit (src/transforms.rs:22-23,
    // 103-112, 194-203, 285-294); Ray::new re-normalises the direction (src/math.rs:208-214).
    Ray<T> q;
    q.o = mk(X.m[0] * r.o.x + X.m[1] * r.o.y + X.m[2] * r.o.z + X.b[0],
             X.m[3] * r.o.x + X.m[4] * r.o.y + X.m[5] * r.o.z + X.b[1],
             X.m[6] * r.o.x + X.m[7] * r.o.y + X.m[8] * r.o.z + X.b[2]);
    q.d = unit_vector(mk(X.m[0] * r.d.x + X.m[1] * r.d.y + X.m[2] * r.d.z,
                         X.m[3] * r.d.x + X.m[4] * r.d.y + X.m[5] * r.d.z,
                         X.m[6] * r.d.x + X.m[7] * r.d.y + X.m[8] * r.d.z));
    q.time = r.time;
    return q;
}
template <typename T>
__device__ __forceinline__ V3<T> rotate_to_world(const DXform<T>& X, V3<T> n) {  // m^T n
    return mk(X.m[0] * n.x + X.m[3] * n.y + X.m[6] * n.z, X.m[1] * n.x + X.m[4] * n.y + X.m[7] * n.z,
              X.m[2] * n.x + X.m[5] * n.y + X.m[8] * n.z);
}

template <typename T>
__device__ __forceinline__ bool distance_to_sphere(const Ray<T>& ray, V3<T> centre, T radius, T min_dist, T max_dist,
                                                   T* out) {  // src/math.rs:258-286
    V3<T> oc = ray.o - centre;
    T a = length_squared(ray.d);
    T half_b = dot(oc, ray.d);
    T c = length_squared(oc) - radius * radius;
    T discriminant = half_b * half_b - a * c;
    if (discriminant < (T)0.0) return false;
    T sqrt_d = sqrt(discriminant);
    T root = (-half_b - sqrt_d) / a;
    if (root < min_dist || root > max_dist) {
        root = (-half_b + sqrt_d) / a;
        if (root < min_dist || root > max_dist) return false;
    }
    *out = root;
    return true;
}

template <typename T>
__device__ __forceinline__ V3<T> moving_centre(const DMovingSphere<T> s, T time) {  // src/objects.rs:94-97
    return ld3(s.centre0) + ((time - s.time0) / (s.time1 - s.time0)) * (ld3(s.centre1) - ld3(s.centre0));
}

template <typename T>
__device__ __forceinline__ bool line_plane_collision(V3<T> la, V3<T> lab, V3<T> p0, V3<T> p01, V3<T> p02, V3<T>* out) {
    // src/math.rs:298-317 (absolute 1e-7 determinant gate)
    const T MARGIN = (T)0.0000001;
    V3<T> cross_p = cross(p01, p02);
    T determinant = dot(-lab, cross_p);
    if (determinant < -MARGIN || determinant > MARGIN) {
        V3<T> w = la - p0;
        T t = dot(cross_p, w) / determinant;
        T u = dot(cross(p02, -lab), w) / determinant;
        T v = dot(cross(-lab, p01), w) / determinant;
        *out = mk(t, u, v);
        return true;
    }
    return false;
}

// XYRect / XZRect / YZRect::hit, src/objects.rs:224-244, 288-308, 352-372.
template <typename T>
__device__ __forceinline__ bool rect_hit(const DRect<T> q, const Ray<T>& r, T min_dist, T max_dist, T* t_out) {
    int axis = q.axis;
    int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
    T t = (q.k - comp(r.o, axis)) / comp(r.d, axis);
    if (t < min_dist || t > max_dist) return false;
    T a = comp(r.o, ia) + t * comp(r.d, ia);
    T b = comp(r.o, ib) + t * comp(r.d, ib);
    if (a < q.a0 || a > q.a1 || b < q.b0 || b > q.b1) return false;
    if (t != t) return false;  // 0/0: declared deviation, NaN hits are rejected
    *t_out = t;
    return true;
}

// The same test for a rect whose record is wave-uniform (sweep traversal): the axis is then known
// per wave, so the ray components are picked at compile time (comp() costs 24 v_cndmask per f64
// rect test) and the comparisons of rect_hit are evaluated without branches.  NaN behaves as in
// rect_hit: every `<` / `>` is false and `t != t` rejects.
template <int I, typename T> __device__ __forceinline__ T cget(const V3<T>& a) { return I == 0 ? a.x : I == 1 ? a.y : a.z; }

template <typename T, int AXIS>
__device__ __forceinline__ void rect_axis_test(const DRect<T>& q, int index, const Ray<T>& r, T min_dist, T max_dist,
                                               bool& have, T& bt, int& bi) {
    constexpr int IA = AXIS == 0 ? 1 : 0, IB = AXIS == 2 ? 1 : 2;
    const T t = (q.k - cget<AXIS>(r.o)) / cget<AXIS>(r.d);
    const T a = cget<IA>(r.o) + t * cget<IA>(r.d);
    const T b = cget<IB>(r.o) + t * cget<IB>(r.d);
    const bool miss = (t < min_dist) | (t > max_dist) | (a < q.a0) | (a > q.a1) | (b < q.b0) | (b > q.b1) | (t != t);
    const bool better = !miss & (!have | (t < bt));
    bt = better ? t : bt;
    bi = better ? index : bi;
    have = have | !miss;
}

template <typename T>
__device__ __forceinline__ bool rects_hit_uniform(const DScene<T>& S, const Ray<T>& r, int first, int count, T min_dist,
                                                  T max_dist, T* t_out, int* prim_out) {
    bool have = false;
    T bt = 0;
    int bi = -1;
    for (int k = 0; k < count; k++) {
        const DRect<T> q = rec<true>(&S.rects[first + k]);
        switch (q.axis) {  // a scalar branch
            case 0: rect_axis_test<T, 0>(q, first + k, r, min_dist, max_dist, have, bt, bi); break;
            case 1: rect_axis_test<T, 1>(q, first + k, r, min_dist, max_dist, have, bt, bi); break;
            default: rect_axis_test<T, 2>(q, first + k, r, min_dist, max_dist, have, bt, bi); break;
        }
    }
    *t_out = bt;
    *prim_out = bi;
    return have;
}

// A Block (src/objects.rs:149-168) is always the rect list z, z, y, y, x, x: with the axes known at
// compile time the six tests are straight-line code (no per-lane component selects, no branches).
// Same list rule as rects_hit: every side sees the same [min,max], strict < keeps the first.
template <typename T, int AXIS>
__device__ __forceinline__ DRect<T> block_face(const DBlock<T>& b, bool far_side) {  // what rect_axis_test reads of a side
    constexpr int IA = AXIS == 0 ? 1 : 0, IB = AXIS == 2 ? 1 : 2;
    DRect<T> q;
    q.a0 = b.lo[IA]; q.a1 = b.hi[IA]; q.b0 = b.lo[IB]; q.b1 = b.hi[IB];
    q.k = far_side ? b.hi[AXIS] : b.lo[AXIS];
    return q;
}
template <typename T>
__device__ __forceinline__ bool block6_hit(const DScene<T>& S, const Ray<T>& r, int first, int block, T min_dist, T max_dist,
                                           T* t_out, int* prim_out) {
    // the six rects of the list are the faces of the box with these corners (checked when the scene is
    // uploaded: same doubles, same comparisons), so 48 bytes stand for six 64-byte records
    const DBlock<T> b = S.blocks[block];
    bool have = false;
    T bt = 0;
    int bi = -1;
    rect_axis_test<T, 2>(block_face<T, 2>(b, false), first + 0, r, min_dist, max_dist, have, bt, bi);
    rect_axis_test<T, 2>(block_face<T, 2>(b, true), first + 1, r, min_dist, max_dist, have, bt, bi);
    rect_axis_test<T, 1>(block_face<T, 1>(b, false), first + 2, r, min_dist, max_dist, have, bt, bi);
    rect_axis_test<T, 1>(block_face<T, 1>(b, true), first + 3, r, min_dist, max_dist, have, bt, bi);
    rect_axis_test<T, 0>(block_face<T, 0>(b, false), first + 4, r, min_dist, max_dist, have, bt, bi);
    rect_axis_test<T, 0>(block_face<T, 0>(b, true), first + 5, r, min_dist, max_dist, have, bt, bi);
    *t_out = bt;
    *prim_out = bi;
    return have;
}

// Vec<Arc<dyn Hittable>>::hit over a Block's sides / a Spotlight's panes (src/hitting.rs:88-104):
// every element sees the same [min,max]; strict < keeps the first of equal distances.
template <typename T, bool UNIFORM>
__device__ __forceinline__ bool rects_hit(const DScene<T>& S, const Ray<T>& r, int first, int count, T min_dist,
                                          T max_dist, T* t_out, int* prim_out) {
    bool have = false;
    T bt = 0;
    int bi = -1;
    for (int k = 0; k < count; k++) {
        T t;
        if (rect_hit(rec<UNIFORM>(&S.rects[first + k]), r, min_dist, max_dist, &t)) {
            if (!have || t < bt) {
                bt = t;
                bi = first + k;
            }
            have = true;
        }
    }
    *t_out = bt;
    *prim_out = bi;
    return have;
}

template <typename T>
__device__ __forceinline__ bool triangle_hit(const DTriangle<T> tr, const Ray<T>& r, T min_dist, T max_dist, T* t_out) {
    // src/objects.rs:537-558
    V3<T> s;
    if (!line_plane_collision(r.o, r.d, ld3(tr.point), ld3(tr.vec1), ld3(tr.vec2), &s)) return false;
    if (s.x < min_dist || s.x > max_dist) return false;
    if (s.y < (T)0.0 || s.z < (T)0.0 || s.y + s.z > (T)1.0) return false;
    *t_out = s.x;
    return true;
}

// AABB::intersects, src/hitting.rs:249-268, with the divisions replaced by multiplications
// with 1/d (identical for the two planes of an axis, so a zero-thickness box still fails
// the `max <= min` test exactly as in the reference).  fmin/fmax ignore NaN as f64::min/max do.
template <typename T>
__device__ __forceinline__ bool box_hit(const DNode<T>& n, V3<T> o, V3<T> inv_d, T min_dist, T max_dist) {
    T ta = (n.lo[0] - o.x) * inv_d.x, tb = (n.hi[0] - o.x) * inv_d.x;
    min_dist = fmax(fmin(ta, tb), min_dist);
    max_dist = fmin(fmax(ta, tb), max_dist);
    if (max_dist <= min_dist) return false;
    ta = (n.lo[1] - o.y) * inv_d.y; tb = (n.hi[1] - o.y) * inv_d.y;
    min_dist = fmax(fmin(ta, tb), min_dist);
    max_dist = fmin(fmax(ta, tb), max_dist);
    if (max_dist <= min_dist) return false;
    ta = (n.lo[2] - o.z) * inv_d.z; tb = (n.hi[2] - o.z) * inv_d.z;
    min_dist = fmax(fmin(ta, tb), min_dist);
    max_dist = fmin(fmax(ta, tb), max_dist);
    if (max_dist <= min_dist) return false;
    return true;
}

// Conservative f32 slab test against a padded, outward-rounded child box (DNode2).
// The reference's AABB::intersects (src/hitting.rs:249-268) works in f64 on the exact box and
// rejects when max <= min; this test may only err towards "hit": the boxes are padded by far
// more than the f32 rounding of the test (tree_layout.hpp, box_pad), tmin is rounded down and the
// bound up, NaN (0 * inf) is ignored by fminf / fmaxf, and equality counts as a hit.  A node
// that is visited although the reference would have culled it costs time only: its primitives
// are tested exactly (f64) against the current bound.  Boxes the reference *always* culls
// (zero thickness) are removed when the layout is built (RT_EMPTY).
struct RayF {
    float ox, oy, oz, ix, iy, iz;
};
// `shift`: the box tests see the ray from the point o + shift d (far_shift below); 0 for every ray that starts near the scene.
template <typename T>
__device__ __forceinline__ RayF ray_f32(const Ray<T>& r, float shift) {
    RayF f;
    f.ox = (float)(r.o.x + (T)shift * r.d.x); f.oy = (float)(r.o.y + (T)shift * r.d.y); f.oz = (float)(r.o.z + (T)shift * r.d.z);
    f.ix = 1.0f / (float)r.d.x; f.iy = 1.0f / (float)r.d.y; f.iz = 1.0f / (float)r.d.z;
    return f;
}
// The f32 box tests round the ray's origin to f32: an error of |o| 2^-24, which the padding of the boxes (8e-6 of the
// scene's extent, tree_layout.hpp::box_pad) covers for origins within ~100 scene sizes.  A path scattered off an
// unbounded Plane can start much further out.  Such a ray is looked at from a point further along its line, where it
// comes within one radius of the bounding sphere of the world BVH (every box lies inside it; a ray that misses the
// sphere hits no box, whatever the tests say): distances along the ray are then measured from there (t - shift), in
// world space and - transforms are rigid, directions unit - in every object space alike.  Only box tests use it;
// primitives are tested exactly, on the ray as it is.
template <typename T>
__device__ __forceinline__ float far_shift(const DScene<T>& S, const Ray<T>& r) {
    const float m = fmaxf(fmaxf(fabsf((float)r.o.x), fabsf((float)r.o.y)), fabsf((float)r.o.z));
    if (m <= S.far_origin) return 0.0f;
    const T along = dot(ld3(S.bound_centre) - r.o, r.d) - S.bound_radius;
    return along > (T)0.0 ? (float)along : 0.0f;
}
__device__ __forceinline__ float round_down(double x) { return __double2float_rd(x); }
__device__ __forceinline__ float round_up(double x) { return __double2float_ru(x); }
__device__ __forceinline__ float round_down(float x) { return x; }
__device__ __forceinline__ float round_up(float x) { return x; }

__device__ __forceinline__ bool box2_hit(const float* lo, const float* hi, const RayF& f, float tmin, float tmax,
                                         float* entry) {
    float ta = (lo[0] - f.ox) * f.ix, tb = (hi[0] - f.ox) * f.ix;
    tmin = fmaxf(fminf(ta, tb), tmin);
    tmax = fminf(fmaxf(ta, tb), tmax);
    ta = (lo[1] - f.oy) * f.iy; tb = (hi[1] - f.oy) * f.iy;
    tmin = fmaxf(fminf(ta, tb), tmin);
    tmax = fminf(fmaxf(ta, tb), tmax);
    ta = (lo[2] - f.oz) * f.iz; tb = (hi[2] - f.oz) * f.iz;
    tmin = fmaxf(fminf(ta, tb), tmin);
    tmax = fminf(fmaxf(ta, tb), tmax);
    *entry = tmin;  // where the ray enters the (padded) box: only used to pick which child to walk first
    return tmax >= tmin;
}

struct alignas(16) Node2Regs {  // a DNode2 fetched as four 16-byte loads
    uint4 q[4];
    __device__ __forceinline__ const float* lo(int c) const { return (const float*)q + 3 * c; }
    __device__ __forceinline__ const float* hi(int c) const { return (const float*)q + 6 + 3 * c; }
    __device__ __forceinline__ int32_t child(int c) const { return ((const int32_t*)q)[12 + c]; }
};
__device__ __forceinline__ Node2Regs load_node2(const DNode2* nodes, int idx) {
    const uint4* p = (const uint4*)(nodes + idx);
    Node2Regs n;
    n.q[0] = p[0]; n.q[1] = p[1]; n.q[2] = p[2]; n.q[3] = p[3];
    return n;
}

// Per-lane traversal stack in LDS: entry e of lane l lives at stack[e * BLOCK + l], so the
// 64 lanes of a wave touch 64 consecutive dwords (no bank conflicts).
struct Stack {
    static constexpr bool COOP = false;
    int32_t* base;
    int stride;
    __device__ __forceinline__ void put(int e, int32_t v) { base[e * stride] = v; }
    __device__ __forceinline__ int32_t get(int e) const { return base[e * stride]; }
    __device__ __forceinline__ bool room(int, int) const { return false; }  // (depth unknown here: checked stores only)
    __device__ __forceinline__ void put_near(int e, int32_t v) { base[e * stride] = v; }
};
// The same with a private array behind it for entries beyond `cap`: a four-wide walk keeps up to 3
// siblings per level on the stack, which in its worst visiting order is more than the LDS part holds.
struct DeepStack {
    static constexpr bool COOP = false;
    int32_t* base;
    int stride;
    int cap;
    int32_t* overflow;
    __device__ __forceinline__ void put(int e, int32_t v) {
        if (e < cap) base[e * stride] = v;
        else overflow[e - cap] = v;
    }
    __device__ __forceinline__ int32_t get(int e) const { return e < cap ? base[e * stride] : overflow[e - cap]; }
    __device__ __forceinline__ bool room(int e, int n) const { return e + n <= cap; }  // entries e .. e+n-1 are in LDS
    __device__ __forceinline__ void put_near(int e, int32_t v) { base[e * stride] = v; }
};
// A DeepStack held by each lane of a GROUP of `group` neighbouring lanes (a power of two, aligned) that carry the
// same ray: the walk down to a mesh is the same in every lane, and the mesh walk itself is shared out over the
// group (blas_hit4_group), which uses the LDS rows above the caller's entries in all the group's columns.
struct GroupStack : DeepStack {
    static constexpr bool COOP = true;
    int group;
};

// BVHNode::hit over a triangle BVH (src/hitting.rs:211-229).  The reference's recursion "left with
// [min,max], then right with [min, min(left.t, max)], right wins if it hits" returns the leaf with
// the smallest t and, among leaves with exactly equal t, the one that comes LAST in depth-first
// (left-first) order.  That answer does not depend on the order in which leaves are tested, so the
// walk here goes into the nearer child first (the bound shrinks sooner and far subtrees are
// culled: the reference's fixed left-first order tests ~3x more triangles on the knight) and
// settles exact ties by the leaf's depth-first rank (DTriangle::dfs_rank, set when the scene is
// uploaded).  `root` is a DNode2 whose child 0 is the reference's root.
template <typename T, typename ST>
__device__ bool blas_hit(const DScene<T>& S, const Ray<T>& r, int root, T min_dist, T max_dist, ST st, int sp0, float shift,
                         T* t_out, int* prim_out) {
    const RayF f = ray_f32(r, shift);
    const float tmin32 = round_down(min_dist - (T)shift);
    float bound32 = round_up(max_dist - (T)shift);
    int sp = sp0;
    int32_t next = root;
    bool found = false;
    int32_t best_rank = -1;
    for (;;) {
        // One 64-byte fetch decides a level (both children's boxes); leaves that come up - as the
        // chosen child or from the stack - are tested exactly before the next node is fetched.
        const Node2Regs n = load_node2(S.nodes2, next);
        const int32_t c0 = n.child(0), c1 = n.child(1);
        float e0, e1;
        const bool h0 = c0 != RT_EMPTY && box2_hit(n.lo(0), n.hi(0), f, tmin32, bound32, &e0);
        const bool h1 = c1 != RT_EMPTY && box2_hit(n.lo(1), n.hi(1), f, tmin32, bound32, &e1);
        next = RT_EMPTY;
        if (h0 && h1) {
            const bool far_first = S.ordered && e1 < e0;
            next = far_first ? c1 : c0;
            st.put(sp++, far_first ? c0 : c1);
        } else if (h0) {
            next = c0;
        } else if (h1) {
            next = c1;
        }
        for (;;) {
            if (next == RT_EMPTY) {
                if (sp == sp0) {
                    *t_out = max_dist;
                    return found;
                }
                next = st.get(--sp);
            }
            if (next >= 0) break;
            const int tri = -1 - next;  // leaf: the exact test, against the current bound (t == bound passes)
            T t;
            const DTriangle<T> tr = S.triangles[tri];
            if (triangle_hit(tr, r, min_dist, max_dist, &t)) {
                if (!found || t < max_dist || tr.dfs_rank > best_rank) {
                    max_dist = t;
                    bound32 = round_up(t - (T)shift);
                    *prim_out = tri;
                    best_rank = tr.dfs_rank;
                    found = true;
                }
            }
            next = RT_EMPTY;
        }
    }
    *t_out = max_dist;
    return found;
}

struct alignas(16) Node4Regs {  // a DNode4 fetched as four 16-byte loads
    uint4 q[4];
    __device__ __forceinline__ float origin(int axis) const { return __uint_as_float(((const uint32_t*)q)[axis]); }
    __device__ __forceinline__ float step(int axis) const { return __uint_as_float(((const uint32_t*)q)[3 + axis]); }
    __device__ __forceinline__ uint32_t qlo(int axis) const { return ((const uint32_t*)q)[6 + axis]; }  // four slots, one byte each
    __device__ __forceinline__ uint32_t qhi(int axis) const { return ((const uint32_t*)q)[9 + axis]; }
    __device__ __forceinline__ int32_t child(int c) const { return ((const int32_t*)q)[12 + c]; }
};
__device__ __forceinline__ Node4Regs load_node4(const DNode4* nodes, int idx) {
    const uint4* p = (const uint4*)(nodes + idx);
    Node4Regs n;
#pragma unroll
    for (int k = 0; k < 4; k++) n.q[k] = p[k];
    return n;
}
__device__ __forceinline__ void order2(float& ka, int32_t& ca, float& kb, int32_t& cb) {  // compare-exchange
    const bool swap = kb < ka;
    const float k0 = swap ? kb : ka, k1 = swap ? ka : kb;
    const int32_t c0 = swap ? cb : ca, c1 = swap ? ca : cb;
    ka = k0; kb = k1; ca = c0; cb = c1;
}

// Measurement hooks (lane / wave step counts, live lanes of the tail launch, steps per ray): no-ops in the product
// build; a library built with -DRT_EXP=<bits> takes their definitions from rt_measure.inc (scripts/exp_*.py).
#ifdef RT_EXP
#include "rt_measure.inc"
#else
#define RT_STEP_COUNT
#define RT_MESH_LIVE
#define RT_PROF_DECL
#define RT_PROF(k)
#define RT_PROF_FLUSH
#define RT_EXTEND_STEPS_BEGIN
#define RT_EXTEND_STEPS_END(depth, d, r, lane)
#define RT_TAIL_WAVE_BEGIN
#define RT_TAIL_QUEUE_DRY(exhausted)
#define RT_TAIL_ITER_BEGIN(alive)
#define RT_TAIL_ITER_END
#define RT_TAIL_WAVE_END
#endif

// One step of a four-wide walk: the four child boxes of `n` against the ray, without branches - a
// plane's distance is fma(q, step / d, (origin - o) / d) with q the plane's 8-bit offset, one
// packed operation per pair of children; NaN (0 * inf on an axis the ray does not move along)
// leaves that axis out, like fminf / fmaxf in box2_hit - then the children that are hit ordered
// nearest first (or left to right), all but the first pushed - farthest first - and the first
// returned (RT_EMPTY: none was hit).
typedef float float2v __attribute__((ext_vector_type(2)));
// Entry / exit distances of the four child boxes of `n` (intersected with [tmin32, bound32]).
__device__ __forceinline__ void node4_boxes(const Node4Regs& n, const RayF& f, float tmin32, float bound32, float* near, float* far) {
    for (int k = 0; k < 4; k++) {
        near[k] = tmin32;
        far[k] = bound32;
    }
    const float o[3] = {f.ox, f.oy, f.oz}, inv[3] = {f.ix, f.iy, f.iz};
#pragma unroll
    for (int axis = 0; axis < 3; axis++) {
        const float a = n.step(axis) * inv[axis], b = (n.origin(axis) - o[axis]) * inv[axis];
        const float2v a2 = {a, a}, b2 = {b, b};
        const uint32_t ql = n.qlo(axis), qh = n.qhi(axis);
#pragma unroll
        for (int pair = 0; pair < 2; pair++) {
            const float2v l2 = {(float)((ql >> (16 * pair)) & 0xffu), (float)((ql >> (16 * pair + 8)) & 0xffu)};
            const float2v h2 = {(float)((qh >> (16 * pair)) & 0xffu), (float)((qh >> (16 * pair + 8)) & 0xffu)};
            const float2v ta = __builtin_elementwise_fma(l2, a2, b2), tb = __builtin_elementwise_fma(h2, a2, b2);
            near[2 * pair] = fmaxf(fminf(ta.x, tb.x), near[2 * pair]);
            far[2 * pair] = fminf(fmaxf(ta.x, tb.x), far[2 * pair]);
            near[2 * pair + 1] = fmaxf(fminf(ta.y, tb.y), near[2 * pair + 1]);
            far[2 * pair + 1] = fminf(fmaxf(ta.y, tb.y), far[2 * pair + 1]);
        }
    }
}
// FULL_SORT = false: only the nearest child is singled out, the others are pushed as they come (10 instructions
// fewer a step).  Measured: world trees of Blocks win (blocks_ball extend 13.3 -> 12.6 ms: a step saved matters
// more than the order of the far children, whose object tests mostly still run), meshes lose (knight 34.2 -> 35.2).
template <typename ST, bool FULL_SORT = true>
__device__ __forceinline__ int32_t node4_step(const Node4Regs& n, const RayF& f, float tmin32, float bound32, bool ordered,
                                              ST& st, int& sp) {
    const float MISS = __int_as_float(0x7f800000);
    float near[4], far[4];
    node4_boxes(n, f, tmin32, bound32, near, far);
    int32_t c0 = n.child(0), c1 = n.child(1), c2 = n.child(2), c3 = n.child(3);
    float k0 = ((c0 != RT_EMPTY) & (far[0] >= near[0])) ? (ordered ? near[0] : 0.0f) : MISS;
    float k1 = ((c1 != RT_EMPTY) & (far[1] >= near[1])) ? (ordered ? near[1] : 1.0f) : MISS;
    float k2 = ((c2 != RT_EMPTY) & (far[2] >= near[2])) ? (ordered ? near[2] : 2.0f) : MISS;
    float k3 = ((c3 != RT_EMPTY) & (far[3] >= near[3])) ? (ordered ? near[3] : 3.0f) : MISS;
    order2(k0, c0, k1, c1);
    order2(k2, c2, k3, c3);
    order2(k0, c0, k2, c2);
    if constexpr (FULL_SORT) {
        order2(k1, c1, k3, c3);
        order2(k1, c1, k2, c2);
    }
    if (st.room(sp, 3)) {  // the usual case: three stores, the stack pointer moves past those that count
        st.put_near(sp, c3);
        sp += k3 < MISS;
        st.put_near(sp, c2);
        sp += k2 < MISS;
        st.put_near(sp, c1);
        sp += k1 < MISS;
    } else {
        if (k3 < MISS) st.put(sp++, c3);
        if (k2 < MISS) st.put(sp++, c2);
        if (k1 < MISS) st.put(sp++, c1);
    }
    return k0 < MISS ? c0 : RT_EMPTY;
}


// The same walk as blas_hit over the four-wide tree: one 128-byte fetch decides two levels of the
// reference's tree, the children that are hit are visited nearest first (S.ordered) or in the
// reference's left-to-right order, exact ties go to the higher depth-first rank.
template <typename T, typename ST>
__device__ bool blas_hit4_lane(const DScene<T>& S, const Ray<T>& r, int root, T min_dist, T max_dist, ST st, int sp0, float shift,
                              T* t_out, int* prim_out) {
    const RayF f = ray_f32(r, shift);
    const float tmin32 = round_down(min_dist - (T)shift);
    float bound32 = round_up(max_dist - (T)shift);
    int sp = sp0;
    int32_t next = root;
    bool found = false;
    int32_t best_rank = -1;
    RT_PROF_DECL;
    const int vote = S.leaf_vote, frac = S.leaf_frac;
    for (;;) {
        if (next >= 0) {
            RT_PROF(0);
            RT_STEP_COUNT;
            RT_MESH_LIVE;
            const Node4Regs n = load_node4(S.nodes4, next);
            next = node4_step(n, f, tmin32, bound32, S.ordered != 0, st, sp);
            if (next == RT_EMPTY) {
                if (sp == sp0) break;
                next = st.get(--sp);
            }
        }
        // One step per round and lane: a node step above for the lanes that hold a node, one exact test below for those
        // that hold a leaf - never a loop over a lane's leaves while the node lanes of the wave wait (the walk used to
        // take the leaves off the stack and test them in a row until a node came up: knight 35.0 -> 30.0 ms without it).
        // On top, a vote: the tests run when S.leaf_frac sixteenths of the lanes still walking hold a leaf (at most
        // S.leaf_vote lanes), or none has a node left; a lane with a leaf in hand waits meanwhile.  A test is ~130 f64
        // instructions, a node step ~100 of f32.  knight at 0 (no waiting) / 3 / 5 / 8 / 12 / 16 sixteenths: 30.0 /
        // 30.0 / 29.7 / 30.3 / 32.3 / 42.8 ms.  The answer does not depend on the order of the tests (smallest t,
        // among equal t the highest rank).
        const uint64_t with_leaf = __ballot(next < 0);
        if (with_leaf == 0) continue;
        {
            const uint64_t with_node = __ballot(next >= 0);
            const int n_leaf = __popcll(with_leaf), n_node = __popcll(with_node);
            const int need = frac > 0 ? min(vote, ((n_leaf + n_node) * frac + 15) >> 4) : vote;  // frac sixteenths of the lanes still walking
            if (n_leaf < need && n_node != 0) continue;
        }
        if (next < 0) {
            RT_PROF(2);
            RT_STEP_COUNT;
            const int tri = -1 - next;  // leaf: the exact test, against the current bound (t == bound passes)
            T t;
            const DTriangle<T> tr = S.triangles[tri];
            if (triangle_hit(tr, r, min_dist, max_dist, &t)) {
                if (!found || t < max_dist || tr.dfs_rank > best_rank) {
                    max_dist = t;
                    bound32 = round_up(t - (T)shift);
                    *prim_out = tri;
                    best_rank = tr.dfs_rank;
                    found = true;
                }
            }
            if (sp == sp0) break;
            next = st.get(--sp);
        }
    }
    RT_PROF_FLUSH;
    *t_out = max_dist;
    return found;
}

// The same query answered by a group of lanes that all hold the same ray (GroupStack): the nodes and leaves that
// are still to be visited sit in a frontier in LDS, and every round the group takes up to `group` of them off its
// end, one per lane: a node's lane tests its four boxes and appends the children that are hit, a leaf's lane runs
// the exact triangle test.  The answer is the one of the one-lane walk, bit for bit: that is "the smallest t, among
// equal t the highest depth-first rank" over the leaves the ray hits inside [min, max], whatever the order of the
// visits (see blas_hit); culling by the current bound only ever drops subtrees that hold no such leaf (t == bound
// passes the box test, which is conservative).  A path's tail of bounces is walked by one lane per wave at the
// end of the tail launch: with 64 lanes per ray the chain of dependent fetches is as long as the tree is deep
// instead of as long as the walk.
template <typename T, typename ST>
__device__ bool blas_hit4_group(const DScene<T>& S, const Ray<T>& r, int root, T min_dist, T max_dist, ST st, int sp0, float shift,
                               T* t_out, int* prim_out) {
    const int G = st.group;
    const int lane = threadIdx.x & 63, gl = lane & (G - 1);
    const uint64_t gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane - gl);
    const uint64_t below = gmask & ((1ull << lane) - 1ull);  // the lanes of my group before me
    const int fcap = (st.cap - sp0) * G;                     // frontier entries the group's free LDS rows hold
    // (entry e: row sp0 + e / G of the column of the group's lane e % G)
    auto slot = [&](int e) -> int32_t* { return st.base + (sp0 + e / G) * st.stride + ((e & (G - 1)) - gl); };
    // A frontier that could outgrow its rows is walked depth first by one entry a round: that needs no more than the
    // one-lane walk does (S.coop_reserve entries: 3 per level of the deepest four-wide tree of the scene).
    const int reserve = S.coop_reserve;
    if (fcap < 2 * reserve + 4 * G) return blas_hit4_lane(S, r, root, min_dist, max_dist, (DeepStack)st, sp0, shift, t_out, prim_out);
    const RayF f = ray_f32(r, shift);
    const float tmin32 = round_down(min_dist - (T)shift);
    float bound32 = round_up(max_dist - (T)shift);
    const float MISS = __int_as_float(0x7f800000);
    int count = 1;
    if (gl == 0) *slot(0) = root;
    bool found = false;
    int32_t best_rank = -1;
    int best_prim = -1;
    while (count > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int n = count < G ? count : G;
        if (count + 3 * n > fcap - reserve) n = 1;
        const bool active = gl < n;
        const int32_t ref = active ? *slot(count - n + gl) : RT_EMPTY;
        count -= n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int32_t c[4] = {RT_EMPTY, RT_EMPTY, RT_EMPTY, RT_EMPTY};
        float key[4] = {MISS, MISS, MISS, MISS};
        T t = max_dist;
        int32_t rank = -1;
        int prim = -1;
        bool hit = false;
        if (ref >= 0) {
            const Node4Regs nd = load_node4(S.nodes4, ref);
            float near[4], far[4];
            node4_boxes(nd, f, tmin32, bound32, near, far);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[k] = nd.child(k);
                key[k] = ((c[k] != RT_EMPTY) & (far[k] >= near[k])) ? near[k] : MISS;
            }
            // farthest first: the nearest boxes end up at the frontier's end and are taken first
            order2(key[0], c[0], key[1], c[1]);
            order2(key[2], c[2], key[3], c[3]);
            order2(key[0], c[0], key[2], c[2]);
            order2(key[1], c[1], key[3], c[3]);
            order2(key[1], c[1], key[2], c[2]);
        } else if (ref != RT_EMPTY) {
            const int tri = -1 - ref;
            const DTriangle<T> tr = S.triangles[tri];
            T th;
            if (triangle_hit(tr, r, min_dist, max_dist, &th)) {
                hit = true;
                t = th;
                rank = tr.dfs_rank;
                prim = tri;
            }
        }
        // the round's hits, one after the other (a round rarely has more than one or two; most have none), each
        // under the rule of the one-lane walk: nearer wins, the higher rank among equal distances
        for (uint64_t hits = __ballot(hit) & gmask; hits; hits &= hits - 1) {
            const int from = __ffsll((long long)hits) - 1;
            const T o_t = __shfl(t, from);
            const int32_t o_rank = __shfl(rank, from);
            const int o_prim = __shfl(prim, from);
            if (o_t <= max_dist && (!found || o_t < max_dist || o_rank > best_rank)) {  // (the bound may have moved since the test)
                max_dist = o_t;
                bound32 = round_up(o_t - (T)shift);
                best_prim = o_prim;
                best_rank = o_rank;
                found = true;
            }
        }
        // append the children that are still inside the bound (slot 3 holds the farthest of a node)
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const bool push = key[k] < MISS && key[k] <= bound32;
            const uint64_t m = __ballot(push) & gmask;
            if (push) *slot(count + __popcll(m & below)) = c[k];
            count += __popcll(m);
        }
    }
    *t_out = max_dist;
    if (found) *prim_out = best_prim;
    return found;
}

template <typename T, typename ST>
__device__ __forceinline__ bool blas_hit4(const DScene<T>& S, const Ray<T>& r, int root, T min_dist, T max_dist, ST st, int sp0,
                                          float shift, T* t_out, int* prim_out) {
    if constexpr (ST::COOP) return blas_hit4_group(S, r, root, min_dist, max_dist, st, sp0, shift, t_out, prim_out);
    else return blas_hit4_lane(S, r, root, min_dist, max_dist, st, sp0, shift, t_out, prim_out);
}

struct Winner {
    int obj;   // rt_object index of the instance that was hit (-1: none)
    int prim;  // index into the array of the object's kind (rect / triangle / ... / medium)
};

// Hittable::hit of one non-medium object instance with bounds [min_dist, max_dist].
// FEAT: which object kinds a kernel variant compiles in besides rect lists (scenes made of walls,
// boxes and spheres get a much smaller kernel that needs no scratch memory).
enum { FEAT_SPHERES = 1, FEAT_FLAT = 2 /* Triangle / Plane objects */, FEAT_BLAS = 4, FEAT_MEDIA = 8, FEAT_ALL = 15,
       FEAT_BLAS4 = 16 /* meshes are walked through the four-wide tree */,
       FEAT_NESTED_MEDIA = 32 /* a medium's boundary may itself be a medium (one level) */ };

template <typename T, bool UNIFORM, int FEAT, typename ST>
__device__ bool simple_object_hit(const DScene<T>& S, int oi, const Ray<T>& world_ray, T min_dist, T max_dist,
                                  ST st, int sp, float shift, T* t_out, int* prim_out) {
    const DObject ob = rec<UNIFORM>(&S.objects[oi]);
    Ray<T> r = world_ray;
    if (ob.xform >= 0) r = to_object_space(rec<UNIFORM>(&S.xforms[ob.xform]), world_ray);
    if (ob.kind == RT_OBJ_RECTS) {
        if (!UNIFORM && ob.block6) return block6_hit(S, r, ob.first, ob.block6 - 1, min_dist, max_dist, t_out, prim_out);
        return rects_hit<T, UNIFORM>(S, r, ob.first, ob.count, min_dist, max_dist, t_out, prim_out);
    }
    if constexpr ((FEAT & FEAT_SPHERES) != 0) {
        if (ob.kind == RT_OBJ_SPHERE) {
            const DSphere<T> s = rec<UNIFORM>(&S.spheres[ob.first]);
            *prim_out = ob.first;
            return distance_to_sphere(r, ld3(s.centre), s.radius, min_dist, max_dist, t_out);
        }
        if (ob.kind == RT_OBJ_MOVING_SPHERE) {
            const DMovingSphere<T> s = rec<UNIFORM>(&S.moving_spheres[ob.first]);
            *prim_out = ob.first;
            return distance_to_sphere(r, moving_centre(s, r.time), s.radius, min_dist, max_dist, t_out);
        }
    }
    if constexpr ((FEAT & FEAT_FLAT) != 0) {
        if (ob.kind == RT_OBJ_TRIANGLE) {
            *prim_out = ob.first;
            return triangle_hit(rec<UNIFORM>(&S.triangles[ob.first]), r, min_dist, max_dist, t_out);
        }
        if (ob.kind == RT_OBJ_PLANE) {  // src/objects.rs:599-606
            const DPlane<T> p = rec<UNIFORM>(&S.planes[ob.first]);
            V3<T> s;
            if (!line_plane_collision(r.o, r.d, ld3(p.point), ld3(p.vec1), ld3(p.vec2), &s)) return false;
            if (s.x < min_dist || s.x > max_dist) return false;
            *t_out = s.x;
            *prim_out = ob.first;
            return true;
        }
    }
    if constexpr ((FEAT & FEAT_BLAS) != 0) {
        if (ob.kind == RT_OBJ_BLAS) {
            if constexpr ((FEAT & FEAT_BLAS4) != 0) return blas_hit4(S, r, ob.first4, min_dist, max_dist, st, sp, shift, t_out, prim_out);
            else return blas_hit(S, r, ob.first, min_dist, max_dist, st, sp, shift, t_out, prim_out);
        }
    }
    return false;
}

// The stream of a path's draws is named by (pixel, sample); both come from the path id by an
// integer division, which only the rare constant-medium draw needs: computed there.
struct PathRef {
    // What names a path's random streams, by value: holding a pointer to the kernel's DBatch argument instead made the
    // compiler keep that argument in scratch memory (its address escaped into the out-of-line draw below), and with it
    // everything derived from the seed - the ten round keys of every Philox block - lived in per-lane registers.
    uint64_t seed;
    uint32_t npix, width, j0, s0, trace_mode;
    uint32_t pid;
};
__device__ __forceinline__ PathRef path_of(const DBatch& B, uint32_t pid) {
    return PathRef{B.seed, B.npix, B.width, B.j0, B.s0, B.trace_mode, pid};
}

// ln(xi) of the constant-medium draw (src/objects.rs:428-429).  Out of line: f64 log and the stream
// set-up are rare and would otherwise keep their constants live across the whole traversal loop.
template <typename T>
__device__ __noinline__ T medium_log_draw(PathRef path, uint32_t rng_key, uint32_t depth) {
    uint32_t pixel, sample;  // (path_ids, from the fields the reference carries)
    if (path.trace_mode) {
        pixel = path.s0 + path.pid;
        sample = 0;
    } else {
        const uint32_t pl = path.pid % path.npix, sl = path.pid / path.npix;
        pixel = path.j0 * path.width + pl;
        sample = path.s0 + sl;
    }
    Rng<T> rng(path.seed, DOMAIN_MEDIUM | rng_key, pixel, sample, depth);
    return log(rng.next());
}

#define RT_MEDIUM_ATTR  // (out of line it cost 70 %: the call's own saves and the lost scalarisation)
template <typename T, bool UNIFORM, int FEAT, typename ST>
__device__ RT_MEDIUM_ATTR bool medium_hit(const DScene<T>& S, const DObject ob, const Ray<T>& world_ray, T min_dist, T max_dist, ST st,
                                          int sp, float shift, PathRef path, uint32_t depth, T* t_out, int* prim_out);
template <typename T, bool UNIFORM, int FEAT, typename ST>
__device__ bool object_hit(const DScene<T>& S, int oi, const Ray<T>& world_ray, T min_dist, T max_dist, ST st,
                           int sp, float shift, PathRef path, uint32_t depth, T* t_out, int* prim_out) {
    const DObject ob = rec<UNIFORM>(&S.objects[oi]);
    if ((FEAT & FEAT_MEDIA) == 0 || ob.kind != RT_OBJ_MEDIUM)
        return simple_object_hit<T, UNIFORM, FEAT>(S, oi, world_ray, min_dist, max_dist, st, sp, shift, t_out, prim_out);
    return medium_hit<T, UNIFORM, FEAT>(S, ob, world_ray, min_dist, max_dist, st, sp, shift, path, depth, t_out, prim_out);
}
// boundary.hit(ray, lo, hi) of a ConstantMedium (src/objects.rs:404-409).  The boundary is any Hittable: where the
// kernel is compiled for it (FEAT_NESTED_MEDIA: the general four-wide kernel and rt_trace's second walk) it may be another
// medium, whose own hit - two queries of ITS boundary, its own draw - answers the query; that inner medium's
// boundary is not a medium (checked when the scene is loaded / uploaded).
template <typename T, bool UNIFORM, int FEAT, typename ST>
__device__ __forceinline__ bool boundary_hit(const DScene<T>& S, int bo, const Ray<T>& r, T lo, T hi, ST st, int sp, float shift, PathRef path,
                                             uint32_t depth, T* t_out) {
    int ignore;
    if constexpr ((FEAT & FEAT_NESTED_MEDIA) != 0) {
        const DObject bob = rec<UNIFORM>(&S.objects[bo]);
        if (bob.kind == RT_OBJ_MEDIUM)
            return medium_hit<T, UNIFORM, (FEAT & ~FEAT_NESTED_MEDIA)>(S, bob, r, lo, hi, st, sp, shift, path, depth, t_out, &ignore);
    }
    return simple_object_hit<T, UNIFORM, FEAT>(S, bo, r, lo, hi, st, sp, shift, t_out, &ignore);
}
// ConstantMedium::hit, src/objects.rs:398-441.
template <typename T, bool UNIFORM, int FEAT, typename ST>
__device__ RT_MEDIUM_ATTR bool medium_hit(const DScene<T>& S, const DObject ob, const Ray<T>& world_ray, T min_dist, T max_dist, ST st,
                                          int sp, float shift, PathRef path, uint32_t depth, T* t_out, int* prim_out) {
    const DMedium<T> md = rec<UNIFORM>(&S.media[ob.first]);
    Ray<T> r = world_ray;
    if (ob.xform >= 0) r = to_object_space(rec<UNIFORM>(&S.xforms[ob.xform]), world_ray);
    const T inf = Lim<T>::inf();
    T t1, t2;
    // (the second query runs only in the lanes whose first one hit, so it is not wave-uniform)
    if (!boundary_hit<T, UNIFORM, FEAT>(S, md.boundary_object, r, -inf, inf, st, sp, shift, path, depth, &t1)) return false;
    T beyond = t1 + (T)0.0001;
    // (fast mode: 0.0001 is below f32 resolution from |t| ~ 800 on, and a query from t1 itself finds the first crossing again - the
    // mist of blocks_ball, whose boundary is 2000 units away, vanished for a third of the rays: at least two ulps beyond t1)
    if constexpr (sizeof(T) == 4) beyond = fmaxf(beyond, t1 + fabsf(t1) * 2.4e-7f);
    if (!boundary_hit<T, UNIFORM, FEAT>(S, md.boundary_object, r, beyond, inf, st, sp, shift, path, depth, &t2)) return false;
    if (t1 < min_dist) t1 = min_dist;
    if (t2 > max_dist) t2 = max_dist;
    if (t1 >= t2) return false;
    if (t1 < (T)0.0) t1 = (T)0.0;
    T ray_length = length(r.d);
    T distance_inside_boundary = (t2 - t1) * ray_length;
    T hit_distance = md.neg_inv_density * medium_log_draw<T>(path, md.rng_key, depth);
    if (hit_distance > distance_inside_boundary) return false;
    *t_out = t1 + hit_distance / ray_length;
    *prim_out = ob.first;
    return true;
}

// world.hit(ray, min, max): the top-level Vec (src/hitting.rs:188-194 -> :88-104) over the
// world BVH and the unbounded objects; inside the BVH, BVHNode::hit (:211-229).
enum { WALK_NODE2 = 0, WALK_OWN_BOX_LDS = 1, WALK_SWEEP = 2, WALK_SWEEP_LEAN = 3 /* rect lists and spheres only */,
       WALK_NODE2_MESH4 = 4 /* four-wide world and mesh trees, every object kind */,
       // the same walk compiled without the object kinds a scene does not hold (smaller kernels, fewer spills):
       WALK_NODE4_MESH = 5 /* no media */, WALK_NODE4_MEDIA = 6 /* no meshes */, WALK_NODE4_PLAIN = 7 /* neither */,
       WALK_SWEEP_FLAT = 8 /* a sweep of a scene without meshes or media: rect lists, spheres, planes, triangles */ };
constexpr bool walks_node4(int mode) { return mode >= WALK_NODE2_MESH4 && mode <= WALK_NODE4_PLAIN; }
template <int MODE> struct WalkFeat {
    static constexpr int value = MODE == WALK_SWEEP_LEAN     ? FEAT_SPHERES
                                 : MODE == WALK_SWEEP_FLAT   ? (FEAT_SPHERES | FEAT_FLAT)
                                 : MODE == WALK_NODE2_MESH4  ? (FEAT_ALL | FEAT_BLAS4 | FEAT_NESTED_MEDIA)
                                 : MODE == WALK_NODE4_MESH   ? ((FEAT_ALL & ~FEAT_MEDIA) | FEAT_BLAS4)
                                 : MODE == WALK_NODE4_MEDIA  ? (FEAT_ALL & ~FEAT_BLAS)
                                 : MODE == WALK_NODE4_PLAIN  ? (FEAT_ALL & ~FEAT_BLAS & ~FEAT_MEDIA)
                                                             : FEAT_ALL;
};

// The four-wide WORLD tree walked by a group of lanes that hold the same ray (worlds without meshes: a leaf is an
// object test that needs no stack): blas_hit4_group's frontier, with object_hit at the leaves and the objects'
// depth-first ranks for exact ties, as in the one-lane loop of closest_hit.  false: the frontier has too little
// room (the caller walks with one lane).
template <typename T, int FEAT, typename ST>
__device__ bool world_hit4_group(const DScene<T>& S, const Ray<T>& ray, T min_dist, T max_dist, ST st, float shift, PathRef path,
                                 uint32_t depth, bool* found_out, T* t_out, Winner* w_out) {
    const int G = st.group;
    const int lane = threadIdx.x & 63, gl = lane & (G - 1);
    const uint64_t gmask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane - gl);
    const uint64_t below = gmask & ((1ull << lane) - 1ull);
    const int fcap = st.cap * G;
    const int reserve = S.coop_reserve;
    if (fcap < 2 * reserve + 4 * G) return false;
    auto slot = [&](int e) -> int32_t* { return st.base + (e / G) * st.stride + ((e & (G - 1)) - gl); };
    const RayF f = ray_f32(ray, shift);
    const float tmin32 = round_down(min_dist - (T)shift);
    float bound32 = round_up(max_dist - (T)shift);
    const float MISS = __int_as_float(0x7f800000);
    int count = 1;
    if (gl == 0) *slot(0) = S.tlas_root4;
    bool found = false;
    int32_t best_rank = -1;
    Winner best{-1, -1};
    while (count > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int n = count < G ? count : G;
        if (count + 3 * n > fcap - reserve) n = 1;
        const bool active = gl < n;
        const int32_t ref = active ? *slot(count - n + gl) : RT_EMPTY;
        count -= n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int32_t c[4] = {RT_EMPTY, RT_EMPTY, RT_EMPTY, RT_EMPTY};
        float key[4] = {MISS, MISS, MISS, MISS};
        T t = max_dist;
        int32_t rank = -1;
        Winner w{-1, -1};
        bool hit = false;
        if (ref >= 0) {
            const Node4Regs nd = load_node4(S.nodes4, ref);
            float near[4], far[4];
            node4_boxes(nd, f, tmin32, bound32, near, far);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[k] = nd.child(k);
                key[k] = ((c[k] != RT_EMPTY) & (far[k] >= near[k])) ? near[k] : MISS;
            }
        } else if (ref != RT_EMPTY) {
            const int oi = -1 - ref;
            T th;
            int prim;
            if (object_hit<T, false, FEAT>(S, oi, ray, min_dist, max_dist, st, st.cap, shift, path, depth, &th, &prim)) {
                hit = true;
                t = th;
                rank = S.objects[oi].dfs_rank;
                w = Winner{oi, prim};
            }
        }
        for (uint64_t hits = __ballot(hit) & gmask; hits; hits &= hits - 1) {
            const int from = __ffsll((long long)hits) - 1;
            const T o_t = __shfl(t, from);
            const int32_t o_rank = __shfl(rank, from);
            const int o_obj = __shfl(w.obj, from), o_prim = __shfl(w.prim, from);
            if (o_t <= max_dist && (!found || o_t < max_dist || o_rank > best_rank)) {
                max_dist = o_t;
                bound32 = round_up(o_t - (T)shift);
                best = Winner{o_obj, o_prim};
                best_rank = o_rank;
                found = true;
            }
        }
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const bool push = key[k] < MISS && key[k] <= bound32;
            const uint64_t m = __ballot(push) & gmask;
            if (push) *slot(count + __popcll(m & below)) = c[k];
            count += __popcll(m);
        }
    }
    *found_out = found;
    *t_out = max_dist;
    *w_out = best;
    return true;
}

// world.hit for a swept world of rect lists, answered by a group of G lanes that all hold the same ray (the end of a tail
// launch, where a wave is down to a few paths and a lone lane would test the rects one after the other): lane gl of the
// group takes the rects gl, gl + G, .. of S.sweep_items, each in its object's space, each exactly as rects_hit_uniform
// tests it; the group's answer is the sweep's: the smallest t, among equal t the object that comes LATER in the sweep
// (an object is tested against the bound the earlier ones left, and t == bound passes), within one object the EARLIER rect
// (strict < keeps the first, src/hitting.rs:88-104).  Every lane of the group returns the same answer.
template <typename T>
__device__ __forceinline__ bool sweep_hit_group(const DScene<T>& S, const Ray<T>& ray, T min_dist, int G, T* t_out, Winner* w_out) {
    const int gl = (int)(threadIdx.x & 63u) & (G - 1);
    bool have = false;
    T bt = 0;
    int b_k = -1, b_rect = -1, b_obj = -1;
    for (int i = gl; i < S.n_sweep_items; i += G) {
        const int4 it = S.sweep_items[i];  // (object, rect, transform, place in the sweep)
        Ray<T> r = ray;
        if (it.z >= 0) r = to_object_space(S.xforms[it.z], ray);
        T th;
        if (rect_hit(S.rects[it.y], r, min_dist, Lim<T>::inf(), &th)) {
            // (a lane's rects come in the sweep's order: a later object takes an equal t, a later rect of the same object does not)
            if (!have || th < bt || (th == bt && it.w != b_k)) {
                bt = th; b_k = it.w; b_rect = it.y; b_obj = it.x;
            }
            have = true;
        }
    }
    for (int off = G >> 1; off > 0; off >>= 1) {
        const bool o_have = __shfl_xor((int)have, off) != 0;
        const T o_t = __shfl_xor(bt, off);
        const int o_k = __shfl_xor(b_k, off), o_rect = __shfl_xor(b_rect, off), o_obj = __shfl_xor(b_obj, off);
        const bool take = o_have && (!have || o_t < bt || (o_t == bt && (o_k > b_k || (o_k == b_k && o_rect < b_rect))));
        if (take) {
            bt = o_t; b_k = o_k; b_rect = o_rect; b_obj = o_obj;
        }
        have = have || o_have;
    }
    *t_out = bt;
    *w_out = Winner{b_obj, b_rect};
    return have;
}

template <typename T, int MODE, typename ST>
__device__ bool closest_hit(const DScene<T>& S, const Ray<T>& ray, T min_dist, T max_dist, ST st, PathRef path,
                            uint32_t depth, T* t_out, Winner* w_out) {
    constexpr bool SWEEP = MODE == WALK_SWEEP || MODE == WALK_SWEEP_LEAN || MODE == WALK_SWEEP_FLAT;
    constexpr bool UNI = SWEEP;  // the top-level list is walked in step by all lanes
    constexpr int FEAT = WalkFeat<MODE>::value;
    bool have = false;
    T best_t = 0;
    Winner best{-1, -1};
    const float shift = (SWEEP || MODE == WALK_OWN_BOX_LDS) && (FEAT & FEAT_BLAS) == 0 ? 0.0f : far_shift(S, ray);  // (box tests in f32 only)
    for (int e = 0; e < S.n_top; e++) {
        int entry = UNI ? rec<true>(&S.top_list[e]) : S.top_list[e];
        bool found = false;
        T t = 0;
        Winner w{-1, -1};
        if (entry != RT_TOP_BVH) {
            found = object_hit<T, UNI, FEAT>(S, entry, ray, min_dist, max_dist, st, 0, shift, path, depth, &t, &w.prim);
            w.obj = entry;
        } else if (SWEEP) {
            // Worlds of a few objects: no tree walk at all.  The leaves of the reference BVH are visited
            // in its depth-first order (left child first) with the same shrinking bound and the same
            // "t <= bound: the later leaf wins" rule, so the winner is the one BVHNode::hit returns
            // (src/hitting.rs:211-229); the boxes the reference tests on the way only ever skip leaves
            // that cannot be hit inside the bound.  Every lane of the wave tests the same object in the
            // same iteration: no divergence, no stack, and the object / transform / primitive records
            // come in through scalar loads.
            T bound = max_dist;
            for (int k = 0; k < S.n_sweep; k++) {
                const int oi = rec<true>(&S.sweep[k]);
                const DObject ob = rec<true>(&S.objects[oi]);
                T th = 0;
                int prim = -1;
                bool hit;
                if (ob.kind == RT_OBJ_RECTS) {  // walls, Blocks, Spotlights: the lean path (scalar branches only)
                    if (ob.xform >= 0) {
                        const Ray<T> r = to_object_space(rec<true>(&S.xforms[ob.xform]), ray);
                        hit = rects_hit_uniform(S, r, ob.first, ob.count, min_dist, bound, &th, &prim);
                    } else {
                        hit = rects_hit_uniform(S, ray, ob.first, ob.count, min_dist, bound, &th, &prim);
                    }
                } else {
                    hit = object_hit<T, true, FEAT>(S, oi, ray, min_dist, bound, st, 0, shift, path, depth, &th, &prim);
                }
                bound = hit ? th : bound;
                w.obj = hit ? oi : w.obj;
                w.prim = hit ? prim : w.prim;
                found = found | hit;
            }
            t = bound;
        } else if (MODE == WALK_OWN_BOX_LDS) {
            // Small worlds (staged in LDS): reference-shaped nodes, each with its own f64 box.  Nearly
            // every ray of a closed box visits nearly every node, so the lanes of a wave stay in step.
            V3<T> inv_d = mk((T)1.0 / ray.d.x, (T)1.0 / ray.d.y, (T)1.0 / ray.d.z);
            T bound = max_dist;
            int sp = 0;
            int cur = S.tlas_root_own;
            for (;;) {
                const DNode<T>& n = S.nodes[cur];
                int left = n.left;
                if (left < 0) {
                    int oi = -1 - left;
                    T th;
                    int prim;
                    if (object_hit<T, false, FEAT>(S, oi, ray, min_dist, bound, st, sp, shift, path, depth, &th, &prim)) {
                        bound = th;
                        w.obj = oi;
                        w.prim = prim;
                        found = true;
                    }
                } else if (box_hit(n, ray.o, inv_d, min_dist, bound)) {
                    st.put(sp++, n.right);
                    cur = left;
                    continue;
                }
                if (sp == 0) break;
                cur = st.get(--sp);
            }
            t = bound;
        } else {
            bool grouped = false;
            if constexpr (ST::COOP && walks_node4(MODE) && (FEAT & FEAT_BLAS) == 0)
                grouped = world_hit4_group<T, FEAT>(S, ray, min_dist, max_dist, st, shift, path, depth, &found, &t, &w);
            if (!grouped) {
            const RayF f = ray_f32(ray, shift);
            const float tmin32 = round_down(min_dist - (T)shift);
            T bound = max_dist;
            float bound32 = round_up(bound - (T)shift);
            int sp = 0;
            int32_t next = walks_node4(MODE) ? S.tlas_root4 : S.tlas_root;
            int32_t best_rank = -1;
            RT_PROF_DECL;
            // A node step for the lanes that hold a node, then the exact object tests for those that hold a leaf - at once
            // (S.obj_frac = 0), or when S.obj_frac sixteenths of the lanes still walking hold one or none has a node left
            // (a mesh leaf is a whole mesh walk: the lanes that meet the mesh had better enter it together).
            for (;;) {
                if (next >= 0) {
                    if constexpr (walks_node4(MODE)) {  // four-wide step, as in blas_hit4
                        RT_PROF(6);
                        RT_STEP_COUNT;
                        const Node4Regs n = load_node4(S.nodes4, next);
                        next = node4_step<ST, (FEAT & FEAT_BLAS) != 0>(n, f, tmin32, bound32, S.ordered != 0, st, sp);
                    } else {
                        const Node2Regs n = load_node2(S.nodes2, next);
                        const int32_t c0 = n.child(0), c1 = n.child(1);
                        float e0, e1;
                        const bool h0 = c0 != RT_EMPTY && box2_hit(n.lo(0), n.hi(0), f, tmin32, bound32, &e0);
                        const bool h1 = c1 != RT_EMPTY && box2_hit(n.lo(1), n.hi(1), f, tmin32, bound32, &e1);
                        next = RT_EMPTY;
                        if (h0 && h1) {
                            const bool far_first = S.ordered && e1 < e0;
                            next = far_first ? c1 : c0;
                            st.put(sp++, far_first ? c0 : c1);
                        } else if (h0) {
                            next = c0;
                        } else if (h1) {
                            next = c1;
                        }
                    }
                }
                if (next == RT_EMPTY) {
                    if (sp == 0) break;
                    next = st.get(--sp);
                }
                if constexpr (!ST::COOP) {
                    if (S.obj_frac > 0) {
                        const int n_leaf = __popcll(__ballot(next < 0)), n_node = __popcll(__ballot(next >= 0));
                        if (n_node != 0 && n_leaf < (((n_leaf + n_node) * S.obj_frac + 15) >> 4)) continue;
                    }
                }
                if (next < 0) {  // (one test per round and lane, as in blas_hit4_lane)
                    RT_PROF(4);
                    RT_STEP_COUNT;
                    const int oi = -1 - next;
                    T th;
                    int prim;
                    if (object_hit<T, false, FEAT>(S, oi, ray, min_dist, bound, st, sp, shift, path, depth, &th, &prim)) {
                        // nearer-child-first walk (see blas_hit): among exactly equal distances the leaf
                        // that is last in the reference's depth-first order wins
                        const int32_t rank = S.objects[oi].dfs_rank;
                        if (!found || th < bound || rank > best_rank) {
                            bound = th;
                            bound32 = round_up(th - (T)shift);
                            w.obj = oi;
                            w.prim = prim;
                            best_rank = rank;
                            found = true;
                        }
                    }
                    if (sp == 0) break;
                    next = st.get(--sp);
                }
            }
            RT_PROF_FLUSH;
            t = bound;
            }
        }
        if (found && (!have || t < best_t)) {
            best_t = t;
            best = w;
            have = true;
        }
    }
    *t_out = best_t;
    *w_out = best;
    return have;
}

// Returned by value (in registers): with out-parameters the caller's whole Surface had to live in scratch
// memory, and its stores and reloads were HBM traffic in every kernel that can shade a sphere.
template <typename T>
struct UV {
    T u, v;
};
template <typename T>
__device__ __noinline__ UV<T> get_sphere_uv(V3<T> p) {  // src/math.rs:288-294
    const T PI = (T)3.14159265358979323846264338327950288;
    T theta = acos(-p.y);
    T phi = atan2(-p.z, p.x) + PI;
    return UV<T>{phi / ((T)2.0 * PI), theta / PI};
}

template <typename T>
__device__ __noinline__ T plane_uv(T x, T repeat) {  // src/objects.rs:609-616 (f64 fmod is a long loop: out of line)
    T u = fmod(x, repeat) / repeat;
    return u < (T)0.0 ? (T)1.0 + u : u;
}

template <typename T>
struct Surface {  // HitRecord, src/hitting.rs:33-41
    V3<T> normal;
    T u, v;
    bool front_face;
    int material, owner, sub;
};

// Rebuilds the HitRecord of the winning primitive: outward normal and (u,v) of the
// primitive, HitRecord::new's face logic (src/hitting.rs:57-79), then the normal rotated
// back to world space as the Rotate*::hit wrappers do (src/transforms.rs:119-121 etc.).
template <typename T, int FEAT>
__device__ Surface<T> make_surface(const DScene<T>& S, const Ray<T>& world_ray, T t, Winner w, bool want_uv) {
    // want_uv == false (no material of the scene reads (u, v), and nobody asked for the hit record):
    // (u, v) stay 0; a rect then needs no object-space ray at all, a sphere no acos / atan2.
    const DObject ob = S.objects[w.obj];
    const int kind = ob.kind;
    const bool rect_short = kind == RT_OBJ_RECTS && !want_uv;
    Ray<T> r = world_ray;
    if (ob.xform >= 0 && !rect_short) r = to_object_space(S.xforms[ob.xform], world_ray);
    bool front_known = false;
    Surface<T> sf;
    V3<T> outward = mk((T)1.0, (T)0.0, (T)0.0);
    sf.u = 0;
    sf.v = 0;
    sf.sub = 0;
    bool medium = false, unit_already = false;
    sf.material = 0;
    sf.owner = -1;
    sf.front_face = false;
    if (kind == RT_OBJ_RECTS) {
        const DRect<T>& q = S.rects[w.prim];
        int axis = q.axis;
        T s = q.facing_positive ? (T)1.0 : (T)-1.0;
        if (want_uv) {
            int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
            T a = comp(r.o, ia) + t * comp(r.d, ia);
            T b = comp(r.o, ib) + t * comp(r.d, ib);
            sf.u = (a - q.a0) / (q.a1 - q.a0);
            sf.v = (b - q.b0) / (q.b1 - q.b0);
        } else {
            // front_face = dot(d_object, outward) < 0 = (d_object[axis] * s) < 0 (the other two products
            // are +-0).  d_object = unit_vector(M d) has the sign of (M d)[axis]: the division by the
            // length cannot change it, so one matrix row replaces the whole transform.
            T dn = comp(world_ray.d, axis);
            if (ob.xform >= 0) {
                const T* m = S.xforms[ob.xform].m + 3 * axis;
                dn = m[0] * world_ray.d.x + m[1] * world_ray.d.y + m[2] * world_ray.d.z;
            }
            sf.front_face = dn * s < (T)0.0;
            front_known = true;
        }
        outward = mk(axis == 0 ? s : (T)0.0, axis == 1 ? s : (T)0.0, axis == 2 ? s : (T)0.0);
        unit_already = true;  // (+-1, 0, 0) / sqrt(1) is itself, bit for bit: skip HitRecord::new's unit_vector
        sf.material = q.material;
        sf.owner = q.owner;
        sf.sub = q.sub;
    } else if ((FEAT & FEAT_SPHERES) != 0 && kind == RT_OBJ_SPHERE) {
        const DSphere<T>& s = S.spheres[w.prim];
        outward = (r.at(t) - ld3(s.centre)) / s.radius;
        if (want_uv) {
            const UV<T> uv = get_sphere_uv(outward);
            sf.u = uv.u;
            sf.v = uv.v;
        }
        sf.material = s.material;
        sf.owner = s.owner;
    } else if ((FEAT & FEAT_SPHERES) != 0 && kind == RT_OBJ_MOVING_SPHERE) {
        const DMovingSphere<T>& s = S.moving_spheres[w.prim];
        outward = (r.at(t) - moving_centre(s, r.time)) / s.radius;
        if (want_uv) {
            const UV<T> uv = get_sphere_uv(outward);
            sf.u = uv.u;
            sf.v = uv.v;
        }
        sf.material = s.material;
        sf.owner = s.owner;
    } else if ((FEAT & (FEAT_FLAT | FEAT_BLAS)) != 0 && (kind == RT_OBJ_TRIANGLE || kind == RT_OBJ_BLAS)) {
        const DTriangle<T>& tr = S.triangles[w.prim];
        V3<T> s;
        line_plane_collision(r.o, r.d, ld3(tr.point), ld3(tr.vec1), ld3(tr.vec2), &s);
        sf.u = s.z;  // (solution[2], solution[1]), src/objects.rs:552
        sf.v = s.y;
        outward = ld3(tr.normal);
        sf.material = tr.material;
        sf.owner = tr.owner;
        sf.sub = tr.sub;
    } else if ((FEAT & FEAT_FLAT) != 0 && kind == RT_OBJ_PLANE) {  // src/objects.rs:607-617
        const DPlane<T>& p = S.planes[w.prim];
        V3<T> s;
        line_plane_collision(r.o, r.d, ld3(p.point), ld3(p.vec1), ld3(p.vec2), &s);
        sf.u = plane_uv(s.y, p.uv_repeat);
        sf.v = plane_uv(-s.z, p.uv_repeat);
        outward = ld3(p.normal);
        sf.material = p.material;
        sf.owner = p.owner;
    } else if ((FEAT & FEAT_MEDIA) != 0 && kind == RT_OBJ_MEDIUM) {  // normal (1,0,0), front_face true, uv 0 (src/objects.rs:431-439)
        const DMedium<T>& md = S.media[w.prim];
        sf.material = md.material;
        sf.owner = md.owner;
        medium = true;
    }
    if (medium) {
        sf.front_face = true;
        sf.normal = outward;
    } else {
        if (!front_known) sf.front_face = dot(r.d, outward) < (T)0.0;
        V3<T> n = unit_already ? outward : unit_vector(outward);
        sf.normal = sf.front_face ? n : -n;
    }
    if (ob.xform >= 0) sf.normal = rotate_to_world(S.xforms[ob.xform], sf.normal);
    return sf;
}

// ------------------------------------------------------------------ kernels
constexpr int BLOCK = 256;
#ifndef RT_XCD_RUN
#define RT_XCD_RUN 16  // chunks of 256 rays per run of k_extend's XCD-aware order (16 = one 4096-slot tile); 0: every eighth chunk
#endif
#define RT_XCD_RUN_ (RT_XCD_RUN > 0 ? (uint32_t)RT_XCD_RUN : 1u)
// Waves per SIMD the tail launch (k_bounce<.., CAN_TAIL = true>) is compiled for, measured per kind of walk (2 / 3 / 4):
// lean sweep 3 (cornell, 13 spp: 3.65 / 3.55 / 3.65 ms), mesh walks 4 (knight 44.96 / 44.16 / 43.78, 13 spp 9.78 /
// 9.73 / 9.31), world trees without meshes 2 (blocks_ball 19.38 / 19.62 / 20.16).  capi.cpp sizes the grid to match
// (Engine::tail_blocks_per_cu).  -DRT_TAIL_WAVES=n forces one value (A/B builds).
constexpr int tail_waves(int mode) {
#ifdef RT_TAIL_WAVES
    return RT_TAIL_WAVES;
#else
    return mode == 3 /* WALK_SWEEP_LEAN */ || mode == 8 /* WALK_SWEEP_FLAT */ ? 3 : (mode == 4 || mode == 5) /* WALK_NODE2_MESH4, WALK_NODE4_MESH */ ? 4 : 2;
#endif
}
// Waves per SIMD the per-bounce k_bounce is compiled for: the fused bounce of a world tree with media (walk, medium test
// and shading in one kernel) needs 168 registers - at 4 waves it spills 48 of them, at 3 five: blocks_ball 16.0 -> 15.5 ms.
// (f64 only: in f32 the kernel fits and the fourth wave is worth 10 %)
constexpr int bounce_waves(int mode, size_t elem) { return mode == 6 /* WALK_NODE4_MEDIA */ && elem == 8 ? 3 : RT_EXTEND_WAVES; }
template <int MODE, bool DEEP = walks_node4(MODE)> struct StackFor {
    typedef Stack type;
    static constexpr int SPILL = 1;
    static __device__ __forceinline__ Stack make(int32_t* lds, int, int32_t*) { return Stack{lds, BLOCK}; }
};
template <int MODE> struct StackFor<MODE, true> {
    typedef DeepStack type;
    static constexpr int SPILL = RT_STACK_OVERFLOW;
    static __device__ __forceinline__ DeepStack make(int32_t* lds, int cap, int32_t* spill) { return DeepStack{lds, BLOCK, cap, spill}; }
};
// Paths live in fixed tiles of TILE queue slots; tile t of the output queue receives the
// survivors of tile t of the input queue, packed at its front, and a per-tile count.
constexpr uint32_t TILE = RT_TILE;

template <typename T>
__device__ __forceinline__ Ray<T> load_ray(const DQueue<T>& q, uint32_t i) {
    Ray<T> r;
    r.o = mk(q.ox()[i], q.oy()[i], q.oz()[i]);
    r.d = mk(q.dx()[i], q.dy()[i], q.dz()[i]);
    r.time = q.time()[i];
    return r;
}
template <typename T>
__device__ __forceinline__ void store_ray(const DQueue<T>& q, uint32_t i, const Ray<T>& r) {
    q.ox()[i] = r.o.x; q.oy()[i] = r.o.y; q.oz()[i] = r.o.z;
    q.dx()[i] = r.d.x; q.dy()[i] = r.d.y; q.dz()[i] = r.d.z;
    q.time()[i] = r.time;
}


// Dense work index -> queue slot.  Live entries are packed at the front of each tile;
// tile_offsets is the exclusive prefix sum of the per-tile counts (tile_offsets[n_tiles] =
// total), so work item d lives in the tile t with offsets[t] <= d < offsets[t+1].  The tile of
// a wave's first item is found once per wave by a (scalar) binary search; the other lanes
// step forward from it.
__device__ __forceinline__ uint32_t slot_of(const uint32_t* __restrict__ offsets, uint32_t n_tiles, uint32_t d,
                                            uint32_t d_first, uint32_t* tile_out) {
    uint32_t d0 = __builtin_amdgcn_readfirstlane(d_first);
    uint32_t lo = 0, hi = n_tiles;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= d0) lo = mid; else hi = mid;
    }
    uint32_t tile = lo;
    while (d >= offsets[tile + 1]) tile++;
    *tile_out = tile;
    return tile * TILE + (d - offsets[tile]);
}

// One block: exclusive scan of the per-tile counts of the queue that is read next; also
// clears the counts of the queue that is written next and keeps the global totals.
#if RT_PART_HAS(2)
constexpr int SCAN_THREADS = 1024;
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_tiles(const uint32_t* __restrict__ tile_counts, uint32_t n_tiles,
                                                              uint32_t* __restrict__ tile_offsets,
                                                              uint32_t* __restrict__ counts_to_zero,
                                                              uint32_t* __restrict__ total_out,
                                                              unsigned long long* __restrict__ ray_total,
                                                              uint32_t* __restrict__ host_total, uint32_t dense_total) {
    __shared__ uint32_t partial[SCAN_THREADS];
    const uint32_t per = (n_tiles + SCAN_THREADS - 1) / SCAN_THREADS;
    const uint32_t begin = min(n_tiles, threadIdx.x * per), end = min(n_tiles, begin + per);
    if (tile_counts == nullptr) {
        // a dense queue of dense_total entries that nobody has written yet (the first bounce makes its own camera rays):
        // its totals and the zeroing of the other queue's counts, no offsets
        for (uint32_t t = begin; t < end; t++)
            if (counts_to_zero) counts_to_zero[t] = 0;
        if (threadIdx.x == SCAN_THREADS - 1) {
            total_out[0] = dense_total;
            total_out[1] = 0;
            total_out[3] = 0;
            if (ray_total) *ray_total += dense_total;
            if (host_total) __hip_atomic_store(host_total, dense_total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    uint32_t sum = 0;
    for (uint32_t t = begin; t < end; t++) sum += tile_counts[t];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t v = threadIdx.x >= (uint32_t)off ? partial[threadIdx.x - off] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;  // exclusive prefix of this thread's segment
    for (uint32_t t = begin; t < end; t++) {
        tile_offsets[t] = run;
        run += tile_counts[t];
        if (counts_to_zero) counts_to_zero[t] = 0;
    }
    if (threadIdx.x == SCAN_THREADS - 1) {
        uint32_t total = partial[SCAN_THREADS - 1];
        tile_offsets[n_tiles] = total;
        total_out[0] = total;
        total_out[1] = 0;  // k_extend_refill's cursor
        total_out[3] = 0;  // the tail launch's cursor
        if (ray_total) *ray_total += total;  // every live entry becomes one closest-hit query
        // the host follows the live count through pinned memory, without waiting for the stream
        if (host_total) __hip_atomic_store(host_total, total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

// The camera ray of the path that sits in queue slot `slot` of a batch of n_paths (src/main.rs:132-135,
// Camera::find_ray src/camera.rs:67-75), and the path's id.  The slot -> path mapping is the order of the queue
// (DBatch::blocked); the path's id, which its random streams are keyed by, does not depend on it.
template <typename T>
__device__ __forceinline__ Ray<T> primary_ray(const DScene<T>& S, const DBatch& B, uint32_t slot, uint32_t n_paths, uint32_t* pid_out) {
    uint32_t pid = slot;
    if (B.blocked == 2u) {
        // A wave = 64 / n neighbouring pixels x n samples of each, n the largest power of two (<= 64) that the batch's
        // samples still hold: its samples are taken in runs of 64, then 32, 16, ... as the binary digits of their count
        // say.  (The width is a multiple of 8; every quotient but one is a shift.)
        const uint32_t n_samples = n_paths / B.npix, s_slot = slot / B.npix;
        uint32_t run_first = s_slot & ~63u, run = 64u;  // the run of samples this slot is in
        if (run_first + 64u > n_samples) {
            run_first = n_samples & ~63u;
            for (run = 32u; run > 1u; run >>= 1) {
                if ((n_samples - run_first) & run) {
                    if (s_slot < run_first + run) break;
                    run_first += run;
                }
            }
        }
        const uint32_t area_log2 = 6u - (31u - (uint32_t)__clz((int)run));           // pixels of a wave = 64 / run
        const uint32_t bw_log2 = area_log2 >> 1, bh_log2 = area_log2 - bw_log2;        // 1x1, 1x2, 2x2, 2x4, 4x4, 4x8, 8x8 (wide x high)
        const uint32_t rem = slot - run_first * B.npix;
        const uint32_t npix8 = B.width * (B.rows & ~7u);  // the rows that make whole strips of 8; the others follow, along their rows
        if (rem < run * npix8) {
            const uint32_t blk = rem >> 6, inner = rem & 63u;
            const uint32_t s_in = inner >> area_log2, in_px = inner & ((1u << area_log2) - 1u);
            const uint32_t py = in_px >> bw_log2, px = in_px & ((1u << bw_log2) - 1u);
            const uint32_t across = B.width >> bw_log2, by = blk / across, bx = blk - by * across;
            pid = (run_first + s_in) * B.npix + ((by << bh_log2) + py) * B.width + (bx << bw_log2) + px;
        } else {
            const uint32_t rest = rem - run * npix8, tail_px = B.npix - npix8, s_in = rest / tail_px;
            pid = (run_first + s_in) * B.npix + npix8 + (rest - s_in * tail_px);
        }
    } else if (B.blocked) {  // the slot counts through 8 x 8 blocks of pixels (DBatch::blocked)
        const uint32_t sl = slot / B.npix, pl = slot - sl * B.npix;
        const uint32_t W8 = B.width * 8u, strip = pl / W8;
        uint32_t p = pl;  // (a last strip of fewer than 8 rows: along its rows, as pl counts them)
        if (8u * strip + 8u <= B.rows) {
            const uint32_t rem = pl - strip * W8, chunk = rem >> 6, inner = rem & 63u;
            p = (8u * strip + (inner >> 3)) * B.width + 8u * chunk + (inner & 7u);
        }
        pid = sl * B.npix + p;
    }
    uint32_t pixel, sample;
    path_ids(B, pid, &pixel, &sample);
    uint32_t i = pixel % B.width, j = pixel / B.width;
    Rng<T> rng(B.seed, DOMAIN_CAMERA, pixel, sample, 0);
    // src/main.rs:132-134
    T s = ((T)i + rng.range((T)0.0, (T)1.0)) / (T)(B.width - 1);
    T t = ((T)j + rng.range((T)0.0, (T)1.0)) / (T)(B.height - 1);
    // Camera::find_ray, src/camera.rs:67-75
    const DCamera<T>& c = S.camera;
    V3<T> rd = c.lens_radius * random_in_unit_disc(rng);
    V3<T> offset = ld3(c.u) * rd.x + ld3(c.v) * rd.y;
    Ray<T> r;
    r.o = ld3(c.origin) + offset;
    r.d = unit_vector(ld3(c.lower_left_corner) + s * ld3(c.horizontal) + t * ld3(c.vertical) - ld3(c.origin) - offset);
    r.time = rng.range(c.start_time, c.end_time);
    *pid_out = pid;
    return r;
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_raygen(DScene<T> S, DBatch B, DQueue<T> q, uint32_t n_paths,
                                                   uint32_t* tile_counts) {
    const uint32_t slot = blockIdx.x * BLOCK + threadIdx.x;  // place in the queue
    if (slot >= n_paths) return;
    if (slot % TILE == 0) tile_counts[slot / TILE] = min(TILE, n_paths - slot);
    uint32_t pid;
    const Ray<T> r = primary_ray(S, B, slot, n_paths, &pid);
    store_ray(q, slot, r);
    q.tr()[slot] = (T)1.0; q.tg()[slot] = (T)1.0; q.tb()[slot] = (T)1.0;
    q.pid[slot] = pid;
}

// Copies a small scene's traversal records into LDS (all threads of the block) and points S at them.
template <typename T>
__device__ __forceinline__ void stage_small_scene(DScene<T>& S, const DSmallScene& small, uint4* lds_scene) {
    const uint4* src = (const uint4*)small.blob;
    for (uint32_t k = threadIdx.x; k < small.bytes / 16; k += BLOCK) lds_scene[k] = src[k];
    __syncthreads();
    const unsigned char* base = (const unsigned char*)lds_scene;
    S.nodes = (const DNode<T>*)(base + small.off[0]);
    S.objects = (const DObject*)(base + small.off[1]);
    S.xforms = (const DXform<T>*)(base + small.off[2]);
    S.spheres = (const DSphere<T>*)(base + small.off[3]);
    S.moving_spheres = (const DMovingSphere<T>*)(base + small.off[4]);
    S.rects = (const DRect<T>*)(base + small.off[5]);
    S.planes = (const DPlane<T>*)(base + small.off[6]);
    S.media = (const DMedium<T>*)(base + small.off[7]);
    S.top_list = (const int32_t*)(base + small.off[8]);
}

// LDS_SCENE: scenes whose acceleration structure and primitives (everything but triangles,
// materials and texels) fit in RT_LDS_SCENE_BYTES are copied into LDS by every block once and
// traversed from there: the traversal is a chain of dependent record fetches (node -> object ->
// transform -> primitive), and an LDS read returns in ~1/5 of the time of an L2 hit.
template <typename T, int STACK, int MODE>
__global__ void __launch_bounds__(BLOCK, RT_EXTEND_WAVES) k_extend(DScene<T> S, DSmallScene small, DBatch B, DQueue<T> q, DHits<T> h, const T* tmin_arr,
                                                   const T* tmax_arr, uint32_t depth,
                                                   const uint32_t* __restrict__ tile_offsets, uint32_t n_tiles) {
    __shared__ int32_t lds_stack[STACK * BLOCK];
    constexpr bool LDS_SCENE = MODE == WALK_OWN_BOX_LDS;
    __shared__ uint4 lds_scene[LDS_SCENE ? RT_LDS_SCENE_BYTES / 16 : 1];
    int32_t stack_spill[StackFor<MODE>::SPILL];
    typename StackFor<MODE>::type st = StackFor<MODE>::make(lds_stack + threadIdx.x, S.stack_cap > 0 && S.stack_cap < STACK ? S.stack_cap : STACK, stack_spill);
    if (LDS_SCENE) stage_small_scene(S, small, lds_scene);
    const uint32_t total = tile_offsets[n_tiles];
    const uint32_t lane = threadIdx.x & 63;
    // XCD-aware order of the 256-ray chunks (RT_XCD_RUN > 0): the blocks that share an XCD (blockIdx % 8) take
    // runs of RT_XCD_RUN consecutive chunks, dealt round-robin to the eight groups, instead of every eighth
    // chunk.  Consecutive chunks hold neighbouring pixels' paths, so the rays an XCD has in flight at a time
    // come from an eighth as many image rows, and its 4 MB L2 has an eighth of the nodes and triangles to hold.
    const bool xcd_order = RT_XCD_RUN > 0 && (gridDim.x & 7u) == 0;
    const uint32_t chunk_step = xcd_order ? gridDim.x >> 3 : gridDim.x;
    for (uint32_t m = xcd_order ? blockIdx.x >> 3 : blockIdx.x;; m += chunk_step) {
        const uint32_t chunk = xcd_order ? ((m / RT_XCD_RUN_) * 8u + (blockIdx.x & 7u)) * RT_XCD_RUN_ + m % RT_XCD_RUN_ : m;
        if (chunk * BLOCK >= total) break;  // chunk grows with m in either order
        const uint32_t base = chunk * BLOCK + (threadIdx.x & ~63u);
        const uint32_t d = base + lane;
        if (d >= total) continue;
        uint32_t tile;
        const uint32_t i = slot_of(tile_offsets, n_tiles, d, base, &tile);
        Ray<T> r = load_ray(q, i);
        const PathRef path = path_of(B, q.pid[i]);
        T min_dist = tmin_arr ? tmin_arr[i] : (T)0.001;  // src/hitting.rs:20-21
        T max_dist = tmax_arr ? tmax_arr[i] : Lim<T>::inf();
        T t;
        Winner w;
        RT_EXTEND_STEPS_BEGIN;
        const bool got = closest_hit<T, MODE>(S, r, min_dist, max_dist, st, path, depth, &t, &w);
        RT_EXTEND_STEPS_END(depth, d, r, lane);
        if (got) {
            Surface<T> sf = make_surface<T, WalkFeat<MODE>::value>(S, r, t, w, S.needs_uv != 0 || h.owner != nullptr);
            h.t()[i] = t;
            h.nx()[i] = sf.normal.x; h.ny()[i] = sf.normal.y; h.nz()[i] = sf.normal.z;
            if (S.needs_uv != 0 || h.owner != nullptr) {  // nobody reads (u, v) otherwise: 16 B per hit not written, not read
                h.u()[i] = sf.u; h.v()[i] = sf.v;
            }
            h.mat_front[i] = (uint32_t)sf.material | (sf.front_face ? 0x80000000u : 0u);
            if (h.owner) {
                h.owner[i] = sf.owner;
                h.sub[i] = sf.sub;
                h.world_index[i] = S.objects[w.obj].world_index;
                h.inst[i] = w.obj;
            }
        } else {
            h.mat_front[i] = RT_MISS;
        }
    }
}

// ---- k_extend for the four-wide trees with lane refill.
// A wave of k_extend is as slow as its slowest ray: on the knight the lanes of a wave are 30 %
// (node steps) and 6 % (triangle tests) busy.  Here a wave keeps walking while lanes whose ray is
// done wait; when REFILL lanes are waiting (or nobody walks any more) they write their hit records
// and take the next rays of the queue (one atomic per refill on a launch-wide cursor).  For that
// the walk is one flat loop - a step is a node, a world leaf or a triangle, at whichever level the
// lane is - instead of closest_hit's nested loops; the arithmetic and the tie rules are the same.
template <typename T, int MODE, typename ST>
struct Walk4 {
    static constexpr int FEAT = WalkFeat<MODE>::value;
    // meshes met in the world tree are walked by step() itself; the object test only needs them for media boundaries
    static constexpr int LEAF_FEAT = (FEAT & FEAT_MEDIA) ? FEAT : (FEAT & ~(FEAT_BLAS | FEAT_BLAS4));
    Ray<T> ray;   // the query, world space
    T min_dist;
    T bound;      // world level: nearest hit so far (or the query's upper bound)
    Winner w;
    int32_t best_rank;
    bool found;
    RayF f;       // current level
    float shift;  // far_shift of the query
    float tmin32, bound32;
    int sp;
    int32_t next;
    bool in_mesh;  // mesh level (between a BLAS leaf of the world tree and the emptying of its part of the stack)
    int m_sp0, m_oi, m_prim;
    int32_t m_rank;
    bool m_found;
    Ray<T> mr;
    T m_bound;

    __device__ __forceinline__ void begin(const DScene<T>& S, const Ray<T>& r, T lo, T hi) {
        ray = r;
        min_dist = lo;
        bound = hi;
        w = Winner{-1, -1};
        best_rank = -1;
        found = false;
        shift = far_shift(S, r);
        f = ray_f32(r, shift);
        tmin32 = round_down(lo - (T)shift);
        bound32 = round_up(hi - (T)shift);
        sp = 0;
        next = S.tlas_root4;
        in_mesh = false;
        m_sp0 = 0; m_oi = -1; m_prim = -1; m_rank = -1; m_found = false;
        mr = r;
        m_bound = hi;
    }
    // next is empty: takes the next entry off the stack (leaving a finished mesh on the way); false when
    // the walk is over (found / bound / w hold the answer of the world BVH).
    __device__ __forceinline__ bool pop(const DScene<T>& S, ST& st) {
        if (in_mesh && sp == m_sp0) {  // the mesh is done: its nearest triangle is this leaf's answer
            in_mesh = false;
            if (m_found) {
                const int32_t rank = S.objects[m_oi].dfs_rank;
                if (!found || m_bound < bound || rank > best_rank) {
                    bound = m_bound;
                    w = Winner{m_oi, m_prim};
                    best_rank = rank;
                    found = true;
                }
            }
            f = ray_f32(ray, shift);
            bound32 = round_up(bound - (T)shift);
        }
        if (sp == 0) return false;
        next = st.get(--sp);
        return true;
    }
    __device__ __forceinline__ void node(const DScene<T>& S, ST& st) {  // next >= 0
        const Node4Regs n = load_node4(S.nodes4, next);
        next = node4_step<ST, (FEAT & FEAT_BLAS) != 0>(n, f, tmin32, bound32, S.ordered != 0, st, sp);
    }
    __device__ __forceinline__ void leaf(const DScene<T>& S, ST& st, PathRef path, uint32_t depth) {  // next < 0, not empty
        if (in_mesh) {
            const int tri = -1 - next;  // the exact test, against the mesh's current bound (t == bound passes)
            T t;
            const DTriangle<T> tr = S.triangles[tri];
            if (triangle_hit(tr, mr, min_dist, m_bound, &t)) {
                if (!m_found || t < m_bound || tr.dfs_rank > m_rank) {
                    m_bound = t;
                    bound32 = round_up(t - (T)shift);
                    m_prim = tri;
                    m_rank = tr.dfs_rank;
                    m_found = true;
                }
            }
            next = RT_EMPTY;
        } else {
            const int oi = -1 - next;
            const DObject ob = S.objects[oi];
            if ((FEAT & FEAT_BLAS) != 0 && ob.kind == RT_OBJ_BLAS) {  // enter the mesh (simple_object_hit's BLAS case)
                mr = ray;
                if (ob.xform >= 0) mr = to_object_space(S.xforms[ob.xform], ray);
                f = ray_f32(mr, shift);
                in_mesh = true;
                m_sp0 = sp;
                m_oi = oi;
                m_bound = bound;
                m_found = false;
                m_rank = -1;
                next = ob.first4;
            } else {
                T th;
                int prim;
                if (object_hit<T, false, LEAF_FEAT>(S, oi, ray, min_dist, bound, st, sp, shift, path, depth, &th, &prim)) {
                    if (!found || th < bound || ob.dfs_rank > best_rank) {
                        bound = th;
                        bound32 = round_up(th - (T)shift);
                        w = Winner{oi, prim};
                        best_rank = ob.dfs_rank;
                        found = true;
                    }
                }
                next = RT_EMPTY;
            }
        }
    }
};

template <typename T, int STACK, int MODE>
__global__ void __launch_bounds__(BLOCK, RT_EXTEND_WAVES) k_extend_refill(DScene<T> S, DBatch B, DQueue<T> q, DHits<T> h, const T* tmin_arr,
                                                                         const T* tmax_arr, uint32_t depth,
                                                                         const uint32_t* __restrict__ tile_offsets, uint32_t n_tiles,
                                                                         uint32_t* __restrict__ cursor, uint32_t refill) {
    static_assert(walks_node4(MODE), "four-wide trees only");
    constexpr int FEAT = WalkFeat<MODE>::value;
    __shared__ int32_t lds_stack[STACK * BLOCK];
    int32_t stack_spill[StackFor<MODE>::SPILL];
    typedef typename StackFor<MODE>::type ST;
    ST st = StackFor<MODE>::make(lds_stack + threadIdx.x, S.stack_cap > 0 && S.stack_cap < STACK ? S.stack_cap : STACK, stack_spill);
    const uint32_t total = tile_offsets[n_tiles];
    const uint32_t lane = threadIdx.x & 63;
    enum { IDLE = 0, WALK = 1, DONE = 2 };
    int state = IDLE;
    bool exhausted = false;  // (wave-uniform)
    uint32_t slot = 0, pid = 0;
    // rays a wave takes from the queue at a time: a quarter of its even share, so that the waves end together,
    // within [64, 2048] (an atomic on the launch-wide cursor per share)
    const uint32_t CHUNK = max(64u, min(2048u, (total / (gridDim.x * (BLOCK / 64) * 4u)) & ~63u));
    uint32_t chunk_lo = 0, chunk_hi = 0;                                // (wave-uniform) what is left of the wave's share
    Walk4<T, MODE, ST> walk;
    for (;;) {
        const uint64_t walking = __ballot(state == WALK);
        if (walking == 0 || (uint32_t)__popcll(__ballot(state == DONE)) >= refill) {
            if (state == DONE) {
                // the rest of world.hit: the unbounded objects of the top-level list, in its order (src/hitting.rs:88-104)
                const PathRef path = path_of(B, pid);
                const T max_dist = tmax_arr ? tmax_arr[slot] : Lim<T>::inf();
                bool have = false;
                T best_t = 0;
                Winner best{-1, -1};
                for (int e = 0; e < S.n_top; e++) {
                    const int entry = S.top_list[e];
                    bool found_here;
                    T t = 0;
                    Winner w{-1, -1};
                    if (entry != RT_TOP_BVH) {
                        found_here = object_hit<T, false, FEAT>(S, entry, walk.ray, walk.min_dist, max_dist, st, 0, walk.shift, path, depth, &t, &w.prim);
                        w.obj = entry;
                    } else {
                        found_here = walk.found;
                        t = walk.bound;
                        w = walk.w;
                    }
                    if (found_here && (!have || t < best_t)) {
                        best_t = t;
                        best = w;
                        have = true;
                    }
                }
                const uint32_t i = slot;
                if (have) {
                    Surface<T> sf = make_surface<T, FEAT>(S, walk.ray, best_t, best, S.needs_uv != 0 || h.owner != nullptr);
                    h.t()[i] = best_t;
                    h.nx()[i] = sf.normal.x; h.ny()[i] = sf.normal.y; h.nz()[i] = sf.normal.z;
                    if (S.needs_uv != 0 || h.owner != nullptr) {
                        h.u()[i] = sf.u; h.v()[i] = sf.v;
                    }
                    h.mat_front[i] = (uint32_t)sf.material | (sf.front_face ? 0x80000000u : 0u);
                    if (h.owner) {
                        h.owner[i] = sf.owner;
                        h.sub[i] = sf.sub;
                        h.world_index[i] = S.objects[best.obj].world_index;
                        h.inst[i] = best.obj;
                    }
                } else {
                    h.mat_front[i] = RT_MISS;
                }
                state = IDLE;
            }
            while (!exhausted) {
                const uint64_t idle = __ballot(state == IDLE);
                if (idle == 0) break;
                if (chunk_lo >= chunk_hi) {
                    // the wave's share of the queue has run out: a new one (one atomic per CHUNK rays - every
                    // wave of the launch shares the cursor, and atomics on one address are served one by one)
                    uint32_t fresh = 0;
                    if (lane == 0) fresh = atomicAdd(cursor, CHUNK);
                    fresh = (uint32_t)__shfl((int)fresh, 0);
                    if (fresh >= total) {
                        exhausted = true;
                        break;
                    }
                    chunk_lo = fresh;
                    chunk_hi = min(fresh + CHUNK, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t avail = chunk_hi - chunk_lo;
                if (state == IDLE && rank < avail) {
                    uint32_t tile;
                    slot = slot_of(tile_offsets, n_tiles, chunk_lo + rank, chunk_lo, &tile);
                    const Ray<T> r = load_ray(q, slot);
                    pid = q.pid[slot];
                    walk.begin(S, r, tmin_arr ? tmin_arr[slot] : (T)0.001, tmax_arr ? tmax_arr[slot] : Lim<T>::inf());
                    state = WALK;
                }
                chunk_lo += min((uint32_t)__popcll(idle), avail);
            }
            if (__ballot(state == WALK) == 0) break;
        }
        if (state == WALK) {
            // node steps until this lane holds a leaf (or is through); then the leaves of all lanes together
            // (a vote as in blas_hit4_lane - one node step per round, the leaf round when half the lanes hold a leaf -
            // loses here: blocks_ball 17.8 -> 18.1 ms at its best setting; the object tests of a world tree are many
            // times a node step, and every round pays the wave's bookkeeping above)
            for (;;) {
                if (walk.next == RT_EMPTY && !walk.pop(S, st)) {
                    state = DONE;
                    break;
                }
                if (walk.next < 0) break;
                walk.node(S, st);
            }
        }
        if (state == WALK) walk.leaf(S, st, path_of(B, pid), depth);
    }
}

// Texture::value: SolidColour (src/textures.rs:18-21) / ImageTexture (:39-52).
template <typename T>
__device__ __forceinline__ V3<T> texture_value(const DScene<T>& S, const DMaterial<T>& m, T u, T v) {
    if (m.image < 0) return ld3(m.colour);
    const DImage im = S.images[m.image];
    u = u < (T)0.0 ? (T)0.0 : u > (T)1.0 ? (T)1.0 : u;
    v = (T)1.0 - (v < (T)0.0 ? (T)0.0 : v > (T)1.0 ? (T)1.0 : v);
    T fi = u * (T)im.width, fj = v * (T)im.height;
    uint32_t i = fi != fi ? 0u : fi <= (T)0.0 ? 0u : fi >= (T)4294967295.0 ? 4294967295u : (uint32_t)fi;
    uint32_t j = fj != fj ? 0u : fj <= (T)0.0 ? 0u : fj >= (T)4294967295.0 ? 4294967295u : (uint32_t)fj;
    i = min(i, (uint32_t)im.width - 1);
    j = min(j, (uint32_t)im.height - 1);
    const uint8_t* px = S.texels + im.texel_offset + ((uint64_t)j * (uint32_t)im.width + i) * 3;
    const T colour_scale = (T)1.0 / (T)255.0;
    return colour_scale * mk((T)px[0], (T)px[1], (T)px[2]);
}

template <typename T>
__device__ __noinline__ T reflectance(T cosine, T ref_idx) {  // src/materials.rs:107-111 (1.1 as written)
    T r0 = ((T)1.1 - ref_idx) / ((T)1.0 + ref_idx);
    r0 = r0 * r0;
    return r0 + ((T)1.0 - r0) * pow((T)1.0 - cosine, (T)5.0);
}

// Checkered::{emitted, scatter} (src/materials.rs:166-183) pick a part by the sign of
// sin(k f u) sin(k f v), f = PI in scatter and 1 in emitted, possibly through nested checkers.
// Only the SIGN of each sine is needed, and sin(x) = (-1)^n sin(x - n pi) with |x - n pi| <= pi / 2 has the sign of
// (-1)^n (x - n pi): found from the place of x among the multiples of pi (two-term pi, fused steps) in a dozen
// instructions instead of two f64 sines of ~100 each.  Where that place is not safely decided - x within 1e-12 |n| of a
// multiple, huge, tiny, not finite - or where a product of two tiny sines could underflow, the sines themselves are
// taken (out of line: f64 sin carries a large argument reduction that would otherwise sit in every kernel), so the part
// chosen is always the one `sines < 0` chooses.
template <typename T>
__device__ __noinline__ bool checker_is_odd_by_sines(T a, T b) {
    return sin(a) * sin(b) < (T)0.0;
}
// +1 / -1: the sign of sin(x); 0: undecided here
__device__ __forceinline__ int sin_sign(double x) {
    const double ax = fabs(x);
    if (!(ax < 1e15) || ax < 1e-100) return 0;
    const double n = rint(x * 0.31830988618379067154);
    double r = fma(-n, 3.141592653589793116, x);     // pi = 0x1.921fb54442d18p+1 + 0x1.1a62633145c07p-53 + ...
    r = fma(-n, 1.2246467991473532072e-16, r);
    if (!(fabs(r) > 1e-12 * fmax(1.0, fabs(n)))) return 0;
    const bool odd = ((long long)n & 1LL) != 0;
    return ((r > 0.0) != odd) ? 1 : -1;
}
__device__ __forceinline__ int sin_sign(float) { return 0; }  // (fast mode: the sines themselves)
template <typename T>
__device__ __forceinline__ int resolve_checkered(const DScene<T>& S, int mi, T u, T v, T f) {
    DMaterial<T> m = S.materials[mi];
    while (m.type == RT_MAT_CHECKERED) {
        const T a = m.param * f * u, b = m.param * f * v;
        const int sa = sin_sign(a), sb = sin_sign(b);
        const bool odd = (sa != 0 && sb != 0) ? (sa != sb) : checker_is_odd_by_sines(a, b);
        mi = odd ? m.odd : m.even;
        m = S.materials[mi];
    }
    return mi;
}

// One bounce of cast_ray for one path (src/hitting.rs:11-31) after its closest-hit query: adds what
// the path sees (light / sky) to its sample, and scatters.  mf = material | front_face << 31, or
// RT_MISS.  Returns true if the path goes on: nr is the scattered ray, thr the new throughput.
// EVERY lane of the wave must call this (active = false: no path in this lane): the unit-sphere draw in the middle
// is made by the wave together (wave_first_in_unit_sphere).  UNIFORM_DEPTH: depth is the same in every lane.
#ifndef RT_COOP_DRAW
#define RT_COOP_DRAW 1  // 0: every lane runs its own rejection loop (A/B builds)
#endif
template <typename T, bool UNIFORM_DEPTH>
__device__ __forceinline__ bool shade_path(const DScene<T>& S, const DBatch& B, bool active, const Ray<T>& ray, V3<T>& thr,
                                           uint32_t pid, uint32_t mf, T t, V3<T> n, T u, T v, T* sample_rgb, uint32_t depth,
                                           uint32_t max_bounces, Ray<T>& nr) {
    const T PI = (T)3.14159265358979323846264338327950288;
    bool alive = false;
    V3<T> add = mk((T)0.0, (T)0.0, (T)0.0);
    bool contributes = false;
    const bool hit = active && mf != RT_MISS;
    const bool front = (mf & 0x80000000u) != 0;
    DMaterial<T> m;  // the material that scatters (hit lanes)
    m.type = -1;
    uint32_t pixel = 0, sample = 0;
    if (active && mf == RT_MISS) {  // sky(ray), src/hitting.rs:29
        V3<T> c;
        const DSky<T>& sky = *S.sky;
        if (sky.gradient) {  // src/camera.rs:78-86
            T gradient_pos = dot(ld3(sky.unit_dir), unit_vector(ray.d));
            T g = (T)0.5 * (gradient_pos + (T)1.0);
            c = (T)1.0 * (((T)1.0 - g) * ld3(sky.colour0) + g * ld3(sky.colour1));
        } else {
            c = ld3(sky.colour);
        }
        add = coeff(thr, c);
        contributes = true;
    }
    if (hit) {
        int mi = (int)(mf & 0x7fffffffu);
        // emitted(): Checkered picks its part with sin(k u) sin(k v) - no PI (src/materials.rs:175-183)
        {
            int me = mi;
            DMaterial<T> em = S.materials[me];
            // (a checker none of whose parts is a light emits nothing: m.image == 0 then, set at upload,
            // and the two f64 sines are skipped)
            if (em.type == RT_MAT_CHECKERED && em.image != 0) {
                me = resolve_checkered(S, me, u, v, (T)1.0);
                em = S.materials[me];
            }
            if (em.type == RT_MAT_DIFFUSE_LIGHT && front) {  // src/materials.rs:129-136
                add = coeff(thr, texture_value(S, em, u, v));
                contributes = true;
            }
        }
        // scatter(): Checkered picks with sin(k PI u) sin(k PI v) (src/materials.rs:166-174)
        m = S.materials[mi];
        if (m.type == RT_MAT_CHECKERED) {
            mi = resolve_checkered(S, mi, u, v, PI);
            m = S.materials[mi];
        }
        path_ids(B, pid, &pixel, &sample);
    }
    // Lambertian, Metal and Isotropic each begin with one random_in_unit_sphere (src/materials.rs:33, 58, 148) on the
    // bounce's fresh stream and draw nothing after it: drawn here, once, by the wave together.
    const bool draws = m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_METAL || m.type == RT_MAT_ISOTROPIC;
    V3<T> in_sphere;
    if constexpr (RT_COOP_DRAW != 0) {
        in_sphere = wave_first_in_unit_sphere<T, UNIFORM_DEPTH>((uint32_t)B.seed, (uint32_t)(B.seed >> 32) ^ DOMAIN_MATERIAL, pixel,
                                                                sample, depth, draws);
    } else {
        in_sphere = mk((T)0.0, (T)0.0, (T)0.0);
        if (draws) {
            Rng<T> rng(B.seed, DOMAIN_MATERIAL, pixel, sample, depth);
            in_sphere = random_in_unit_sphere(rng);
        }
    }
    if (hit) {
        V3<T> p = ray.at(t);  // HitRecord::new: intersection = ray.at(distance)
        V3<T> attenuation = mk((T)1.0, (T)1.0, (T)1.0);
        V3<T> dir = n;
        bool scattered = false, metal = false;
        switch (m.type) {
            case RT_MAT_LAMBERTIAN: {  // src/materials.rs:27-42
                dir = n + unit_vector(in_sphere);
                if (near_zero(dir)) dir = n;
                attenuation = texture_value(S, m, u, v);
                scattered = true;
                break;
            }
            case RT_MAT_METAL: {  // :53-66
                V3<T> reflected = reflect(unit_vector(ray.d), n);
                dir = reflected + m.param * in_sphere;
                attenuation = ld3(m.colour);
                metal = true;
                scattered = true;
                break;
            }
            case RT_MAT_DIELECTRIC: {  // :76-101
                T ratio = front ? (T)1.0 / m.param : m.param;
                V3<T> ud = unit_vector(ray.d);
                T cos_theta = fmin(dot(-ud, n), (T)1.0);
                T sin_theta = sqrt((T)1.0 - cos_theta * cos_theta);
                bool cannot_refract = ratio * sin_theta > (T)1.0;
                Rng<T> rng(B.seed, DOMAIN_MATERIAL, pixel, sample, depth);
                T random_fraction = rng.range((T)0.0, (T)1.0);
                dir = (cannot_refract || reflectance(cos_theta, ratio) > random_fraction) ? reflect(ud, n)
                                                                                         : refract(ud, n, ratio);
                scattered = true;
                break;
            }
            case RT_MAT_ISOTROPIC: {  // :146-153
                dir = in_sphere;
                attenuation = texture_value(S, m, u, v);
                scattered = true;
                break;
            }
            default: break;  // DiffuseLight::scatter -> None (:126-128)
        }
        // cast_ray(new_ray, .., bounces - 1) returns black once bounces hits 0 (src/hitting.rs:17-19).
        V3<T> new_dir = unit_vector(dir);  // Ray::new normalises (src/math.rs:208-214)
        if (metal) scattered = dot(new_dir, n) > (T)0.0;  // absorbed otherwise (src/materials.rs:61-65)
        if (scattered && depth + 1 < max_bounces) {
            alive = true;
            nr.o = p;
            nr.d = new_dir;
            nr.time = ray.time;
            thr = coeff(thr, attenuation);
        }
    }
    // Where a path can only add once, when it ends (DScene::emit_once: every scene without a light inside a checker),
    // its sample is the addend itself - or zero if it ends in the dark: ONE plain store per path, at its end, no read of
    // the line (cornell: 11 % of k_bounce's HBM-side reads) and no clearing of the sample buffer beforehand (24 B per
    // path and a launch per batch: the host skips its memset for such scenes).  Otherwise the buffer starts at zero.
    if (active) {
        if (S.emit_once) {
            if (!alive) {
                T* dst = sample_rgb + (uint64_t)pid * 3;
                dst[0] = add.x; dst[1] = add.y; dst[2] = add.z;  // (add is zero unless the path saw a light or the sky)
            }
        } else if (contributes) {
            T* dst = sample_rgb + (uint64_t)pid * 3;
            dst[0] += add.x; dst[1] += add.y; dst[2] += add.z;
        }
    }
    return alive;
}

// Compaction: the survivors of input tile t are packed at the front of output tile t.  One
// returning atomic per wave and tile on that tile's own counter (thousands of counters spread
// over the L2 channels; a single global counter saturates at ~90 atomics/us on MI355X).  A
// wave's items span one tile, rarely two; lanes are placed by ballot prefix.  Every lane of the
// wave must call this.
template <typename T>
__device__ __forceinline__ void compact_store(bool alive, uint32_t tile, const Ray<T>& nr, V3<T> thr, uint32_t pid,
                                              const DQueue<T>& qout, uint32_t* tile_counts_out) {
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(alive);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        uint32_t t_lead = __shfl(tile, leader);
        unsigned long long same = __ballot(alive && tile == t_lead);
        uint32_t start = 0;
        if ((int)lane == leader) start = atomicAdd(&tile_counts_out[t_lead], (uint32_t)__popcll(same));
        start = __shfl(start, leader);
        if (alive && tile == t_lead) {
            uint32_t o = t_lead * TILE + start + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            store_ray(qout, o, nr);
            qout.tr()[o] = thr.x; qout.tg()[o] = thr.y; qout.tb()[o] = thr.z;
            qout.pid[o] = pid;
        }
        todo &= ~same;
    }
}

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 1
#endif
template <typename T>
__global__ void __launch_bounds__(BLOCK, RT_SHADE_WAVES) k_shade(DScene<T> S, DBatch B, DQueue<T> qin, DHits<T> h, DQueue<T> qout,
                                                  T* sample_rgb, uint32_t depth, uint32_t max_bounces,
                                                  const uint32_t* __restrict__ tile_offsets, uint32_t* tile_counts_out,
                                                  uint32_t n_tiles) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t total = tile_offsets[n_tiles];
    // The loop bound is wave-uniform so that every lane reaches the ballots of compact_store.
    for (uint32_t base = blockIdx.x * BLOCK + (threadIdx.x & ~63u); base < total; base += gridDim.x * BLOCK) {
        const uint32_t d = base + lane;
        const bool valid = d < total;
        uint32_t tile = 0;
        const uint32_t i = valid ? slot_of(tile_offsets, n_tiles, d, base, &tile) : 0u;
        bool alive = false;
        Ray<T> nr;
        V3<T> thr;
        uint32_t pid = 0;
        Ray<T> ray;
        ray.o = ray.d = thr = mk((T)0.0, (T)0.0, (T)0.0);
        ray.time = (T)0.0;
        uint32_t mf = RT_MISS;
        T t = 0, u = 0, v = 0;
        V3<T> n = mk((T)0.0, (T)0.0, (T)0.0);
        if (valid) {
            ray = load_ray(qin, i);
            pid = qin.pid[i];
            thr = mk(qin.tr()[i], qin.tg()[i], qin.tb()[i]);
            mf = h.mat_front[i];
            if (mf != RT_MISS) {
                t = h.t()[i];
                if (S.needs_uv) {
                    u = h.u()[i]; v = h.v()[i];
                }
                n = mk(h.nx()[i], h.ny()[i], h.nz()[i]);
            }
        }
        alive = shade_path<T, true>(S, B, valid, ray, thr, pid, mf, t, n, u, v, sample_rgb, depth, max_bounces, nr);
        compact_store(alive, tile, nr, thr, pid, qout, tile_counts_out);
    }
}

// extend + shade of one bounce in one kernel (the render path): the hit record never leaves the
// registers, the ray is read once, and a bounce is one launch.  rt_trace keeps k_extend.
// CAN_TAIL = false: compiled for per-bounce launches only (the sweep kernels, which run every bounce of
// cornell-like scenes): the bounce loop, its counters and the second ray are gone, and with them the last
// register spills of the lean kernel.
// CAN_TAIL = true is the tail launch in the product (a few thousand waves, fewer than two per SIMD): it is
// compiled for 2 waves per SIMD, i.e. up to 256 VGPRs, so that the walk + shading of a tree scene does not spill.
// FIRST = true (the first bounce of a batch on the fused path): the camera rays are made here instead of being
// read from a queue k_raygen has written - 84 B per path that are neither written nor read, and one launch fewer
// (cornell: k_raygen was 0.65 ms of a 19.5 ms image, globe: of 3.0 ms).  Slot d of the dense batch is path d's place.
template <typename T, int STACK, int MODE, bool CAN_TAIL, bool FIRST = false>
__global__ void __launch_bounds__(BLOCK, CAN_TAIL ? tail_waves(MODE) : bounce_waves(MODE, sizeof(T))) k_bounce(DScene<T> S, DSmallScene small, DBatch B, DQueue<T> qin, DQueue<T> qout,
                                                                    T* sample_rgb, uint32_t depth0, uint32_t max_bounces,
                                                                    const uint32_t* __restrict__ tile_offsets,
                                                                    uint32_t* tile_counts_out, uint32_t n_tiles, DTail tail, uint32_t n_first) {
    __shared__ int32_t lds_stack[STACK * BLOCK];
    constexpr bool LDS_SCENE = MODE == WALK_OWN_BOX_LDS;
    __shared__ uint4 lds_scene[LDS_SCENE ? RT_LDS_SCENE_BYTES / 16 : 1];
    int32_t stack_spill[StackFor<MODE>::SPILL];
    typename StackFor<MODE>::type st = StackFor<MODE>::make(lds_stack + threadIdx.x, S.stack_cap > 0 && S.stack_cap < STACK ? S.stack_cap : STACK, stack_spill);
    // tail.ray_total != nullptr: the paths that are left are carried through ALL their remaining bounces here, each
    // at its own pace (a launch per bounce would wait for the slowest ray of every bounce: on the knight ~0.5 ms per
    // bounce for a ray that grazes the mesh, 40 times over).  Same streams, same arithmetic, same image.
    const bool run_tail = CAN_TAIL && tail.ray_total != nullptr;
    DScene<T> W = S;  // the traversal's view of the scene (records in LDS for small worlds)
    if (LDS_SCENE) stage_small_scene(W, small, lds_scene);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t total = FIRST ? n_first : tile_offsets[n_tiles];
    // One bounce of one path: the closest-hit query, the hit record, the shading.
    // (known != nullptr: the closest-hit query has been answered already, by the lane's group)
    struct Answer { bool hit; T t; Winner w; };
    // (called by EVERY lane of the wave, active or not: the shading's draw is the wave's common work, shade_path)
    auto bounce_once = [&](bool active, const Ray<T>& ray, V3<T>& thr, uint32_t pid, uint32_t depth, Ray<T>& nr, const Answer* known, auto uniform_depth) -> bool {
        T t = 0, u = 0, v = 0;
        V3<T> n = mk((T)0.0, (T)0.0, (T)0.0);
        uint32_t mf = RT_MISS;
        if (active) {
            const PathRef path = path_of(B, pid);
            Winner w;
            bool hit;
            if (known) {
                hit = known->hit; t = known->t; w = known->w;
            } else {
                hit = closest_hit<T, MODE>(W, ray, (T)0.001, Lim<T>::inf(), st, path, depth, &t, &w);  // src/hitting.rs:20-21
            }
            if (hit) {
                const Surface<T> sf = make_surface<T, WalkFeat<MODE>::value>(W, ray, t, w, S.needs_uv != 0);
                n = sf.normal; u = sf.u; v = sf.v;
                mf = (uint32_t)sf.material | (sf.front_face ? 0x80000000u : 0u);
            }
        }
        return shade_path<T, decltype(uniform_depth)::value>(S, B, active, ray, thr, pid, mf, t, n, u, v, sample_rgb, depth, max_bounces, nr);
    };
    if (!run_tail) {
        for (uint32_t base = blockIdx.x * BLOCK + (threadIdx.x & ~63u); base < total; base += gridDim.x * BLOCK) {
            const uint32_t d = base + lane;
            const bool valid = d < total;
            uint32_t tile = 0;
            bool alive = false;
            Ray<T> nr;
            V3<T> thr;
            uint32_t pid = 0;
            Ray<T> ray;
            ray.o = ray.d = thr = mk((T)0.0, (T)0.0, (T)0.0);
            ray.time = (T)0.0;
            if constexpr (FIRST) {
                tile = d / TILE;
                if (valid) {
                    ray = primary_ray(S, B, d, total, &pid);
                    thr = mk((T)1.0, (T)1.0, (T)1.0);
                }
            } else {
                const uint32_t i = valid ? slot_of(tile_offsets, n_tiles, d, base, &tile) : 0u;
                if (valid) {
                    ray = load_ray(qin, i);
                    pid = qin.pid[i];
                    thr = mk(qin.tr()[i], qin.tg()[i], qin.tb()[i]);
                }
            }
            alive = bounce_once(valid, ray, thr, pid, depth0, nr, nullptr, std::true_type{});
            compact_store(alive, tile, nr, thr, pid, qout, tile_counts_out);
        }
        return;
    }
    if constexpr (CAN_TAIL) {
        // The paths are handed out from a launch-wide cursor, a share of CHUNK at a time per wave (one atomic per
        // share: every wave of the launch shares the cursor, and atomics on one address are served one by one); a
        // lane whose path has ended takes the next one of the share, so the waves stay full until the queue is
        // empty, and the launch ends as long after that as its longest path.
        uint32_t extra_queries = 0, deepest = depth0, taken = 0;
        constexpr bool COOP_MODE = walks_node4(MODE);  // (meshes: the mesh walk is shared out; without: the world walk)
        const uint32_t CHUNK = max(64u, min(1024u, (total / (gridDim.x * (BLOCK / 64) * 8u)) & ~63u));
        uint32_t chunk_lo = 0, chunk_hi = 0;  // (wave-uniform) what is left of the wave's share
        bool exhausted = false;               // (wave-uniform)
        bool alive = false;
        Ray<T> ray;
        V3<T> thr;
        uint32_t pid = 0, depth = depth0;
        ray.o = ray.d = thr = mk((T)0.0, (T)0.0, (T)0.0);
        ray.time = (T)0.0;
        RT_TAIL_WAVE_BEGIN;
        for (;;) {
            while (!exhausted) {
                const uint64_t idle = __ballot(!alive);
                if (idle == 0) break;
                if (chunk_lo >= chunk_hi) {
                    uint32_t fresh = 0;
                    if (lane == 0) fresh = atomicAdd(tail.cursor, CHUNK);
                    fresh = (uint32_t)__shfl((int)fresh, 0);
                    if (fresh >= total) {
                        exhausted = true;
                        break;
                    }
                    chunk_lo = fresh;
                    chunk_hi = min(fresh + CHUNK, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t avail = chunk_hi - chunk_lo;
                if (!alive && rank < avail) {
                    uint32_t tile;
                    const uint32_t i = slot_of(tile_offsets, n_tiles, chunk_lo + rank, chunk_lo, &tile);
                    ray = load_ray(qin, i);
                    pid = qin.pid[i];
                    thr = mk(qin.tr()[i], qin.tg()[i], qin.tb()[i]);
                    depth = depth0;
                    alive = true;
                    taken++;
                }
                chunk_lo += min((uint32_t)__popcll(idle), avail);
            }
            RT_TAIL_QUEUE_DRY(exhausted);
            if (__ballot(alive) == 0) break;
            RT_TAIL_ITER_BEGIN(alive);
            // A wave with a few live lanes left (the end of the launch is waves of ONE path each, walking the mesh at
            // one lane's pace): each live path's query is answered by a group of 64 / 32 / 16 / 8 lanes.
            Answer ans;
            bool grouped = false;
            if constexpr (MODE == WALK_SWEEP_LEAN) {
                // a swept world of rect lists: a wave down to a few paths shares each path's rects out among a group of lanes
                const uint64_t live = __ballot(alive);
                const int n_live = __popcll(live);
                if (__builtin_expect(S.sweep_items != nullptr && n_live <= S.coop, 0)) {  // (wave-uniform)
                    const int G = n_live <= 1 ? 64 : n_live <= 2 ? 32 : n_live <= 4 ? 16 : n_live <= 8 ? 8 : n_live <= 16 ? 4 : 2;
                    const int group = (int)lane / G;
                    uint64_t rest = live;
                    for (int k = 0; k < group && rest; k++) rest &= rest - 1;
                    const bool has_ray = rest != 0;
                    const int owner = has_ray ? __ffsll((long long)rest) - 1 : 0;  // the group-th live lane
                    Ray<T> gr;
                    gr.o = mk(__shfl(ray.o.x, owner), __shfl(ray.o.y, owner), __shfl(ray.o.z, owner));
                    gr.d = mk(__shfl(ray.d.x, owner), __shfl(ray.d.y, owner), __shfl(ray.d.z, owner));
                    gr.time = __shfl(ray.time, owner);
                    T g_t = 0;
                    Winner g_w{-1, -1};
                    bool g_hit = false;
                    // (every lane runs the group's reduction; a group without a path tests nothing)
                    g_hit = sweep_hit_group<T>(W, gr, has_ray ? (T)0.001 : Lim<T>::inf(), G, &g_t, &g_w);
                    const int from = __popcll(live & ((1ull << lane) - 1ull)) * G;  // the k-th live lane reads lane 0 of group k
                    ans.hit = __shfl((int)g_hit, from & 63) != 0;
                    ans.t = __shfl(g_t, from & 63);
                    ans.w.obj = __shfl(g_w.obj, from & 63);
                    ans.w.prim = __shfl(g_w.prim, from & 63);
                    grouped = true;
                }
            }
            if constexpr (COOP_MODE) {
                const uint64_t live = __ballot(alive);
                const int n_live = __popcll(live);
                if (__builtin_expect(n_live <= S.coop, 0)) {  // (wave-uniform; rare next to the full waves: kept out of their registers' way)
                    const int G = n_live <= 1 ? 64 : n_live <= 2 ? 32 : n_live <= 4 ? 16 : n_live <= 8 ? 8 : n_live <= 16 ? 4 : 2;
                    const int group = (int)lane / G;
                    uint64_t rest = live;
                    for (int k = 0; k < group && rest; k++) rest &= rest - 1;
                    const bool has_ray = rest != 0;
                    const int owner = has_ray ? __ffsll((long long)rest) - 1 : 0;  // the group-th live lane
                    Ray<T> gr;
                    gr.o = mk(__shfl(ray.o.x, owner), __shfl(ray.o.y, owner), __shfl(ray.o.z, owner));
                    gr.d = mk(__shfl(ray.d.x, owner), __shfl(ray.d.y, owner), __shfl(ray.d.z, owner));
                    gr.time = __shfl(ray.time, owner);
                    const uint32_t g_pid = (uint32_t)__shfl((int)pid, owner), g_depth = (uint32_t)__shfl((int)depth, owner);
                    T g_t = 0;
                    Winner g_w{-1, -1};
                    bool g_hit = false;
                    if (has_ray) {
                        GroupStack gs;
                        gs.base = st.base; gs.stride = st.stride; gs.cap = st.cap; gs.overflow = st.overflow; gs.group = G;
                        g_hit = closest_hit<T, MODE>(W, gr, (T)0.001, Lim<T>::inf(), gs, path_of(B, g_pid), g_depth, &g_t, &g_w);
                    }
                    // back to the path's own lane: the k-th live lane reads lane 0 of group k
                    const int from = __popcll(live & ((1ull << lane) - 1ull)) * G;
                    ans.hit = __shfl((int)g_hit, from & 63) != 0;
                    ans.t = __shfl(g_t, from & 63);
                    ans.w.obj = __shfl(g_w.obj, from & 63);
                    ans.w.prim = __shfl(g_w.prim, from & 63);
                    grouped = true;
                }
            }
            {
                const bool was_alive = alive;
                Ray<T> nr = ray;
                if (was_alive) {
                    if (depth > depth0) extra_queries++;
                    deepest = max(deepest, depth + 1);
                }
                alive = bounce_once(was_alive, ray, thr, pid, depth, nr, grouped ? &ans : nullptr, std::false_type{});
                if (was_alive) {
                    ray = nr;  // the scattered ray is the next bounce's query
                    depth++;
                }
            }
            RT_TAIL_ITER_END;
        }
        RT_TAIL_WAVE_END;
        // rays and depth statistics of the bounces that were not launches
        for (int off = 32; off > 0; off >>= 1) {
            extra_queries += __shfl_xor(extra_queries, off);
            taken += __shfl_xor(taken, off);
            deepest = max(deepest, (uint32_t)__shfl_xor(deepest, off));
        }
        if (lane == 0) {
            if (extra_queries) atomicAdd(tail.ray_total, (unsigned long long)extra_queries);
            if (taken) atomicAdd(tail.ray_total + 1, (unsigned long long)taken);
            atomicMax(tail.deepest, deepest);
        }
    }
}

// ------------------------------------------------------------------ rt_hit::second_t (rt_trace only)
// The nearest distance >= tmin at which the ray meets a primitive other than the winner (owner, sub), whatever tmax
// is: what tells an exact tie (second_t == t), where the reference's answer rests on its list order (first wins,
// src/hitting.rs:94) and leaf order (later wins, :219).  A second walk of its own, so that the render kernels carry
// nothing for it: every primitive whose box the ray's line can reach is asked for its candidate distance - a rect,
// triangle or plane where the line crosses it, a sphere at its first root >= tmin, a medium where its draw puts the
// scatter event - and the smallest one from another primitive is kept (the walk is bounded by it, nothing else).
// Stack: a private array (the binary trees are at most 64 deep each: world, mesh, a medium boundary's mesh).
struct PrivStack {
    static constexpr bool COOP = false;
    int32_t* base;
    __device__ __forceinline__ void put(int e, int32_t v) { base[e] = v; }
    __device__ __forceinline__ int32_t get(int e) const { return base[e]; }
    __device__ __forceinline__ bool room(int, int) const { return false; }
    __device__ __forceinline__ void put_near(int e, int32_t v) { base[e] = v; }
};
template <typename T>
struct Second {
    T best, min_dist;
    int owner, sub;
    __device__ __forceinline__ void consider(T t, int o, int s) {  // (NaN fails t >= min_dist)
        if (t >= min_dist && (o != owner || s != sub) && t < best) best = t;
    }
};
// Depth-first over a DNode2 tree from `root`: leaf(payload, sp) for every leaf whose padded box the ray reaches
// between min_dist and the best candidate so far.
template <typename T, typename LEAF>
__device__ __forceinline__ void walk2_all(const DScene<T>& S, const Ray<T>& r, int root, float shift, PrivStack st, int sp0, const Second<T>& sc, LEAF leaf) {
    const RayF f = ray_f32(r, shift);
    const float tmin32 = round_down(sc.min_dist - (T)shift);
    int sp = sp0;
    int32_t next = root;
    for (;;) {
        if (next == RT_EMPTY) {
            if (sp == sp0) return;
            next = st.get(--sp);
        }
        if (next < 0) {
            leaf(-1 - next, sp);
            next = RT_EMPTY;
            continue;
        }
        const Node2Regs n = load_node2(S.nodes2, next);
        const float bound32 = round_up(sc.best - (T)shift);
        const int32_t c0 = n.child(0), c1 = n.child(1);
        float e0, e1;
        const bool h0 = c0 != RT_EMPTY && box2_hit(n.lo(0), n.hi(0), f, tmin32, bound32, &e0);
        const bool h1 = c1 != RT_EMPTY && box2_hit(n.lo(1), n.hi(1), f, tmin32, bound32, &e1);
        next = RT_EMPTY;
        if (h0 && h1) {
            next = c0;
            st.put(sp++, c1);
        } else if (h0) {
            next = c0;
        } else if (h1) {
            next = c1;
        }
    }
}
template <typename T>
__device__ void object_candidates(const DScene<T>& S, int oi, const Ray<T>& world_ray, PrivStack st, int sp, float shift, PathRef path,
                                  Second<T>& sc) {
    const DObject ob = S.objects[oi];
    const T inf = Lim<T>::inf();
    if (ob.kind == RT_OBJ_MEDIUM) {  // the scatter event itself; the boundary's own crossings are not candidates
        T t;
        int prim;
        if (medium_hit<T, false, (FEAT_ALL | FEAT_NESTED_MEDIA)>(S, ob, world_ray, sc.min_dist, inf, st, sp, shift, path, 0u, &t, &prim))
            sc.consider(t, S.media[prim].owner, 0);
        return;
    }
    Ray<T> r = world_ray;
    if (ob.xform >= 0) r = to_object_space(S.xforms[ob.xform], world_ray);
    switch (ob.kind) {
        case RT_OBJ_RECTS:
            for (int k = 0; k < ob.count; k++) {
                const DRect<T> q = S.rects[ob.first + k];
                const int axis = q.axis, ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
                const T t = (q.k - comp(r.o, axis)) / comp(r.d, axis);
                const T a = comp(r.o, ia) + t * comp(r.d, ia), b = comp(r.o, ib) + t * comp(r.d, ib);
                if (!(a < q.a0 || a > q.a1 || b < q.b0 || b > q.b1)) sc.consider(t, q.owner, q.sub);
            }
            break;
        case RT_OBJ_SPHERE: {
            const DSphere<T> s = S.spheres[ob.first];
            T t;
            if (distance_to_sphere(r, ld3(s.centre), s.radius, sc.min_dist, inf, &t)) sc.consider(t, s.owner, 0);
            break;
        }
        case RT_OBJ_MOVING_SPHERE: {
            const DMovingSphere<T> s = S.moving_spheres[ob.first];
            T t;
            if (distance_to_sphere(r, moving_centre(s, r.time), s.radius, sc.min_dist, inf, &t)) sc.consider(t, s.owner, 0);
            break;
        }
        case RT_OBJ_TRIANGLE: {
            const DTriangle<T> tr = S.triangles[ob.first];
            V3<T> sol;
            if (line_plane_collision(r.o, r.d, ld3(tr.point), ld3(tr.vec1), ld3(tr.vec2), &sol) &&
                !(sol.y < (T)0.0 || sol.z < (T)0.0 || sol.y + sol.z > (T)1.0))
                sc.consider(sol.x, tr.owner, tr.sub);
            break;
        }
        case RT_OBJ_PLANE: {
            const DPlane<T> pl = S.planes[ob.first];
            V3<T> sol;
            if (line_plane_collision(r.o, r.d, ld3(pl.point), ld3(pl.vec1), ld3(pl.vec2), &sol)) sc.consider(sol.x, pl.owner, 0);
            break;
        }
        case RT_OBJ_BLAS:
            walk2_all(S, r, ob.first, shift, st, sp, sc, [&](int tri, int) {
                const DTriangle<T> tr = S.triangles[tri];
                V3<T> sol;
                if (line_plane_collision(r.o, r.d, ld3(tr.point), ld3(tr.vec1), ld3(tr.vec2), &sol) &&
                    !(sol.y < (T)0.0 || sol.z < (T)0.0 || sol.y + sol.z > (T)1.0))
                    sc.consider(sol.x, tr.owner, tr.sub);
            });
            break;
        default: break;
    }
}
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_second(DScene<T> S, DBatch B, DQueue<T> q, DHits<T> h, const T* tmin_arr, uint32_t n) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const T inf = Lim<T>::inf();
    if (h.mat_front[i] == RT_MISS) {
        h.second[i] = inf;
        return;
    }
    int32_t stack[192];
    PrivStack st{stack};
    const Ray<T> r = load_ray(q, i);
    Second<T> sc{inf, tmin_arr[i], h.owner[i], h.sub[i]};
    const float shift = far_shift(S, r);
    const PathRef path = path_of(B, q.pid[i]);
    for (int e = 0; e < S.n_top; e++) {
        const int entry = S.top_list[e];
        if (entry != RT_TOP_BVH) object_candidates(S, entry, r, st, 0, shift, path, sc);
        else if (S.tlas_root >= 0)
            walk2_all(S, r, S.tlas_root, shift, st, 0, sc, [&](int oi, int sp) { object_candidates(S, oi, r, st, sp, shift, path, sc); });
    }
    h.second[i] = sc.best;
}

// colour += cast_ray(..) over the samples of a batch, in sample order (src/main.rs:131-137).
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_accumulate(const T* sample_rgb, uint32_t npix, uint32_t n_samples, double* accum) {
    uint32_t e = blockIdx.x * BLOCK + threadIdx.x;  // element = pixel * 3 + channel
    if (e >= npix * 3u) return;
    double acc = accum[e];
    for (uint32_t s = 0; s < n_samples; s++) acc += (double)sample_rgb[(uint64_t)s * npix * 3u + e];
    accum[e] = acc;
}

#if RT_PART_HAS(2)
__global__ void __launch_bounds__(BLOCK) k_quantize(const double* accum, uint64_t n, uint32_t spp_total, uint8_t* rgb8) {
    uint64_t e = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= n) return;
    double c = accum[e] / (double)spp_total;              // colour /= samples_per_pixel (src/main.rs:138)
    c = sqrt(c);                                          // gamma 2 (:140-141)
    c = fabs(c);                                          // colour_to_raw (:172-177)
    c = c < 0.0 ? 0.0 : c > 0.999 ? 0.999 : c;            // clamp() passes NaN through (src/math.rs:220-228)
    double f = floor(255.0 * c);
    rgb8[e] = f != f ? 0 : f <= 0.0 ? 0 : f >= 255.0 ? 255 : (uint8_t)f;  // `as u8`: saturating, NaN -> 0
}
#endif

// rt_trace plumbing: AoS f64 host layout <-> SoA queues.
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_load_rays(const double* rays9, uint32_t n, DQueue<T> q, T* tmin, T* tmax,
                                                      uint32_t* tile_counts) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (i % TILE == 0) tile_counts[i / TILE] = min(TILE, n - i);
    const double* s = rays9 + (uint64_t)i * 9;
    Ray<T> r;
    r.o = mk((T)s[0], (T)s[1], (T)s[2]);
    r.d = unit_vector(mk((T)s[3], (T)s[4], (T)s[5]));  // Ray::new normalises (src/math.rs:208-214)
    r.time = (T)s[6];
    store_ray(q, i, r);
    q.pid[i] = i;
    tmin[i] = (T)s[7];
    tmax[i] = (T)s[8];
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_store_hits(DQueue<T> q, DHits<T> h, uint32_t n, rt_hit* out) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    rt_hit o;
    uint32_t mf = h.mat_front[i];
    if (mf == RT_MISS) {
        o.hit = 0; o.owner = -1; o.sub = 0; o.world_index = -1; o.front_face = 0; o.material = -1;
        o.t = 0; o.p[0] = o.p[1] = o.p[2] = 0; o.n[0] = o.n[1] = o.n[2] = 0; o.u = o.v = 0;
        o.second_t = (double)Lim<double>::inf(); o.inst_id = -1;
    } else {
        Ray<T> r = load_ray(q, i);
        T t = h.t()[i];
        V3<T> p = r.at(t);
        o.hit = 1;
        o.owner = h.owner[i]; o.sub = h.sub[i]; o.world_index = h.world_index[i];
        o.front_face = (mf >> 31) & 1;
        o.material = (int32_t)(mf & 0x7fffffffu);
        o.t = (double)t;
        o.p[0] = (double)p.x; o.p[1] = (double)p.y; o.p[2] = (double)p.z;
        o.n[0] = (double)h.nx()[i]; o.n[1] = (double)h.ny()[i]; o.n[2] = (double)h.nz()[i];
        o.u = (double)h.u()[i]; o.v = (double)h.v()[i];
        o.second_t = (double)h.second[i];
        o.inst_id = h.inst[i];
    }
    o.pad = 0;
    out[i] = o;
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_store_rays(DQueue<T> q, uint32_t n, double* rays9) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double* o = rays9 + (uint64_t)i * 9;
    o[0] = (double)q.ox()[i]; o[1] = (double)q.oy()[i]; o[2] = (double)q.oz()[i];
    o[3] = (double)q.dx()[i]; o[4] = (double)q.dy()[i]; o[5] = (double)q.dz()[i];
    o[6] = (double)q.time()[i];
    o[7] = 0.001;
    o[8] = (double)Lim<double>::inf();
}

// ------------------------------------------------------------------ launch wrappers
static inline int blocks_for(uint64_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

template <typename T>
void Launch<T>::raygen(const DScene<T>& S, const DBatch& B, DQueue<T> q, uint32_t n_paths, uint32_t* tile_counts,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_raygen<T>, dim3(blocks_for(n_paths ? n_paths : 1)), dim3(BLOCK), 0, st, S, B, q, n_paths, tile_counts);
}
template <typename T>
void Launch<T>::extend(const DScene<T>& S, const DSmallScene& small, const DBatch& B, DQueue<T> q, DHits<T> h, const T* tmin, const T* tmax,
                       uint32_t depth, const uint32_t* tile_offsets, uint32_t n_tiles, int stack, int grid_blocks,
                       uint32_t* cursor, int refill, hipStream_t st) {
    if (S.n_sweep > 0 && S.lean == 1)
        hipLaunchKernelGGL((k_extend<T, 1, WALK_SWEEP_LEAN>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else if (S.n_sweep > 0 && S.lean == 2)
        hipLaunchKernelGGL((k_extend<T, 1, WALK_SWEEP_FLAT>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else if (S.n_sweep > 0 && stack <= 16)
        hipLaunchKernelGGL((k_extend<T, 16, WALK_SWEEP>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else if (S.n_sweep > 0)
        hipLaunchKernelGGL((k_extend<T, 64, WALK_SWEEP>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else if (small.blob && stack <= 16)
        hipLaunchKernelGGL((k_extend<T, 16, WALK_OWN_BOX_LDS>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
#define RT_EXTEND(STACK, MODE)                                                                                           \
    hipLaunchKernelGGL((k_extend<T, STACK, MODE>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, \
                       tile_offsets, n_tiles)
#define RT_EXTEND4(MODE)                       \
    do {                                       \
        if (stack <= 16) RT_EXTEND(16, MODE);  \
        else if (stack <= 32) RT_EXTEND(32, MODE); \
        else RT_EXTEND(64, MODE);              \
    } while (0)
#define RT_REFILL(STACK, MODE)                                                                                                \
    hipLaunchKernelGGL((k_extend_refill<T, STACK, MODE>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, B, q, h, tmin, tmax, depth, \
                       tile_offsets, n_tiles, cursor, (uint32_t)refill)
#define RT_REFILL4(MODE)                       \
    do {                                       \
        if (stack <= 16) RT_REFILL(16, MODE);  \
        else if (stack <= 32) RT_REFILL(32, MODE); \
        else RT_REFILL(64, MODE);              \
    } while (0)
    else if (S.nodes4 && refill > 0 && S.has_meshes && S.has_media) RT_REFILL4(WALK_NODE2_MESH4);
    else if (S.nodes4 && refill > 0 && S.has_meshes) RT_REFILL4(WALK_NODE4_MESH);
    else if (S.nodes4 && refill > 0 && S.has_media) RT_REFILL4(WALK_NODE4_MEDIA);
    else if (S.nodes4 && refill > 0) RT_REFILL4(WALK_NODE4_PLAIN);
    else if (S.nodes4 && S.has_meshes && S.has_media) RT_EXTEND4(WALK_NODE2_MESH4);
    else if (S.nodes4 && S.has_meshes) RT_EXTEND4(WALK_NODE4_MESH);
    else if (S.nodes4 && S.has_media) RT_EXTEND4(WALK_NODE4_MEDIA);
    else if (S.nodes4) RT_EXTEND4(WALK_NODE4_PLAIN);
#undef RT_REFILL4
#undef RT_REFILL
#undef RT_EXTEND4
#undef RT_EXTEND
    else if (stack <= 16)
        hipLaunchKernelGGL((k_extend<T, 16, WALK_NODE2>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else if (stack <= 32)
        hipLaunchKernelGGL((k_extend<T, 32, WALK_NODE2>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
    else
        hipLaunchKernelGGL((k_extend<T, 64, WALK_NODE2>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, q, h, tmin, tmax, depth, tile_offsets, n_tiles);
}
template <typename T>
void Launch<T>::bounce(const DScene<T>& S, const DSmallScene& small, const DBatch& B, DQueue<T> qin, DQueue<T> qout, T* sample_rgb,
                       uint32_t depth, uint32_t max_bounces, const uint32_t* tile_offsets, uint32_t* tile_counts_out,
                       uint32_t n_tiles, int stack, int grid_blocks, const DTail* tail, uint32_t n_first, hipStream_t st) {
    const DTail tl = tail ? *tail : DTail{nullptr, nullptr};
    // n_first > 0: the first bounce of a dense batch of n_first paths, camera rays made in the kernel (sweep scenes, no tail)
#define RT_BOUNCE_(STACK, MODE, CAN_TAIL, FIRST)                                                                                 \
    hipLaunchKernelGGL((k_bounce<T, STACK, MODE, CAN_TAIL, FIRST>), dim3(grid_blocks), dim3(BLOCK), 0, st, S, small, B, qin, qout, sample_rgb, \
                       depth, max_bounces, tile_offsets, tile_counts_out, n_tiles, tl, n_first)
#define RT_BOUNCE(STACK, MODE) RT_BOUNCE_(STACK, MODE, true, false)
#define RT_BOUNCE_SWEEP(STACK, MODE)                            \
    do {                                                        \
        if (tail) RT_BOUNCE_(STACK, MODE, true, false);         \
        else if (n_first) RT_BOUNCE_(STACK, MODE, false, true); \
        else RT_BOUNCE_(STACK, MODE, false, false);             \
    } while (0)
    if (S.n_sweep > 0 && S.lean == 1) RT_BOUNCE_SWEEP(1, WALK_SWEEP_LEAN);
    else if (S.n_sweep > 0 && S.lean == 2) RT_BOUNCE_SWEEP(1, WALK_SWEEP_FLAT);
    else if (S.n_sweep > 0 && stack <= 16) RT_BOUNCE_SWEEP(16, WALK_SWEEP);
    else if (S.n_sweep > 0) RT_BOUNCE_SWEEP(64, WALK_SWEEP);
    else if (small.blob && stack <= 16) RT_BOUNCE(16, WALK_OWN_BOX_LDS);
#define RT_BOUNCE4(MODE)                       \
    do {                                       \
        if (stack <= 16) RT_BOUNCE(16, MODE);  \
        else if (stack <= 32) RT_BOUNCE(32, MODE); \
        else RT_BOUNCE(64, MODE);              \
    } while (0)
    // world trees without meshes run their bounces fused (capi.cpp): the launches that are not the tail launch take the
    // kernel compiled without the bounce loop (48 registers fewer spilled on blocks_ball)
#define RT_BOUNCE_T(STACK, MODE)                                \
    do {                                                        \
        if (tail) RT_BOUNCE_(STACK, MODE, true, false);         \
        else RT_BOUNCE_(STACK, MODE, false, false);             \
    } while (0)
#define RT_BOUNCE4_T(MODE)                       \
    do {                                         \
        if (stack <= 16) RT_BOUNCE_T(16, MODE);  \
        else if (stack <= 32) RT_BOUNCE_T(32, MODE); \
        else RT_BOUNCE_T(64, MODE);              \
    } while (0)
    else if (S.nodes4 && S.has_meshes && S.has_media) RT_BOUNCE4(WALK_NODE2_MESH4);
    else if (S.nodes4 && S.has_meshes) RT_BOUNCE4(WALK_NODE4_MESH);
    else if (S.nodes4 && S.has_media) RT_BOUNCE4_T(WALK_NODE4_MEDIA);
    else if (S.nodes4) RT_BOUNCE4_T(WALK_NODE4_PLAIN);
#undef RT_BOUNCE4
#undef RT_BOUNCE4_T
#undef RT_BOUNCE_T
    else if (stack <= 16) RT_BOUNCE(16, WALK_NODE2);
    else if (stack <= 32) RT_BOUNCE(32, WALK_NODE2);
    else RT_BOUNCE(64, WALK_NODE2);
#undef RT_BOUNCE_SWEEP
#undef RT_BOUNCE
#undef RT_BOUNCE_
}
template <typename T>
void Launch<T>::shade(const DScene<T>& S, const DBatch& B, DQueue<T> qin, DHits<T> h, DQueue<T> qout, T* sample_rgb,
                      uint32_t depth, uint32_t max_bounces, const uint32_t* tile_offsets, uint32_t* tile_counts_out,
                      uint32_t n_tiles, int grid_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_shade<T>, dim3(grid_blocks), dim3(BLOCK), 0, st, S, B, qin, h, qout, sample_rgb, depth, max_bounces, tile_offsets, tile_counts_out, n_tiles);
}
template <typename T>
void Launch<T>::accumulate(const T* sample_rgb, uint32_t npix, uint32_t n_samples, double* accum, hipStream_t st) {
    hipLaunchKernelGGL(k_accumulate<T>, dim3(blocks_for((uint64_t)npix * 3)), dim3(BLOCK), 0, st, sample_rgb, npix, n_samples, accum);
}
template <typename T>
void Launch<T>::load_rays(const double* rays9, uint32_t n, DQueue<T> q, T* tmin, T* tmax, uint32_t* tile_counts,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_load_rays<T>, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, st, rays9, n, q, tmin, tmax, tile_counts);
}
template <typename T>
void Launch<T>::second(const DScene<T>& S, const DBatch& B, DQueue<T> q, DHits<T> h, const T* tmin, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_second<T>, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, st, S, B, q, h, tmin, n);
}
template <typename T>
void Launch<T>::store_hits(DQueue<T> q, DHits<T> h, uint32_t n, void* out, hipStream_t st) {
    hipLaunchKernelGGL(k_store_hits<T>, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, st, q, h, n, (rt_hit*)out);
}
template <typename T>
void Launch<T>::store_rays(DQueue<T> q, uint32_t n, double* rays9, hipStream_t st) {
    hipLaunchKernelGGL(k_store_rays<T>, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, st, q, n, rays9);
}
#if RT_PART_HAS(2)
void launch_scan_tiles(const uint32_t* tile_counts, uint32_t n_tiles, uint32_t* tile_offsets, uint32_t* counts_to_zero,
                       uint32_t* total_out, unsigned long long* ray_total, uint32_t* host_total, hipStream_t st, uint32_t dense_total) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_THREADS), 0, st, tile_counts, n_tiles, tile_offsets, counts_to_zero, total_out, ray_total, host_total, dense_total);
}
void launch_quantize(const double* accum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb8, hipStream_t st) {
    uint64_t n = n_pixels * 3;
    hipLaunchKernelGGL(k_quantize, dim3(blocks_for(n ? n : 1)), dim3(BLOCK), 0, st, accum, n, spp_total, rgb8);
}
#endif

// Explicit instantiations, one group per translation unit (RT_PART, see the top of the file).
#define RT_INST_EXTEND(T)                                                                                                        \
    template void Launch<T>::extend(const DScene<T>&, const DSmallScene&, const DBatch&, DQueue<T>, DHits<T>, const T*, const T*, \
                                    uint32_t, const uint32_t*, uint32_t, int, int, uint32_t*, int, hipStream_t);
#define RT_INST_BOUNCE(T)                                                                                                     \
    template void Launch<T>::bounce(const DScene<T>&, const DSmallScene&, const DBatch&, DQueue<T>, DQueue<T>, T*, uint32_t, \
                                    uint32_t, const uint32_t*, uint32_t*, uint32_t, int, int, const DTail*, uint32_t, hipStream_t);
#define RT_INST_REST(T)                                                                                                          \
    template void Launch<T>::raygen(const DScene<T>&, const DBatch&, DQueue<T>, uint32_t, uint32_t*, hipStream_t);                \
    template void Launch<T>::shade(const DScene<T>&, const DBatch&, DQueue<T>, DHits<T>, DQueue<T>, T*, uint32_t, uint32_t,       \
                                   const uint32_t*, uint32_t*, uint32_t, int, hipStream_t);                                      \
    template void Launch<T>::accumulate(const T*, uint32_t, uint32_t, double*, hipStream_t);                                     \
    template void Launch<T>::load_rays(const double*, uint32_t, DQueue<T>, T*, T*, uint32_t*, hipStream_t);                      \
    template void Launch<T>::second(const DScene<T>&, const DBatch&, DQueue<T>, DHits<T>, const T*, uint32_t, hipStream_t);       \
    template void Launch<T>::store_hits(DQueue<T>, DHits<T>, uint32_t, void*, hipStream_t);                                      \
    template void Launch<T>::store_rays(DQueue<T>, uint32_t, double*, hipStream_t);
#if RT_PART_HAS(0)
RT_INST_EXTEND(double)
#endif
#if RT_PART_HAS(1)
RT_INST_BOUNCE(double)
#endif
#if RT_PART_HAS(2)
RT_INST_REST(double)
#endif
#if RT_PART_HAS(3)
RT_INST_EXTEND(float)
#endif
#if RT_PART_HAS(4)
RT_INST_BOUNCE(float)
#endif
#if RT_PART_HAS(5)
RT_INST_REST(float)
#endif

}  // namespace rtd
