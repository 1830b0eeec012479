// Traversal layouts built from the scene description when it is uploaded (host code, no arithmetic
// type): the padded f32 boxes, the reference-shaped DNode2 tree with the depth-first leaf ranks of
// the tie rule, its four-wide fold, and the four-wide trees regrouped by SAH.  See DESIGN.md section 3.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <future>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "device/rt_device.h"

namespace rt_layout {

// Padding of the f32 traversal boxes.  The f32 slab test computes (plane - origin) * (1/d) with a
// relative error of a few 2^-24; a pad of 8e-6 x (largest coordinate magnitude in the scene,
// camera and transform offsets included) exceeds that error by more than an order of magnitude
// for every ray that starts inside the scene's extent, so the test never rejects a box the exact
// f64 test of the reference accepts.
inline double scene_scale(const rt_scene_desc& d) {  // the largest coordinate a ray that starts on a bounded object can have
    double scale = 1e-30;
    auto take = [&](double x) {
        if (std::isfinite(x)) scale = std::max(scale, std::fabs(x));
    };
    for (uint64_t i = 0; i < d.n_nodes; i++)
        for (int k = 0; k < 3; k++) {
            take(d.nodes[i].lo[k]);
            take(d.nodes[i].hi[k]);
        }
    for (uint64_t i = 0; i < d.n_xforms; i++)
        for (int k = 0; k < 3; k++) take(d.xforms[i].b[k]);
    for (int k = 0; k < 3; k++) take(d.camera.origin[k]);
    return scale;
}
inline float box_pad(const rt_scene_desc& d) { return (float)(8e-6 * scene_scale(d)); }
// Origins beyond this (paths scattered off an unbounded Plane) are outside what the padding covers (8e-6 / 2^-24 = 134
// scene sizes): the kernels move them along the ray for the box tests (kernels.hip, far_shift).
inline float far_origin_limit(const rt_scene_desc& d) { return (float)(32.0 * scene_scale(d)); }

// One child slot of a DNode2 from reference node `idx`; returns the child reference.
// Depth-first (left child first) rank of every leaf payload: the reference's BVHNode::hit lets the
// later leaf win an exact tie (src/hitting.rs:211-229); the kernels walk the nearer child first and
// use these ranks to give the same answer.
struct LeafRanks {
    std::vector<int32_t> object, triangle;
    int32_t next = 0;
};

inline int32_t node2_child(const rt_scene_desc& d, int32_t idx, float pad, std::vector<rtd::DNode2>& out, float* lo,
                           float* hi, LeafRanks& ranks) {
    const rt_node& nd = d.nodes[idx];
    bool well_formed = true, flat = false;
    for (int k = 0; k < 3; k++) {
        if (!(nd.lo[k] <= nd.hi[k]) || !std::isfinite(nd.lo[k]) || !std::isfinite(nd.hi[k])) well_formed = false;
        if (nd.lo[k] == nd.hi[k]) flat = true;
    }
    for (int k = 0; k < 3; k++) {
        if (well_formed) {
            lo[k] = std::nextafterf((float)nd.lo[k] - pad, -INFINITY);
            hi[k] = std::nextafterf((float)nd.hi[k] + pad, INFINITY);
            if ((double)lo[k] > nd.lo[k]) lo[k] = std::nextafterf(lo[k], -INFINITY);
            if ((double)hi[k] < nd.hi[k]) hi[k] = std::nextafterf(hi[k], INFINITY);
        } else {  // malformed bounds (e.g. a Block given with swapped corners): never cull
            lo[k] = -INFINITY;
            hi[k] = INFINITY;
        }
    }
    if (nd.left < 0) {  // leaf: payload reference, box = its (padded) bounds
        auto& table = nd.right == RT_LEAF_TRIANGLE ? ranks.triangle : ranks.object;
        table[(size_t)(-1 - nd.left)] = ranks.next++;
        return nd.left;
    }
    // Interior node.  AABB::intersects rejects when max_dist <= min_dist (src/hitting.rs:262): a box
    // of zero thickness on an axis yields t0 == t1 there and is culled for every ray, so the whole
    // subtree is invisible in the reference - dropped here for the same result.
    if (well_formed && flat) return rtd::RT_EMPTY;
    int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::DNode2 n;
    std::memset(&n, 0, sizeof(n));
    n.child[0] = node2_child(d, nd.left, pad, out, n.lo[0], n.hi[0], ranks);
    n.child[1] = node2_child(d, nd.right, pad, out, n.lo[1], n.hi[1], ranks);
    out[(size_t)me] = n;
    return me;
}

inline int tree_depth(const rt_scene_desc& d, int32_t root) {  // interior levels above the deepest leaf
    if (root < 0) return 0;
    int deepest = 0;
    std::vector<std::pair<int32_t, int>> todo{{root, 0}};
    while (!todo.empty()) {
        auto [idx, depth] = todo.back();
        todo.pop_back();
        const rt_node& nd = d.nodes[idx];
        if (nd.left < 0) {
            deepest = std::max(deepest, depth);
            continue;
        }
        todo.push_back({nd.left, depth + 1});
        todo.push_back({nd.right, depth + 1});
    }
    return deepest;
}

// ---- four-wide mesh trees
inline void padded_box(const rt_node& nd, float pad, float* lo, float* hi) {
    bool well_formed = true;
    for (int k = 0; k < 3; k++)
        if (!(nd.lo[k] <= nd.hi[k]) || !std::isfinite(nd.lo[k]) || !std::isfinite(nd.hi[k])) well_formed = false;
    for (int k = 0; k < 3; k++) {
        if (well_formed) {
            lo[k] = std::nextafterf((float)nd.lo[k] - pad, -INFINITY);
            hi[k] = std::nextafterf((float)nd.hi[k] + pad, INFINITY);
            if ((double)lo[k] > nd.lo[k]) lo[k] = std::nextafterf(lo[k], -INFINITY);
            if ((double)hi[k] < nd.hi[k]) hi[k] = std::nextafterf(hi[k], INFINITY);
        } else {
            lo[k] = -INFINITY;
            hi[k] = INFINITY;
        }
    }
}
inline bool dead_interior(const rt_node& nd) {  // zero-thickness interior box: culled for every ray
    if (nd.left < 0) return false;
    bool well_formed = true, flat = false;
    for (int k = 0; k < 3; k++) {
        if (!(nd.lo[k] <= nd.hi[k]) || !std::isfinite(nd.lo[k]) || !std::isfinite(nd.hi[k])) well_formed = false;
        if (nd.lo[k] == nd.hi[k]) flat = true;
    }
    return well_formed && flat;
}
inline double half_area(const rt_node& nd) {
    double e[3];
    for (int k = 0; k < 3; k++) e[k] = std::max(0.0, nd.hi[k] - nd.lo[k]);
    double a = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    return std::isfinite(a) ? a : 1e300;
}
// Child reference for reference node `idx` in the four-wide tree; *need = stack entries a walk below it
// can hold at once (3 pushed siblings per level in the worst visiting order).
inline int32_t node4_child(const rt_scene_desc& d, int32_t idx, float pad, std::vector<rtd::WideNode4>& out, int* need) {
    const rt_node& nd = d.nodes[idx];
    *need = 0;
    if (nd.left < 0) return nd.left;
    if (dead_interior(nd)) return rtd::RT_EMPTY;
    // fold: start from the two children, open the largest interior slot until there are four (order kept)
    std::vector<int32_t> slots{nd.left, nd.right};
    while (slots.size() < 4) {
        int pick = -1;
        double best = -1.0;
        for (size_t k = 0; k < slots.size(); k++) {
            const rt_node& c = d.nodes[slots[k]];
            if (c.left < 0 || dead_interior(c)) continue;
            double a = half_area(c);
            if (a > best) {
                best = a;
                pick = (int)k;
            }
        }
        if (pick < 0) break;
        const rt_node& c = d.nodes[slots[(size_t)pick]];
        const int32_t l = c.left, r = c.right;
        slots[(size_t)pick] = l;
        slots.insert(slots.begin() + pick + 1, r);
    }
    const int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::WideNode4 n;
    std::memset(&n, 0, sizeof(n));
    int deepest = 0, live = 0;
    for (int k = 0; k < 4; k++) {
        n.child[k] = rtd::RT_EMPTY;
        if ((size_t)k >= slots.size()) continue;
        float blo[3], bhi[3];
        padded_box(d.nodes[slots[(size_t)k]], pad, blo, bhi);
        n.set_box(k, blo, bhi);
        int below = 0;
        n.child[k] = node4_child(d, slots[(size_t)k], pad, out, &below);
        if (n.child[k] != rtd::RT_EMPTY) live++;
        deepest = std::max(deepest, below);
    }
    out[(size_t)me] = n;
    *need = std::max(0, live - 1) + deepest;
    return me;
}
inline int32_t build_nodes4(const rt_scene_desc& d, int32_t root, float pad, std::vector<rtd::WideNode4>& out, int* need) {
    const int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::WideNode4 n;
    std::memset(&n, 0, sizeof(n));
    for (int k = 0; k < 4; k++) n.child[k] = rtd::RT_EMPTY;
    float blo[3], bhi[3];
    padded_box(d.nodes[root], pad, blo, bhi);
    n.set_box(0, blo, bhi);
    n.child[0] = node4_child(d, root, pad, out, need);
    out[(size_t)me] = n;
    return me;
}

// ---- four-wide trees rebuilt with the surface-area heuristic.
// The reference's tree (object-median split, src/hitting.rs:147-197) only decides which leaves a
// ray gets to test; the answer - nearest leaf, exact ties to the later one in the reference's
// depth-first order - is kept by the walk's tie rule (dfs_rank), whatever the tree.  So the same
// leaves (minus those under a zero-thickness interior box, which the reference never reaches)
// are regrouped top-down by binned SAH, four children per node.
struct SahLeaf {
    int32_t ref;
    float lo[3], hi[3], c[3];
};
struct SahPart {
    size_t lo, hi;
    float blo[3], bhi[3];
    double area;
};
inline bool collect_leaves(const rt_scene_desc& d, int32_t root, float pad, std::vector<SahLeaf>& out) {
    std::vector<int32_t> todo{root};
    while (!todo.empty()) {
        const rt_node& nd = d.nodes[todo.back()];
        todo.pop_back();
        if (nd.left >= 0) {
            if (dead_interior(nd)) continue;
            todo.push_back(nd.right);
            todo.push_back(nd.left);
            continue;
        }
        SahLeaf l;
        l.ref = nd.left;
        padded_box(nd, pad, l.lo, l.hi);
        for (int k = 0; k < 3; k++) {
            if (!std::isfinite(l.lo[k]) || !std::isfinite(l.hi[k])) return false;  // malformed bounds: keep the reference's tree
            l.c[k] = 0.5f * l.lo[k] + 0.5f * l.hi[k];
        }
        out.push_back(l);
    }
    return true;
}
// The same leaves as flat arrays (what the GPU builder takes: bvh_build.hip): padded boxes [n][3] and child references.
inline bool collect_leaf_arrays(const rt_scene_desc& d, int32_t root, float pad, std::vector<float>& lo, std::vector<float>& hi,
                                std::vector<int32_t>& refs) {
    lo.clear();
    hi.clear();
    refs.clear();
    std::vector<int32_t> todo{root};
    while (!todo.empty()) {
        const rt_node& nd = d.nodes[todo.back()];
        todo.pop_back();
        if (nd.left >= 0) {
            if (dead_interior(nd)) continue;
            todo.push_back(nd.right);
            todo.push_back(nd.left);
            continue;
        }
        float blo[3], bhi[3];
        padded_box(nd, pad, blo, bhi);
        for (int k = 0; k < 3; k++) {
            if (!std::isfinite(blo[k]) || !std::isfinite(bhi[k])) return false;
            lo.push_back(blo[k]);
            hi.push_back(bhi[k]);
        }
        refs.push_back(nd.left);
    }
    return true;
}
inline double box_area(const float* lo, const float* hi) {
    const double e0 = (double)hi[0] - lo[0], e1 = (double)hi[1] - lo[1], e2 = (double)hi[2] - lo[2];
    return e0 * e1 + e1 * e2 + e2 * e0;
}
inline SahPart make_part(const std::vector<SahLeaf>& L, size_t lo, size_t hi) {
    SahPart p;
    p.lo = lo;
    p.hi = hi;
    for (int k = 0; k < 3; k++) {
        p.blo[k] = INFINITY;
        p.bhi[k] = -INFINITY;
    }
    for (size_t i = lo; i < hi; i++)
        for (int k = 0; k < 3; k++) {
            p.blo[k] = std::min(p.blo[k], L[i].lo[k]);
            p.bhi[k] = std::max(p.bhi[k], L[i].hi[k]);
        }
    p.area = box_area(p.blo, p.bhi);
    return p;
}
// Splits L[lo, hi) (>= 2 leaves) in place; returns the first index of the right part.
inline size_t sah_split(std::vector<SahLeaf>& L, size_t lo, size_t hi) {
    constexpr int BINS = 32;
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = lo; i < hi; i++)
        for (int k = 0; k < 3; k++) {
            cmin[k] = std::min(cmin[k], L[i].c[k]);
            cmax[k] = std::max(cmax[k], L[i].c[k]);
        }
    struct AxisBest {
        double cost = INFINITY;
        int bin = 0;
        float scale = 0.0f;
    };
    auto try_axis = [&L, lo, hi, &cmin, &cmax](int axis) {
        AxisBest best;
        const float extent = cmax[axis] - cmin[axis];
        if (!(extent > 0.0f)) return best;
        const float scale = (float)BINS / extent;
        struct Bin {
            size_t n = 0;
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        } bins[BINS];
        for (size_t i = lo; i < hi; i++) {
            const int b = std::min(BINS - 1, (int)((L[i].c[axis] - cmin[axis]) * scale));
            bins[b].n++;
            for (int k = 0; k < 3; k++) {
                bins[b].lo[k] = std::min(bins[b].lo[k], L[i].lo[k]);
                bins[b].hi[k] = std::max(bins[b].hi[k], L[i].hi[k]);
            }
        }
        double right_area[BINS];
        size_t right_n[BINS];
        Bin acc;
        for (int b = BINS - 1; b > 0; b--) {
            acc.n += bins[b].n;
            for (int k = 0; k < 3; k++) {
                acc.lo[k] = std::min(acc.lo[k], bins[b].lo[k]);
                acc.hi[k] = std::max(acc.hi[k], bins[b].hi[k]);
            }
            right_n[b] = acc.n;
            right_area[b] = acc.n ? box_area(acc.lo, acc.hi) : 0.0;
        }
        acc = Bin();
        for (int b = 0; b + 1 < BINS; b++) {  // left = bins [0, b], right = bins (b, BINS)
            acc.n += bins[b].n;
            for (int k = 0; k < 3; k++) {
                acc.lo[k] = std::min(acc.lo[k], bins[b].lo[k]);
                acc.hi[k] = std::max(acc.hi[k], bins[b].hi[k]);
            }
            if (acc.n == 0 || right_n[b + 1] == 0) continue;
            const double cost = box_area(acc.lo, acc.hi) * (double)acc.n + right_area[b + 1] * (double)right_n[b + 1];
            if (cost < best.cost) {
                best.cost = cost;
                best.bin = b;
                best.scale = scale;
            }
        }
        return best;
    };
    AxisBest per_axis[3];
    if (hi - lo >= 100000) {  // the few splits near the root of a large mesh: one task per axis
        std::future<AxisBest> f1 = std::async(std::launch::async, try_axis, 1), f2 = std::async(std::launch::async, try_axis, 2);
        per_axis[0] = try_axis(0);
        per_axis[1] = f1.get();
        per_axis[2] = f2.get();
    } else {
        for (int axis = 0; axis < 3; axis++) per_axis[axis] = try_axis(axis);
    }
    int best_axis = -1, best_bin = 0;
    double best_cost = INFINITY;
    float best_scale = 0.0f;
    for (int axis = 0; axis < 3; axis++)  // (strict <: the first axis wins a tie, as in the serial loop)
        if (per_axis[axis].cost < best_cost) {
            best_cost = per_axis[axis].cost;
            best_axis = axis;
            best_bin = per_axis[axis].bin;
            best_scale = per_axis[axis].scale;
        }
    // One more candidate: the largest leaf on its own (a box that spans the others - a medium that
    // fills the scene - has a central centroid, so no centroid split ever isolates it).
    {
        size_t big = lo;
        double big_area = -1.0;
        for (size_t i = lo; i < hi; i++) {
            const double a = box_area(L[i].lo, L[i].hi);
            if (a > big_area) {
                big_area = a;
                big = i;
            }
        }
        std::swap(L[lo], L[big]);
        const SahPart rest = make_part(L, lo + 1, hi);
        if (big_area + rest.area * (double)(hi - lo - 1) < best_cost) return lo + 1;
    }
    if (best_axis < 0) return lo + (hi - lo) / 2;  // all centroids coincide
    const float origin = cmin[best_axis];
    auto mid = std::partition(L.begin() + (long)lo, L.begin() + (long)hi, [&](const SahLeaf& l) {
        return std::min(BINS - 1, (int)((l.c[best_axis] - origin) * best_scale)) <= best_bin;
    });
    const size_t m = (size_t)(mid - L.begin());
    return (m == lo || m == hi) ? lo + (hi - lo) / 2 : m;
}
// par_depth > 0: the subtrees of large children are built by tasks of their own (std::async), each into a vector of
// its own that is appended afterwards with its node indices moved; a child's whole subtree still follows the child
// before it, so `out` is the same pre-order array as the serial build's, index for index (1 M triangles: 850 -> ~200 ms).
inline int32_t sah4_node(std::vector<SahLeaf>& L, size_t lo, size_t hi, std::vector<rtd::WideNode4>& out, int* need, int par_depth = 0) {
    std::vector<SahPart> parts{make_part(L, lo, hi)};
    while (parts.size() < 4) {
        int pick = -1;
        for (size_t k = 0; k < parts.size(); k++)
            if (parts[k].hi - parts[k].lo > 1 && (pick < 0 || parts[k].area > parts[(size_t)pick].area)) pick = (int)k;
        if (pick < 0) break;
        const SahPart p = parts[(size_t)pick];
        const size_t mid = sah_split(L, p.lo, p.hi);
        parts[(size_t)pick] = make_part(L, p.lo, mid);
        parts.insert(parts.begin() + pick + 1, make_part(L, mid, p.hi));
    }
    // Slot order only matters to rays that enter several children at the same distance - those that
    // start inside them.  Largest box first: it is the one most of the other lanes of a wave are in.
    std::stable_sort(parts.begin(), parts.end(), [](const SahPart& a, const SahPart& b) { return a.area > b.area; });
    const int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::WideNode4 n;
    std::memset(&n, 0, sizeof(n));
    int deepest = 0;
    struct Task {
        std::vector<rtd::WideNode4> nodes;
        int below = 0;
        std::future<int32_t> root;
    };
    Task tasks[4];
    for (int k = 0; k < 4 && par_depth > 0; k++) {
        if ((size_t)k >= parts.size() || parts[(size_t)k].hi - parts[(size_t)k].lo < 4096) continue;
        const SahPart p = parts[(size_t)k];
        Task* t = &tasks[k];
        t->root = std::async(std::launch::async, [&L, p, t, par_depth] { return sah4_node(L, p.lo, p.hi, t->nodes, &t->below, par_depth - 1); });
    }
    for (int k = 0; k < 4; k++) {
        n.child[k] = rtd::RT_EMPTY;
        if ((size_t)k >= parts.size()) continue;
        const SahPart& p = parts[(size_t)k];
        n.set_box(k, p.blo, p.bhi);
        int below = 0;
        if (tasks[k].root.valid()) {
            const int32_t local_root = tasks[k].root.get();
            const int32_t offset = (int32_t)out.size();
            for (rtd::WideNode4& w : tasks[k].nodes)
                for (int c = 0; c < 4; c++)
                    if (w.child[c] >= 0) w.child[c] += offset;
            out.insert(out.end(), tasks[k].nodes.begin(), tasks[k].nodes.end());
            std::vector<rtd::WideNode4>().swap(tasks[k].nodes);
            n.child[k] = offset + local_root;
            below = tasks[k].below;
        } else {
            n.child[k] = p.hi - p.lo == 1 ? L[p.lo].ref : sah4_node(L, p.lo, p.hi, out, &below);
        }
        deepest = std::max(deepest, below);
    }
    out[(size_t)me] = n;
    *need = (int)parts.size() - 1 + deepest;
    return me;
}
// Super-root like build_nodes4's.  false: the tree has leaves without finite bounds, nothing was added.
inline bool build_sah4(const rt_scene_desc& d, int32_t root, float pad, std::vector<rtd::WideNode4>& out, int32_t* root_out, int* need) {
    std::vector<SahLeaf> L;
    if (!collect_leaves(d, root, pad, L)) return false;
    const int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::WideNode4 n;
    std::memset(&n, 0, sizeof(n));
    for (int k = 0; k < 4; k++) n.child[k] = rtd::RT_EMPTY;
    *need = 0;
    if (!L.empty()) {
        const SahPart all = make_part(L, 0, L.size());
        n.set_box(0, all.blo, all.bhi);
        n.child[0] = L.size() == 1 ? L[0].ref : sah4_node(L, 0, L.size(), out, need, 2);
    }
    out[(size_t)me] = n;
    *root_out = me;
    return true;
}

// DNode2 super-root for the reference BVH rooted at `root`: child 0 = that root, child 1 empty.
inline int32_t build_nodes2(const rt_scene_desc& d, int32_t root, float pad, std::vector<rtd::DNode2>& out, LeafRanks& ranks) {
    int32_t me = (int32_t)out.size();
    out.emplace_back();
    rtd::DNode2 n;
    std::memset(&n, 0, sizeof(n));
    n.child[0] = node2_child(d, root, pad, out, n.lo[0], n.hi[0], ranks);
    n.child[1] = rtd::RT_EMPTY;
    out[(size_t)me] = n;
    return me;
}

// WideNode4 -> DNode4: every child box as 8-bit steps from the node's corner.  The kernels evaluate
// plane distances as fma(q, step * 1/d, (origin - o) * 1/d) in f32; its error (a few 2^-24 of
// |origin - o| / |d|) is covered by the padding the boxes already carry (box_pad), as long as the
// planes decoded here - with the same fmaf - lie on or outside the padded planes, which is checked
// plane by plane.  false: a box without finite bounds cannot be stored this way.
inline bool quantize_range(const std::vector<rtd::WideNode4>& wide, std::vector<rtd::DNode4>& out, size_t begin, size_t end);
inline bool quantize_nodes4(const std::vector<rtd::WideNode4>& wide, std::vector<rtd::DNode4>& out) {
    out.resize(wide.size());
    const size_t n_tasks = wide.size() >= 65536 ? 8 : 1;  // nodes are independent: large trees in eight slices
    std::vector<std::future<bool>> parts;
    for (size_t k = 1; k < n_tasks; k++)
        parts.push_back(std::async(std::launch::async, [&wide, &out, k, n_tasks] {
            return quantize_range(wide, out, wide.size() * k / n_tasks, wide.size() * (k + 1) / n_tasks);
        }));
    bool ok = quantize_range(wide, out, 0, wide.size() / n_tasks);
    for (auto& p : parts) ok = p.get() && ok;
    return ok;
}
inline bool quantize_range(const std::vector<rtd::WideNode4>& wide, std::vector<rtd::DNode4>& out, size_t begin, size_t end) {
    for (size_t i = begin; i < end; i++) {
        const rtd::WideNode4& w = wide[i];
        rtd::DNode4 n;
        std::memset(&n, 0, sizeof(n));
        for (int k = 0; k < 4; k++) n.child[k] = w.child[k];
        for (int a = 0; a < 3; a++) {
            float lo_min = INFINITY, hi_max = -INFINITY;
            for (int k = 0; k < 4; k++) {
                if (w.child[k] == rtd::RT_EMPTY) continue;
                if (!std::isfinite(w.lo[a][k]) || !std::isfinite(w.hi[a][k])) return false;
                lo_min = std::min(lo_min, w.lo[a][k]);
                hi_max = std::max(hi_max, w.hi[a][k]);
            }
            if (!(lo_min <= hi_max)) {  // no live child
                lo_min = 0.0f;
                hi_max = 0.0f;
            }
            float step = std::max((hi_max - lo_min) / 255.0f, 1e-30f);
            while (std::fmaf(255.0f, step, lo_min) < hi_max) step = std::nextafterf(step, INFINITY);
            if (!std::isfinite(step)) return false;
            n.origin[a] = lo_min;
            n.step[a] = step;
            for (int k = 0; k < 4; k++) {
                if (w.child[k] == rtd::RT_EMPTY) continue;
                int qlo = (int)std::floor((w.lo[a][k] - lo_min) / step);
                qlo = std::min(255, std::max(0, qlo));
                while (qlo > 0 && std::fmaf((float)qlo, step, lo_min) > w.lo[a][k]) qlo--;
                int qhi = (int)std::ceil((w.hi[a][k] - lo_min) / step);
                qhi = std::min(255, std::max(0, qhi));
                while (qhi < 255 && std::fmaf((float)qhi, step, lo_min) < w.hi[a][k]) qhi++;
                if (std::fmaf((float)qlo, step, lo_min) > w.lo[a][k] || std::fmaf((float)qhi, step, lo_min) < w.hi[a][k]) return false;
                n.qlo[a][k] = (uint8_t)qlo;
                n.qhi[a][k] = (uint8_t)qhi;
            }
        }
        out[i] = n;
    }
    return true;
}

}  // namespace rt_layout
