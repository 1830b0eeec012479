// Flattener: see flatten.hpp.  Every rule that shapes the result cites the reference.
#include "flatten.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <map>

namespace rt {

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double PI = 3.14159265358979323846264338327950288;

struct V3 {
    double x, y, z;
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    double& operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
};
V3 v3(const double* p) { return V3{p[0], p[1], p[2]}; }
V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
V3 operator*(double s, V3 a) { return a * s; }
V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V3 cross(V3 l, V3 r) { return V3{l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x}; }
double length(V3 a) { return std::sqrt(dot(a, a)); }
V3 unit(V3 a) { return V3{a.x / length(a), a.y / length(a), a.z / length(a)}; }  // src/math.rs:34-40
void store(double* out, V3 a) { out[0] = a.x; out[1] = a.y; out[2] = a.z; }

struct Box {
    V3 lo, hi;
};
Box surrounding(const Box& a, const Box& b) {  // src/hitting.rs:270-282
    return Box{V3{std::fmin(a.lo.x, b.lo.x), std::fmin(a.lo.y, b.lo.y), std::fmin(a.lo.z, b.lo.z)},
               V3{std::fmax(a.hi.x, b.hi.x), std::fmax(a.hi.y, b.hi.y), std::fmax(a.hi.z, b.hi.z)}};
}

// Bounding box of Rotate{X,Y,Z}::by_radians: 27 corner combinations, index 2 extrapolating
// (src/transforms.rs:70-88, 161-179, 252-270).  (p, q) as in transforms.rs: X -> (z, y),
// Y -> (x, z), Z -> (y, x); new_p = c*p + s*q, new_q = -s*p + c*q.
Box rotate_box(const Box& b, int axis, double radians) {
    double s = std::sin(radians), c = std::cos(radians);
    int p = axis == 0 ? 2 : axis == 1 ? 0 : 1;
    int q = axis == 0 ? 1 : axis == 1 ? 2 : 0;
    V3 mn{INF, INF, INF}, mx{-INF, -INF, -INF};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) {
                V3 corner{(double)i * b.hi.x + (1.0 - (double)i) * b.lo.x,
                          (double)j * b.hi.y + (1.0 - (double)j) * b.lo.y,
                          (double)k * b.hi.z + (1.0 - (double)k) * b.lo.z};
                V3 t = corner;
                t[p] = c * corner[p] + s * corner[q];
                t[q] = -s * corner[p] + c * corner[q];
                for (int a = 0; a < 3; a++) {
                    mn[a] = std::fmin(mn[a], t[a]);
                    mx[a] = std::fmax(mx[a], t[a]);
                }
            }
    return Box{mn, mx};
}

// world -> object rigid transform, composed down a chain of transform objects.
struct Xf {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double b[3] = {0, 0, 0};
    bool identity = true;
};
Xf then(const Xf& outer, const double mi[9], const double bi[3]) {
    // p2 = mi * (outer.m p + outer.b) + bi
    Xf r;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            r.m[i * 3 + j] = mi[i * 3 + 0] * outer.m[0 * 3 + j] + mi[i * 3 + 1] * outer.m[1 * 3 + j] +
                             mi[i * 3 + 2] * outer.m[2 * 3 + j];
        r.b[i] = mi[i * 3 + 0] * outer.b[0] + mi[i * 3 + 1] * outer.b[1] + mi[i * 3 + 2] * outer.b[2] + bi[i];
    }
    r.identity = false;
    return r;
}
Xf then_translate(const Xf& outer, V3 offset) {  // Translate::hit: origin - offset (src/transforms.rs:23)
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const double b[3] = {-offset.x, -offset.y, -offset.z};
    return then(outer, I, b);
}
Xf then_rotate(const Xf& outer, int axis, double radians) {
    double s = std::sin(radians), c = std::cos(radians);
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (axis == 0) {         // RotateX (src/transforms.rs:104-105): z' = c z - s y ; y' = s z + c y
        m[4] = c; m[5] = s; m[7] = -s; m[8] = c;
    } else if (axis == 1) {  // RotateY (:195-196): x' = c x - s z ; z' = s x + c z
        m[0] = c; m[2] = -s; m[6] = s; m[8] = c;
    } else {                 // RotateZ (:286-287): y' = c y - s x ; x' = s y + c x
        m[0] = c; m[1] = s; m[3] = -s; m[4] = c;
    }
    const double z[3] = {0, 0, 0};
    return then(outer, m, z);
}

double signum(double x) {  // f64::signum
    if (x != x) return x;
    return std::signbit(x) ? -1.0 : 1.0;
}
void spotlight_angles(const ObjectDef& o, double* theta1, double* theta2) {  // src/objects.rs:677-696
    V3 direction = v3(o.p1) - v3(o.p0);
    V3 dx{direction.x, 0, 0}, dz{0, 0, direction.z};
    *theta1 = length(direction) != 0.0 ? -std::acos(dot(V3{0, 1, 0}, unit(direction))) : 0.0;
    *theta2 = (dx.x != 0.0 || dz.z != 0.0) ? std::acos(dot(V3{1, 0, 0}, unit(dx + dz))) * -signum(dz.z) : 0.0;
}

uint32_t hash_combine(uint32_t a, uint32_t b) { return a ^ (b + 0x9E3779B9u + (a << 6) + (a >> 2)); }

struct BuildItem {
    int32_t payload;
    Box sort_box;  // bounding_box(0.0, 0.0), the sort key of bbox_compare (src/hitting.rs:199-209)
    Box node_box;  // bounding_box(time0, time1), what node boxes are made of
};

class Flattener {
public:
    explicit Flattener(const Scene& sc) : sc_(sc) {}

    FlatScene run() {
        materials();
        images();
        camera();
        sky();
        prim_first_.assign(sc_.objects.size(), -1);
        blas_root_.assign(sc_.objects.size(), -1);
        blas_depth_.assign(sc_.objects.size(), 0);
        // World: BVHNode::from_vec(world, start_time, end_time), src/config.rs:36 + src/hitting.rs:148-197.
        double t0 = sc_.camera.start_time, t1 = sc_.camera.end_time;
        std::vector<BuildItem> boxed;
        std::vector<int32_t> unboxed;
        int nested_depth = 0;
        for (size_t k = 0; k < sc_.world.size(); k++) {
            int oi = sc_.world[k];
            int depth = 0;
            int32_t obj = instantiate(oi, Xf(), 0u, (int32_t)k, &depth);
            nested_depth = std::max(nested_depth, depth);
            Box nb, sb;
            if (bbox(oi, t0, t1, &nb) && bbox(oi, 0.0, 0.0, &sb)) boxed.push_back(BuildItem{obj, sb, nb});
            else unboxed.push_back(obj);
        }
        int tlas_depth = 0;
        if (boxed.size() >= 2) {
            out_.tlas_root = build_bvh(boxed, 0, boxed.size(), RT_LEAF_OBJECT, &tlas_depth, nullptr);
            out_.top_list.push_back(RT_TOP_BVH);  // vec![bvh_result] then the unboxed (:191-192)
            for (int32_t u : unboxed) out_.top_list.push_back(u);
        } else {
            out_.top_list = unboxed;  // unboxed first, then the single boxed object (:167)
            if (boxed.size() == 1) out_.top_list.push_back(boxed[0].payload);
        }
        out_.stack_depth = tlas_depth + nested_depth + 2;
        return std::move(out_);
    }

private:
    // ---- materials / textures (src/config.rs:65-174)
    void materials() {
        for (auto& m : sc_.materials) {
            rt_material r;
            std::memset(&r, 0, sizeof(r));
            r.image = -1;
            r.odd = r.even = -1;
            switch (m.type) {
                case MaterialDef::Lambertian: r.type = RT_MAT_LAMBERTIAN; texture(m.texture, &r); break;
                case MaterialDef::Metal:
                    r.type = RT_MAT_METAL;
                    std::memcpy(r.colour, m.albedo, sizeof(r.colour));
                    r.param = m.fuzz;
                    break;
                case MaterialDef::Dielectric: r.type = RT_MAT_DIELECTRIC; r.param = m.index_of_refraction; break;
                case MaterialDef::DiffuseLight: r.type = RT_MAT_DIFFUSE_LIGHT; texture(m.texture, &r); break;
                case MaterialDef::Isotropic: r.type = RT_MAT_ISOTROPIC; texture(m.texture, &r); break;
                case MaterialDef::Checkered:
                    r.type = RT_MAT_CHECKERED;
                    r.odd = m.odd;
                    r.even = m.even;
                    r.param = m.tile_density;
                    break;
            }
            out_.materials.push_back(r);
        }
        // Implicit spotlight materials (src/objects.rs:639-641), two per spotlight in object order.
        spot_material_.assign(sc_.objects.size(), -1);
        for (size_t i = 0; i < sc_.objects.size(); i++) {
            if (sc_.objects[i].type != ObjectDef::Spotlight) continue;
            spot_material_[i] = (int)out_.materials.size();
            rt_material dark, light;
            std::memset(&dark, 0, sizeof(dark));
            dark.type = RT_MAT_LAMBERTIAN;
            dark.image = -1;
            dark.odd = dark.even = -1;
            light = dark;
            light.type = RT_MAT_DIFFUSE_LIGHT;
            std::memcpy(light.colour, sc_.objects[i].p2, sizeof(light.colour));
            out_.materials.push_back(dark);
            out_.materials.push_back(light);
        }
    }
    void texture(int ti, rt_material* r) {
        const TextureDef& t = sc_.textures[(size_t)ti];
        if (t.type == TextureDef::SolidColour) std::memcpy(r->colour, t.colour, sizeof(r->colour));
        else r->image = t.image;
    }
    void images() {
        for (auto& im : sc_.images) {
            rt_image r;
            r.width = im.width;
            r.height = im.height;
            r.texel_offset = out_.texels.size();
            out_.texels.insert(out_.texels.end(), im.rgb.begin(), im.rgb.end());
            out_.images.push_back(r);
        }
    }
    void camera() {  // Camera::new, src/camera.rs:25-66
        const CameraDef& c = sc_.camera;
        double theta = c.vertical_fov * (PI / 180.0);
        double viewport_height = 2.0 * std::tan(theta / 2.0);
        double viewport_width = sc_.aspect() * viewport_height;
        V3 w = unit(v3(c.look_from) - v3(c.look_at));
        V3 u = unit(cross(v3(c.direction_up), w));
        V3 v = cross(w, u);
        V3 origin = v3(c.look_from);
        V3 horizontal = c.focus_dist * viewport_width * u;
        V3 vertical = c.focus_dist * viewport_height * v;
        V3 llc = origin - horizontal / 2.0 - vertical / 2.0 - c.focus_dist * w;
        rt_camera& o = out_.camera;
        store(o.origin, origin);
        store(o.lower_left_corner, llc);
        store(o.horizontal, horizontal);
        store(o.vertical, vertical);
        store(o.u, u);
        store(o.v, v);
        o.lens_radius = c.aperture / 2.0;
        o.start_time = c.start_time;
        o.end_time = c.end_time;
    }
    void sky() {  // src/config.rs:405-421, src/camera.rs:78-86
        rt_sky& s = out_.sky;
        std::memset(&s, 0, sizeof(s));
        if (sc_.background.type == BackgroundDef::PlainColour) {
            std::memcpy(s.colour, sc_.background.colour, sizeof(s.colour));
        } else {
            s.gradient = 1;
            store(s.unit_dir, unit(v3(sc_.background.direction)));
            std::memcpy(s.colour0, sc_.background.colour0, sizeof(s.colour0));
            std::memcpy(s.colour1, sc_.background.colour1, sizeof(s.colour1));
        }
    }

    // ---- Hittable::bounding_box of a scene object, by the reference's rules.
    bool bbox(int oi, double t0, double t1, Box* out) {
        const ObjectDef& o = sc_.objects[(size_t)oi];
        switch (o.type) {
            case ObjectDef::Sphere: {  // src/objects.rs:50-55
                V3 c = v3(o.p0), r{o.radius, o.radius, o.radius};
                *out = Box{c - r, c + r};
                return true;
            }
            case ObjectDef::MovingSphere: {  // :118-128
                auto centre = [&](double t) {
                    return v3(o.p0) + ((t - o.time0) / (o.time1 - o.time0)) * (v3(o.p1) - v3(o.p0));
                };
                V3 r{o.radius, o.radius, o.radius};
                *out = surrounding(Box{centre(t0) - r, centre(t0) + r}, Box{centre(t1) - r, centre(t1) + r});
                return true;
            }
            case ObjectDef::Block:  // :181-186 (corners taken as given)
                *out = Box{v3(o.p0), v3(o.p1)};
                return true;
            case ObjectDef::Rect: {  // :245-250, 309-314, 373-378
                Box b{v3(o.p0), v3(o.p1)};
                b.lo[o.rect_axis] = o.p0[o.rect_axis] - 0.0001;
                b.hi[o.rect_axis] = o.p0[o.rect_axis] + 0.0001;
                *out = b;
                return true;
            }
            case ObjectDef::Triangle: {  // vertices_to_bbox, :512-534
                *out = tri_box(v3(o.p0), v3(o.p1), v3(o.p2));
                return true;
            }
            case ObjectDef::Plane: return false;  // :623-625
            case ObjectDef::Mesh: {  // root box of BVHNode::from_vec(triangles, ..) = union of all
                const MeshData& md = sc_.meshes[(size_t)o.mesh];
                Box all = tri_box(v3(md.triangles[0].p[0]), v3(md.triangles[0].p[1]), v3(md.triangles[0].p[2]));
                for (auto& t : md.triangles) all = surrounding(all, tri_box(v3(t.p[0]), v3(t.p[1]), v3(t.p[2])));
                *out = all;
                return true;
            }
            case ObjectDef::Spotlight: {  // Translate(RotateY(RotateZ(prim))), :665-701
                double th1, th2;
                spotlight_angles(o, &th1, &th2);
                Box prim{V3{-o.width / 2.0, -o.length, -o.width / 2.0}, V3{o.width / 2.0, 0, o.width / 2.0}};
                Box b = rotate_box(rotate_box(prim, 2, th1), 1, th2);
                *out = Box{b.lo + v3(o.p0), b.hi + v3(o.p0)};
                return true;
            }
            case ObjectDef::ConstantMedium: return bbox(o.prototype, t0, t1, out);  // :449-451
            case ObjectDef::Translate: {  // src/transforms.rs:38-47
                Box b;
                if (!bbox(o.prototype, t0, t1, &b)) return false;
                *out = Box{b.lo + v3(o.p0), b.hi + v3(o.p0)};
                return true;
            }
            case ObjectDef::RotateX:
            case ObjectDef::RotateY:
            case ObjectDef::RotateZ: {  // cached at construction over (TIME_MIN, TIME_MAX)
                Box b;
                if (!bbox(o.prototype, 0.0, 1.0, &b)) return false;
                int axis = o.type == ObjectDef::RotateX ? 0 : o.type == ObjectDef::RotateY ? 1 : 2;
                *out = rotate_box(b, axis, o.degrees * (PI / 180.0));
                return true;
            }
        }
        return false;
    }
    static Box tri_box(V3 a, V3 b, V3 c) {
        Box r{V3{INF, INF, INF}, V3{-INF, -INF, -INF}};
        const V3 vs[3] = {a, b, c};
        for (auto& v : vs)
            for (int k = 0; k < 3; k++) {
                r.lo[k] = std::fmin(r.lo[k], v[k]);
                r.hi[k] = std::fmax(r.hi[k], v[k]);
            }
        return r;
    }

    // ---- BVHNode::from_vec (src/hitting.rs:148-197) over boxed items [lo, hi).
    // Axis: the reference draws it at random (:158); here it is the longest axis of the
    // union of the items' boxes, ties to the lower axis (the oracle uses the same rule).
    // A tree over m items has 2m - 1 nodes, laid out in pre-order (a node, its left subtree, its right subtree): the
    // place of every node is known before it is built, so large subtrees are built as tasks into the one array, and
    // the result does not depend on how many run at a time.
    int32_t leaf_base_ = -1;
    int32_t build_bvh(std::vector<BuildItem>& items, size_t lo, size_t hi, int leaf_kind, int* depth, Box* box_out) {
        const int32_t root = (int32_t)out_.nodes.size();
        out_.nodes.resize(out_.nodes.size() + 2 * (hi - lo) - 1);
        build_bvh_at(items, lo, hi, leaf_kind, root, depth, box_out, hi - lo >= 65536 ? 4 : 0);
        return root;
    }
    // Sorts items[lo, hi) by sort_box.lo[axis], equal keys in their present order (the reference's sort_by is stable,
    // src/hitting.rs:166): large ranges through (key, position) pairs, so that the records move once.
    static void sort_items(std::vector<BuildItem>& items, size_t lo, size_t hi, int axis) {
        const size_t m = hi - lo;
        if (m < 2048) {
            std::stable_sort(items.begin() + (long)lo, items.begin() + (long)hi,
                             [axis](const BuildItem& a, const BuildItem& b) { return a.sort_box.lo[axis] < b.sort_box.lo[axis]; });
            return;
        }
        std::vector<std::pair<double, uint32_t>> keys(m);
        for (size_t i = 0; i < m; i++) keys[i] = {items[lo + i].sort_box.lo[axis], (uint32_t)i};
        std::stable_sort(keys.begin(), keys.end(),
                         [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) { return a.first < b.first; });
        std::vector<BuildItem> moved(m);
        for (size_t i = 0; i < m; i++) moved[i] = items[lo + keys[i].second];
        std::copy(moved.begin(), moved.end(), items.begin() + (long)lo);
    }
    void build_bvh_at(std::vector<BuildItem>& items, size_t lo, size_t hi, int leaf_kind, int32_t idx, int* depth, Box* box_out,
                      int par_depth) {
        if (hi - lo == 1) {
            rt_node n;
            std::memset(&n, 0, sizeof(n));
            store(n.lo, items[lo].node_box.lo);  // the leaf's own bounds (never tested by the reference)
            store(n.hi, items[lo].node_box.hi);
            // (leaf_base_ >= 0: the payload is the item's final position - items are in depth-first leaf order
            // by the time a leaf is made, see the Mesh case)
            n.left = -1 - (leaf_base_ >= 0 ? leaf_base_ + (int32_t)lo : items[lo].payload);
            n.right = leaf_kind;
            out_.nodes[(size_t)idx] = n;
            if (box_out) *box_out = items[lo].node_box;
            *depth = 1;
            return;
        }
        Box all = items[lo].node_box;
        for (size_t i = lo; i < hi; i++) all = surrounding(all, items[i].node_box);
        int axis = 0;
        double best = -1.0;
        for (int a = 0; a < 3; a++) {
            double e = all.hi[a] - all.lo[a];
            if (e > best) {
                best = e;
                axis = a;
            }
        }
        sort_items(items, lo, hi, axis);
        size_t halfway = lo + (hi - lo) / 2;
        int dl = 0, dr = 0;
        Box bl, br;
        const int32_t left = idx + 1, right = idx + (int32_t)(2 * (halfway - lo));
        if (par_depth > 0 && hi - lo >= 32768) {
            auto task = std::async(std::launch::async, [&] { build_bvh_at(items, lo, halfway, leaf_kind, left, &dl, &bl, par_depth - 1); });
            build_bvh_at(items, halfway, hi, leaf_kind, right, &dr, &br, par_depth - 1);
            task.get();
        } else {
            build_bvh_at(items, lo, halfway, leaf_kind, left, &dl, &bl, 0);
            build_bvh_at(items, halfway, hi, leaf_kind, right, &dr, &br, 0);
        }
        Box nb = surrounding(bl, br);
        rt_node n;
        store(n.lo, nb.lo);
        store(n.hi, nb.hi);
        n.left = left;
        n.right = right;
        out_.nodes[(size_t)idx] = n;
        if (box_out) *box_out = nb;
        *depth = 1 + std::max(dl, dr);
    }

    int32_t add_xform(const Xf& x) {
        if (x.identity) return -1;
        rt_xform r;
        std::memcpy(r.m, x.m, sizeof(r.m));
        std::memcpy(r.b, x.b, sizeof(r.b));
        out_.xforms.push_back(r);
        return (int32_t)out_.xforms.size() - 1;
    }

    void push_rect(int axis, double a0, double a1, double b0, double b1, double k, int material, bool facing,
                   int owner, int sub) {
        rt_rect r;
        std::memset(&r, 0, sizeof(r));
        r.a0 = a0; r.a1 = a1; r.b0 = b0; r.b1 = b1; r.k = k;
        r.axis = axis;
        r.facing_positive = facing ? 1 : 0;
        r.material = material; r.owner = owner; r.sub = sub;
        out_.rects.push_back(r);
    }

    // Makes one rt_object for scene object `oi` reached through the transform chain `x`.
    // Primitive arrays are filled once per scene object and shared by all its instances.
    // `ctx` hashes the transform objects entered so far (volume RNG keying); `depth`
    // returns the traversal-stack entries needed below the TLAS leaf.
    int32_t instantiate(int oi, const Xf& x, uint32_t ctx, int32_t world_index, int* depth) {
        const ObjectDef& o = sc_.objects[(size_t)oi];
        rt_object r;
        std::memset(&r, 0, sizeof(r));
        r.world_index = world_index;
        r.count = 1;
        *depth = 0;
        switch (o.type) {
            case ObjectDef::Translate:
                return instantiate(o.prototype, then_translate(x, v3(o.p0)), hash_combine(ctx, (uint32_t)oi), world_index, depth);
            case ObjectDef::RotateX:
            case ObjectDef::RotateY:
            case ObjectDef::RotateZ: {
                int axis = o.type == ObjectDef::RotateX ? 0 : o.type == ObjectDef::RotateY ? 1 : 2;
                return instantiate(o.prototype, then_rotate(x, axis, o.degrees * (PI / 180.0)),
                                   hash_combine(ctx, (uint32_t)oi), world_index, depth);
            }
            case ObjectDef::Sphere:
                if (prim_first_[(size_t)oi] < 0) {
                    rt_sphere s;
                    std::memcpy(s.centre, o.p0, sizeof(s.centre));
                    s.radius = o.radius;
                    s.material = o.material;
                    s.owner = oi;
                    prim_first_[(size_t)oi] = (int)out_.spheres.size();
                    out_.spheres.push_back(s);
                }
                r.kind = RT_OBJ_SPHERE;
                r.xform = add_xform(x);
                break;
            case ObjectDef::MovingSphere:
                if (prim_first_[(size_t)oi] < 0) {
                    rt_moving_sphere s;
                    std::memcpy(s.centre0, o.p0, sizeof(s.centre0));
                    std::memcpy(s.centre1, o.p1, sizeof(s.centre1));
                    s.time0 = o.time0; s.time1 = o.time1; s.radius = o.radius;
                    s.material = o.material;
                    s.owner = oi;
                    prim_first_[(size_t)oi] = (int)out_.moving_spheres.size();
                    out_.moving_spheres.push_back(s);
                }
                r.kind = RT_OBJ_MOVING_SPHERE;
                r.xform = add_xform(x);
                break;
            case ObjectDef::Block:  // six sides in the order of Block::new, src/objects.rs:149-168
                if (prim_first_[(size_t)oi] < 0) {
                    prim_first_[(size_t)oi] = (int)out_.rects.size();
                    const double *mn = o.p0, *mx = o.p1;
                    push_rect(2, mn[0], mx[0], mn[1], mx[1], mn[2], o.material, true, oi, 0);
                    push_rect(2, mn[0], mx[0], mn[1], mx[1], mx[2], o.material, false, oi, 1);
                    push_rect(1, mn[0], mx[0], mn[2], mx[2], mn[1], o.material, true, oi, 2);
                    push_rect(1, mn[0], mx[0], mn[2], mx[2], mx[1], o.material, false, oi, 3);
                    push_rect(0, mn[1], mx[1], mn[2], mx[2], mn[0], o.material, true, oi, 4);
                    push_rect(0, mn[1], mx[1], mn[2], mx[2], mx[0], o.material, false, oi, 5);
                }
                r.kind = RT_OBJ_RECTS;
                r.count = 6;
                r.xform = add_xform(x);
                break;
            case ObjectDef::Rect:  // axis rule of src/config.rs:247-279
                if (prim_first_[(size_t)oi] < 0) {
                    prim_first_[(size_t)oi] = (int)out_.rects.size();
                    const double *c0 = o.p0, *c1 = o.p1;
                    if (o.rect_axis == 0) push_rect(0, c0[1], c1[1], c0[2], c1[2], c0[0], o.material, o.facing_forward, oi, 0);
                    else if (o.rect_axis == 1) push_rect(1, c0[0], c1[0], c0[2], c1[2], c0[1], o.material, o.facing_forward, oi, 0);
                    else push_rect(2, c0[0], c1[0], c0[1], c1[1], c0[2], o.material, o.facing_forward, oi, 0);
                }
                r.kind = RT_OBJ_RECTS;
                r.xform = add_xform(x);
                break;
            case ObjectDef::Triangle:
                if (prim_first_[(size_t)oi] < 0) {
                    prim_first_[(size_t)oi] = (int)out_.triangles.size();
                    out_.triangles.push_back(make_triangle(v3(o.p0), v3(o.p1), v3(o.p2), nullptr, o.material, oi, 0));
                }
                r.kind = RT_OBJ_TRIANGLE;
                r.xform = add_xform(x);
                break;
            case ObjectDef::Plane:  // Plane::new, src/objects.rs:577-595
                if (prim_first_[(size_t)oi] < 0) {
                    rt_plane p;
                    V3 a = v3(o.p0), b = v3(o.p1), c = v3(o.p2);
                    V3 vec1 = unit(b - a);
                    V3 normal = unit(cross(vec1, c - a));
                    V3 vec2 = cross(normal, vec1);
                    store(p.point, a); store(p.vec1, vec1); store(p.vec2, vec2); store(p.normal, normal);
                    p.uv_repeat = o.uv_repeat;
                    p.material = o.material;
                    p.owner = oi;
                    prim_first_[(size_t)oi] = (int)out_.planes.size();
                    out_.planes.push_back(p);
                }
                r.kind = RT_OBJ_PLANE;
                r.xform = add_xform(x);
                break;
            case ObjectDef::Mesh:  // load_mesh: BVHNode::from_vec(triangles, TIME_MIN, TIME_MAX), :719-762
                if (blas_root_[(size_t)oi] < 0) {
                    const MeshData& md = sc_.meshes[(size_t)o.mesh];
                    // The triangle records are laid out in the depth-first leaf order of the tree (neighbours in
                    // space are neighbours in memory: the walks fetch them through 128-byte cache lines); `sub`
                    // keeps the triangle's index in the mesh.
                    std::vector<BuildItem> items;
                    items.reserve(md.triangles.size());
                    for (size_t k = 0; k < md.triangles.size(); k++) {
                        const MeshTriangle& t = md.triangles[k];
                        Box b = tri_box(v3(t.p[0]), v3(t.p[1]), v3(t.p[2]));
                        items.push_back(BuildItem{(int32_t)k, b, b});
                    }
                    int d = 0;
                    leaf_base_ = (int32_t)out_.triangles.size();
                    const auto t_build = std::chrono::steady_clock::now();
                    blas_root_[(size_t)oi] = build_bvh(items, 0, items.size(), RT_LEAF_TRIANGLE, &d, nullptr);
                    if (std::getenv("RT_DEBUG"))
                        std::fprintf(stderr, "mesh BVH of %zu triangles: %.1f ms\n", items.size(),
                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count());
                    leaf_base_ = -1;
                    blas_depth_[(size_t)oi] = d;
                    const size_t first = out_.triangles.size();
                    out_.triangles.resize(first + items.size());
                    auto make_range = [&](size_t a, size_t b) {
                        for (size_t k = a; k < b; k++) {
                            const BuildItem& it = items[k];
                            const MeshTriangle& t = md.triangles[(size_t)it.payload];
                            V3 n = v3(t.normal);
                            out_.triangles[first + k] = make_triangle(v3(t.p[0]), v3(t.p[1]), v3(t.p[2]),
                                                                      t.has_normal ? &n : nullptr, o.material, oi, it.payload);
                        }
                    };
                    if (items.size() >= 65536) {  // slices as tasks (every record is written by one of them)
                        const size_t slices = 8, per = (items.size() + slices - 1) / slices;
                        std::vector<std::future<void>> tasks;
                        for (size_t a = 0; a < items.size(); a += per)
                            tasks.push_back(std::async(std::launch::async, make_range, a, std::min(items.size(), a + per)));
                        for (auto& t : tasks) t.get();
                    } else {
                        make_range(0, items.size());
                    }
                }
                r.kind = RT_OBJ_BLAS;
                r.first = blas_root_[(size_t)oi];
                r.xform = add_xform(x);
                *depth = blas_depth_[(size_t)oi] + 1;
                out_.objects.push_back(r);
                return (int32_t)out_.objects.size() - 1;
            case ObjectDef::Spotlight: {  // Spotlight::new / new_primitive, src/objects.rs:638-701
                if (prim_first_[(size_t)oi] < 0) {
                    prim_first_[(size_t)oi] = (int)out_.rects.size();
                    double mn[3] = {-o.width / 2.0, -o.length, -o.width / 2.0};
                    double mx[3] = {o.width / 2.0, 0, o.width / 2.0};
                    int dark = spot_material_[(size_t)oi], light = dark + 1;
                    push_rect(2, mn[0], mx[0], mn[1], mx[1], mn[2], dark, true, oi, 0);
                    push_rect(2, mn[0], mx[0], mn[1], mx[1], mx[2], dark, false, oi, 1);
                    push_rect(1, mn[0], mx[0], mn[2], mx[2], mn[1], light, true, oi, 2);
                    push_rect(0, mn[1], mx[1], mn[2], mx[2], mn[0], dark, true, oi, 3);
                    push_rect(0, mn[1], mx[1], mn[2], mx[2], mx[0], dark, false, oi, 4);
                }
                double th1, th2;
                spotlight_angles(o, &th1, &th2);
                Xf chain = then_rotate(then_rotate(then_translate(x, v3(o.p0)), 1, th2), 2, th1);
                r.kind = RT_OBJ_RECTS;
                r.count = 5;
                r.xform = add_xform(chain);
                break;
            }
            case ObjectDef::ConstantMedium: {  // src/objects.rs:384-460
                int bdepth = 0;
                // (ConstantMedium::hit hands the ray it was given to boundary.hit: the boundary sees the same
                // transform context, src/objects.rs:404-409)
                int32_t boundary = instantiate(o.prototype, Xf(), ctx, -1, &bdepth);
                // The boundary is any Hittable, another medium included (src/objects.rs:388-391).  The kernels answer a
                // boundary query that is itself a medium test one level deep; deeper nestings are refused here.
                if (out_.objects[(size_t)boundary].kind == RT_OBJ_MEDIUM &&
                    out_.objects[(size_t)out_.media[(size_t)out_.objects[(size_t)boundary].first].boundary_object].kind == RT_OBJ_MEDIUM)
                    throw SceneError("Object " + o.name +
                                     ": constantMedium boundaries nested more than two deep (a medium bounded by a medium "
                                     "bounded by a medium) are not supported by the GPU path");
                rt_medium m;
                m.neg_inv_density = -1.0 / o.density;
                m.boundary_object = boundary;
                m.material = o.material;
                m.owner = oi;
                m.rng_key = hash_combine(ctx, (uint32_t)oi) & 0x7fffffffu;
                out_.media.push_back(m);
                r.kind = RT_OBJ_MEDIUM;
                r.first = (int32_t)out_.media.size() - 1;
                r.xform = add_xform(x);
                *depth = bdepth;
                out_.objects.push_back(r);
                return (int32_t)out_.objects.size() - 1;
            }
        }
        r.first = prim_first_[(size_t)oi];
        out_.objects.push_back(r);
        return (int32_t)out_.objects.size() - 1;
    }

    // Triangle::new / with_normal, src/objects.rs:472-509.
    static rt_triangle make_triangle(V3 a, V3 b, V3 c, const V3* supplied, int material, int owner, int sub) {
        rt_triangle t;
        std::memset(&t, 0, sizeof(t));
        V3 vec1 = b - a, vec2 = c - a;
        V3 possible = unit(cross(vec1, vec2));
        V3 normal = possible;
        if (supplied) normal = dot(*supplied, possible) < 0.0 ? V3{-supplied->x, -supplied->y, -supplied->z} : *supplied;
        store(t.point, a); store(t.vec1, vec1); store(t.vec2, vec2); store(t.normal, normal);
        t.material = material; t.owner = owner; t.sub = sub;
        return t;
    }

    const Scene& sc_;
    FlatScene out_;
    std::vector<int> prim_first_, blas_root_, blas_depth_, spot_material_;
};

}  // namespace

rt_scene_desc FlatScene::desc() const {
    rt_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.nodes = nodes.data(); d.n_nodes = nodes.size();
    d.objects = objects.data(); d.n_objects = objects.size();
    d.xforms = xforms.data(); d.n_xforms = xforms.size();
    d.spheres = spheres.data(); d.n_spheres = spheres.size();
    d.moving_spheres = moving_spheres.data(); d.n_moving_spheres = moving_spheres.size();
    d.rects = rects.data(); d.n_rects = rects.size();
    d.triangles = triangles.data(); d.n_triangles = triangles.size();
    d.planes = planes.data(); d.n_planes = planes.size();
    d.media = media.data(); d.n_media = media.size();
    d.materials = materials.data(); d.n_materials = materials.size();
    d.images = images.data(); d.n_images = images.size();
    d.texels = texels.data(); d.n_texel_bytes = texels.size();
    d.top_list = top_list.data(); d.n_top = top_list.size();
    d.tlas_root = tlas_root;
    d.stack_depth = stack_depth;
    d.camera = camera;
    d.sky = sky;
    return d;
}

FlatScene flatten_scene(const Scene& scene) { return Flattener(scene).run(); }

}  // namespace rt
