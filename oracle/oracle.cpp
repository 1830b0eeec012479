// TEST INFRASTRUCTURE - CPU oracle.  Not part of the product; see oracle/README.md.
//
// A C++ f64 restatement of the reference's CPU path tracer (FBemf/raytracer @ v0), written
// by reading the Rust sources as text.  It keeps the reference's object graph (heap nodes
// behind virtual calls), its recursion, its operation order and its quirks; it differs in
// two declared ways only:
//   (i)  the BVH split axis is the longest axis of the node's bounds instead of
//        rand::thread_rng().gen_range(0..3) (src/hitting.rs:158), and
//   (ii) every random draw comes from the counter-based streams of oracle/philox.hpp
//        instead of thread_rng (draw sites: SURVEY.md section 8a row a10).
// Build with -ffp-contract=off: rustc does not contract a*b+c into fma.
//
// Pinning: the only known-answer tests the reference holds for this path are
// test_cross_product / test_refract / test_line_plane (src/math.rs:319-356); they are
// replayed by tests/test_oracle_kat.py through the orc_kat_* entry points.  Everything
// else (hit records, BVH, materials, images) is pinned by nothing in the reference:
// "parity unpinned" beyond those three (the reference is Rust and cannot be built here).
//
// Each function cites the reference file:line it follows.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../raytracer_amd/csrc/host/scene_config.hpp"
#include "philox.hpp"

namespace orc {

static const double INF = std::numeric_limits<double>::infinity();
static const double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI

// ---------------------------------------------------------------- src/math.rs:9-191
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    explicit Vec3(const double* p) : x(p[0]), y(p[1]), z(p[2]) {}
    double length_squared() const { return x * x + y * y + z * z; }  // :28-30
    double length() const { return std::sqrt(length_squared()); }    // :31-33
    Vec3 unit_vector() const { return Vec3(x / length(), y / length(), z / length()); }  // :34-40
    bool near_zero() const {  // :41-44
        const double d = 1e-8;
        return std::fabs(x) < d && std::fabs(y) < d && std::fabs(z) < d;
    }
    double operator[](int i) const { return i == 0 ? x : i == 1 ? y : z; }
    double& operator[](int i) { return i == 0 ? x : i == 1 ? y : z; }
};
typedef Vec3 Point3;
typedef Vec3 Colour;

static inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
static inline Vec3 operator*(Vec3 a, double s) { return Vec3(a.x * s, a.y * s, a.z * s); }
static inline Vec3 operator*(double s, Vec3 a) { return a * s; }  // :138-143
static inline Vec3 operator/(Vec3 a, double s) { return Vec3(a.x / s, a.y / s, a.z / s); }

static inline Vec3 cross(Vec3 l, Vec3 r) {  // :47-53
    return Vec3(l.y * r.z - l.z * r.y, l.z * r.x - l.x * r.z, l.x * r.y - l.y * r.x);
}
static inline double dot(Vec3 l, Vec3 r) { return l.x * r.x + l.y * r.y + l.z * r.z; }  // :55-57
static inline Vec3 coeff(Vec3 l, Vec3 r) { return Vec3(l.x * r.x, l.y * r.y, l.z * r.z); }  // :59-65
static inline Vec3 reflect(Vec3 direction, Vec3 normal) {  // :67-69
    return direction - 2.0 * dot(direction, normal) * normal;
}
static inline Vec3 refract(Vec3 direction, Vec3 normal, double etai_over_etat) {  // :71-78
    direction = direction.unit_vector();
    double cos_theta = std::fmin(dot(-direction, normal), 1.0);
    Vec3 r_out_perpendicular = etai_over_etat * (direction + cos_theta * normal);
    Vec3 r_out_parallel = -(std::sqrt(std::fabs(1.0 - r_out_perpendicular.length_squared()))) * normal;
    return r_out_perpendicular + r_out_parallel;
}
static inline double clamp(double a, double lo, double hi) {  // :220-228
    if (a < lo) return lo;
    if (a > hi) return hi;
    return a;
}

// ---------------------------------------------------------------- src/math.rs:199-218
struct Ray {
    Point3 origin;
    Vec3 direction;
    double time;
    uint32_t ctx;  // oracle-only: hash of the transform objects entered (volume RNG keying)
    Ray(Point3 o, Vec3 d, double t, uint32_t c = 0) : origin(o), direction(d.unit_vector()), time(t), ctx(c) {}
    Point3 at(double t) const { return origin + t * direction; }
};

// ---------------------------------------------------------------- random draws
struct PathRng {
    uint64_t seed = 0;
    uint32_t pixel = 0, sample = 0, bounce = 0;
    RngStream camera, material;
    void begin_path(uint64_t s, uint32_t px, uint32_t smp) {
        seed = s; pixel = px; sample = smp; bounce = 0;
        camera = RngStream(seed, DOMAIN_CAMERA, pixel, sample, 0);
    }
    void begin_bounce(uint32_t b) {
        bounce = b;
        material = RngStream(seed, DOMAIN_MATERIAL, pixel, sample, bounce);
    }
    double medium_xi(uint32_t instance_hash) {
        RngStream s(seed, DOMAIN_MEDIUM | (instance_hash & 0x7fffffffu), pixel, sample, bounce);
        return s.next();
    }
};
static thread_local PathRng tl_rng;

struct Counters {
    uint64_t rays = 0, nodes = 0, prims = 0, leaves = 0, xforms = 0;
};
static thread_local Counters tl_count;

struct Candidate {
    double t;
    int owner, sub;
};
static thread_local std::vector<Candidate>* tl_cands = nullptr;
static thread_local int tl_cand_mute = 0;
// Tie detection only (orc_trace's second pass): walk every BVH node the ray's line pierces,
// without the shrinking upper bound, so that a candidate is never hidden by a box that the
// `max_dist <= min_dist` rule culls because it starts exactly at the current best distance.
static thread_local bool tl_exhaustive = false;
static inline void note_candidate(double t, double min_dist, int owner, int sub) {
    if (tl_cands && !tl_cand_mute && t >= min_dist && t == t) tl_cands->push_back({t, owner, sub});
}

static Vec3 random_in_unit_sphere(RngStream& rng) {  // src/math.rs:234-246
    for (;;) {
        double px = rng.range(-1.0, 1.0);
        double py = rng.range(-1.0, 1.0);
        double pz = rng.range(-1.0, 1.0);
        Vec3 p(px, py, pz);
        if (p.length_squared() <= 1.0) return p;
    }
}
static Vec3 random_unit_vector(RngStream& rng) { return random_in_unit_sphere(rng).unit_vector(); }  // :230-232
static Vec3 random_in_unit_disc(RngStream& rng) {  // :248-256
    for (;;) {
        double px = rng.range(-1.0, 1.0);
        double py = rng.range(-1.0, 1.0);
        Vec3 p(px, py, 0.0);
        if (p.length_squared() <= 1.0) return p;
    }
}

// ---------------------------------------------------------------- src/math.rs:258-317
static bool distance_to_sphere(const Ray& ray, Point3 centre, double radius, double min_dist,
                               double max_dist, double* out) {
    Vec3 oc = ray.origin - centre;
    double a = ray.direction.length_squared();
    double half_b = dot(oc, ray.direction);
    double c = oc.length_squared() - radius * radius;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrt_d = std::sqrt(discriminant);
    double root_distance = (-half_b - sqrt_d) / a;
    if (root_distance < min_dist || root_distance > max_dist) {
        root_distance = (-half_b + sqrt_d) / a;
        if (root_distance < min_dist || root_distance > max_dist) return false;
    }
    *out = root_distance;
    return true;
}
static void get_sphere_uv(Point3 p, double* u, double* v) {  // :288-294
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + PI;
    *u = phi / (2.0 * PI);
    *v = theta / PI;
}
static bool line_plane_collision(Point3 la, Vec3 lab, Point3 p0, Vec3 p01, Vec3 p02, Vec3* out) {  // :298-317
    const double MARGIN = 0.0000001;
    Vec3 cross_p = cross(p01, p02);
    double determinant = dot(-lab, cross_p);
    if (determinant < -MARGIN || determinant > MARGIN) {
        double denominator = dot(-lab, cross_p);
        double t = dot(cross_p, la - p0) / denominator;
        double u = dot(cross(p02, -lab), la - p0) / denominator;
        double v = dot(cross(-lab, p01), la - p0) / denominator;
        *out = Vec3(t, u, v);
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- src/textures.rs
struct Texture {
    virtual ~Texture() {}
    virtual Colour value(double u, double v, Point3 p) const = 0;
};
struct SolidColour : Texture {  // :14-25
    Colour colour;
    explicit SolidColour(Colour c) : colour(c) {}
    Colour value(double, double, Point3) const override { return colour; }
};
struct ImageTexture : Texture {  // :27-56
    const rt::ImageRGB8* image;
    explicit ImageTexture(const rt::ImageRGB8* img) : image(img) {}
    Colour value(double u, double v, Point3) const override {
        u = clamp(u, 0.0, 1.0);
        v = 1.0 - clamp(v, 0.0, 1.0);  // flip v
        double fi = u * (double)image->width, fj = v * (double)image->height;
        // Rust `as u32` saturates and maps NaN to 0.
        uint32_t i = fi != fi ? 0u : fi <= 0.0 ? 0u : fi >= 4294967295.0 ? 4294967295u : (uint32_t)fi;
        uint32_t j = fj != fj ? 0u : fj <= 0.0 ? 0u : fj >= 4294967295.0 ? 4294967295u : (uint32_t)fj;
        i = std::min(i, (uint32_t)image->width - 1);
        j = std::min(j, (uint32_t)image->height - 1);
        double colour_scale = 1.0 / 255.0;
        const uint8_t* px = &image->rgb[((size_t)j * image->width + i) * 3];
        return colour_scale * Colour((double)px[0], (double)px[1], (double)px[2]);
    }
};

// ---------------------------------------------------------------- src/hitting.rs:33-139
struct Material;
struct HitRecord {
    Point3 intersection;
    Vec3 normal;
    double distance = 0;
    bool front_face = false;
    double surface_u = 0, surface_v = 0;
    const Material* material = nullptr;
    int owner = -1, sub = 0, world_index = -1;  // oracle-only identification
    HitRecord() {}
    HitRecord(const Ray& ray, double dist, Vec3 outward_normal, const Material* m, double u, double v,
              int owner_, int sub_) {  // :57-79
        front_face = dot(ray.direction, outward_normal) < 0.0;
        normal = front_face ? outward_normal.unit_vector() : -outward_normal.unit_vector();
        intersection = ray.at(dist);
        distance = dist;
        material = m;
        surface_u = u;
        surface_v = v;
        owner = owner_;
        sub = sub_;
    }
};

struct Material {
    int index = -1;
    virtual ~Material() {}
    virtual bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const = 0;
    virtual Colour emitted(const HitRecord&) const { return Colour(0, 0, 0); }  // :135-137
};

// ---------------------------------------------------------------- src/materials.rs
struct Lambertian : Material {  // :10-46
    std::shared_ptr<Texture> albedo;
    bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const override {
        Vec3 scatter_direction = hit.normal + random_unit_vector(tl_rng.material);
        if (scatter_direction.near_zero()) scatter_direction = hit.normal;
        *scattered = Ray(hit.intersection, scatter_direction, ray.time);
        *attenuation = albedo->value(hit.surface_u, hit.surface_v, hit.intersection);
        return true;
    }
};
struct Metal : Material {  // :48-70
    Colour albedo;
    double fuzz = 0;
    bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const override {
        Vec3 reflected = reflect(ray.direction.unit_vector(), hit.normal);
        *scattered = Ray(hit.intersection, reflected + fuzz * random_in_unit_sphere(tl_rng.material), ray.time);
        if (dot(scattered->direction, hit.normal) > 0.0) {
            *attenuation = albedo;
            return true;
        }
        return false;
    }
};
static double reflectance(double cosine, double ref_idx) {  // :107-111 (1.1, as written)
    double r0 = (1.1 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5.0);
}
struct Dielectric : Material {  // :72-105
    double index_of_refraction = 1;
    bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const override {
        double refraction_ratio = hit.front_face ? 1.0 / index_of_refraction : index_of_refraction;
        Vec3 unit_direction = ray.direction.unit_vector();
        double cos_theta = std::fmin(dot(-unit_direction, hit.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        double random_fraction = tl_rng.material.range(0.0, 1.0);
        Vec3 direction = (cannot_refract || reflectance(cos_theta, refraction_ratio) > random_fraction)
                             ? reflect(unit_direction, hit.normal)
                             : refract(unit_direction, hit.normal, refraction_ratio);
        *scattered = Ray(hit.intersection, direction, ray.time);
        *attenuation = Colour(1.0, 1.0, 1.0);
        return true;
    }
};
struct DiffuseLight : Material {  // :113-140
    std::shared_ptr<Texture> emit;
    bool scatter(const Ray&, const HitRecord&, Ray*, Colour*) const override { return false; }
    Colour emitted(const HitRecord& hit) const override {
        if (hit.front_face) return emit->value(hit.surface_u, hit.surface_v, hit.intersection);
        return Colour(0, 0, 0);
    }
};
struct Isotropic : Material {  // :142-157
    std::shared_ptr<Texture> albedo;
    bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const override {
        *scattered = Ray(hit.intersection, random_in_unit_sphere(tl_rng.material), ray.time);
        *attenuation = albedo->value(hit.surface_u, hit.surface_v, hit.intersection);
        return true;
    }
};
struct Checkered : Material {  // :159-192
    std::shared_ptr<Material> odd, even;
    double tile_density = 0;
    bool scatter(const Ray& ray, const HitRecord& hit, Ray* scattered, Colour* attenuation) const override {
        double sines = std::sin(tile_density * PI * hit.surface_u) * std::sin(tile_density * PI * hit.surface_v);
        if (sines < 0.0) return odd->scatter(ray, hit, scattered, attenuation);
        return even->scatter(ray, hit, scattered, attenuation);
    }
    Colour emitted(const HitRecord& hit) const override {  // no PI here, as written (:176-177)
        double sines = std::sin(tile_density * hit.surface_u) * std::sin(tile_density * hit.surface_v);
        if (sines < 0.0) return odd->emitted(hit);
        return even->emitted(hit);
    }
};

// ---------------------------------------------------------------- src/hitting.rs:82-282
struct AABB {
    Point3 minimum, maximum;
    bool intersects(const Ray& ray, double min_dist, double max_dist) const {  // :250-267
        for (int a = 0; a < 3; a++) {
            double t0 = std::fmin((minimum[a] - ray.origin[a]) / ray.direction[a],
                                  (maximum[a] - ray.origin[a]) / ray.direction[a]);
            double t1 = std::fmax((minimum[a] - ray.origin[a]) / ray.direction[a],
                                  (maximum[a] - ray.origin[a]) / ray.direction[a]);
            min_dist = std::fmax(t0, min_dist);
            max_dist = std::fmin(t1, max_dist);
            if (max_dist <= min_dist) return false;
        }
        return true;
    }
};
static AABB surrounding_box(const AABB& b0, const AABB& b1) {  // :270-282
    AABB r;
    r.minimum = Point3(std::fmin(b0.minimum.x, b1.minimum.x), std::fmin(b0.minimum.y, b1.minimum.y),
                       std::fmin(b0.minimum.z, b1.minimum.z));
    r.maximum = Point3(std::fmax(b0.maximum.x, b1.maximum.x), std::fmax(b0.maximum.y, b1.maximum.y),
                       std::fmax(b0.maximum.z, b1.maximum.z));
    return r;
}

struct Hittable {
    virtual ~Hittable() {}
    virtual bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const = 0;
    virtual bool bounding_box(double time0, double time1, AABB* out) const = 0;
};
typedef std::shared_ptr<Hittable> HittablePtr;

struct HittableList : Hittable {  // impl Hittable for Vec<Arc<dyn Hittable>>, :88-131
    std::vector<HittablePtr> items;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        bool have = false;
        HitRecord acc;
        for (auto& x : items) {
            HitRecord next;
            if (x->hit(ray, min_dist, max_dist, &next)) {
                if (!have || next.distance < acc.distance) acc = next;  // strict <: the first wins ties
                have = true;
            }
        }
        if (have) *out = acc;
        return have;
    }
    bool bounding_box(double time0, double time1, AABB* out) const override {
        if (items.empty()) return false;
        AABB working;
        if (!items[0]->bounding_box(time0, time1, &working)) return false;
        for (auto& obj : items) {
            AABB nb;
            if (!obj->bounding_box(time0, time1, &nb)) return false;
            working = surrounding_box(nb, working);
        }
        *out = working;
        return true;
    }
};

struct BVHNode : Hittable {  // :141-240
    HittablePtr left, right;
    AABB bbox;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {  // :212-229
        tl_count.nodes++;
        if (tl_exhaustive) {
            if (!bbox.intersects(ray, min_dist, INF)) return false;
            HitRecord hl, hr;
            bool l = left->hit(ray, min_dist, INF, &hl), r = right->hit(ray, min_dist, INF, &hr);
            if (l && (!r || hl.distance < hr.distance)) *out = hl;
            else if (r) *out = hr;
            return l || r;
        }
        if (!bbox.intersects(ray, min_dist, max_dist)) return false;
        HitRecord hit_left;
        if (left->hit(ray, min_dist, max_dist, &hit_left)) {
            HitRecord hit_right;
            if (right->hit(ray, min_dist, std::fmin(hit_left.distance, max_dist), &hit_right)) {
                *out = hit_right;
            } else {
                *out = hit_left;
            }
            return true;
        }
        return right->hit(ray, min_dist, max_dist, out);
    }
    bool bounding_box(double, double, AABB* out) const override {
        *out = bbox;
        return true;
    }
};

// BVHNode::from_vec, :148-197.  Deviation (i): deterministic axis = longest axis of the
// union of the boxed objects' bounds at (time0, time1), ties to the lower axis.
static HittablePtr bvh_from_vec(std::vector<HittablePtr> objects, double time0, double time1) {
    std::vector<HittablePtr> no_bbox, boxed;
    AABB tmp;
    for (auto& x : objects)
        (x->bounding_box(time0, time1, &tmp) ? boxed : no_bbox).push_back(x);
    int axis = 0;
    if (!boxed.empty()) {
        AABB all;
        boxed[0]->bounding_box(time0, time1, &all);
        for (auto& x : boxed) {
            x->bounding_box(time0, time1, &tmp);
            all = surrounding_box(all, tmp);
        }
        double best = -1.0;
        for (int a = 0; a < 3; a++) {
            double e = all.maximum[a] - all.minimum[a];
            if (e > best) {
                best = e;
                axis = a;
            }
        }
    }
    // bbox_compare (:199-209): minimum[axis] of bounding_box(0.0, 0.0); stable like sort_by.
    std::stable_sort(boxed.begin(), boxed.end(), [axis](const HittablePtr& a, const HittablePtr& b) {
        AABB ba, bb;
        a->bounding_box(0.0, 0.0, &ba);
        b->bounding_box(0.0, 0.0, &bb);
        return ba.minimum[axis] < bb.minimum[axis];
    });
    if (boxed.empty()) {
        if (no_bbox.empty()) throw rt::SceneError("BVHNode cannot be created from empty slice");
        auto list = std::make_shared<HittableList>();
        list->items = no_bbox;
        return list;
    }
    if (boxed.size() == 1) {
        auto list = std::make_shared<HittableList>();
        list->items = no_bbox;
        list->items.push_back(boxed.back());
        return list;
    }
    size_t halfway = boxed.size() / 2;
    std::vector<HittablePtr> left_objects(boxed.begin(), boxed.begin() + (long)halfway);
    std::vector<HittablePtr> right_objects(boxed.begin() + (long)halfway, boxed.end());
    auto node = std::make_shared<BVHNode>();
    node->left = bvh_from_vec(left_objects, time0, time1);
    node->right = bvh_from_vec(right_objects, time0, time1);
    AABB lb, rb;
    if (!node->left->bounding_box(time0, time1, &lb) || !node->right->bounding_box(time0, time1, &rb))
        throw rt::SceneError("BHVNode unable to find bbox of subtree");
    node->bbox = surrounding_box(lb, rb);
    if (no_bbox.empty()) return node;
    auto list = std::make_shared<HittableList>();
    list->items.push_back(node);
    for (auto& x : no_bbox) list->items.push_back(x);
    return list;
}

// ---------------------------------------------------------------- src/objects.rs
struct Sphere : Hittable {  // :16-64
    Point3 centre;
    double radius;
    const Material* material;
    int owner;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        double root_distance;
        if (tl_cands) {
            double any;
            if (distance_to_sphere(ray, centre, radius, min_dist, INF, &any)) note_candidate(any, min_dist, owner, 0);
        }
        if (!distance_to_sphere(ray, centre, radius, min_dist, max_dist, &root_distance)) return false;
        Vec3 outward_normal = (ray.at(root_distance) - centre) / radius;
        double u, v;
        get_sphere_uv(outward_normal, &u, &v);
        *out = HitRecord(ray, root_distance, outward_normal, material, u, v, owner, 0);
        return true;
    }
    bool bounding_box(double, double, AABB* out) const override {
        out->minimum = centre - Vec3(radius, radius, radius);
        out->maximum = centre + Vec3(radius, radius, radius);
        return true;
    }
};

struct MovingSphere : Hittable {  // :66-135
    Point3 centre0, centre1;
    double time0, time1, radius;
    const Material* material;
    int owner;
    Point3 centre(double time) const {  // :94-97
        return centre0 + ((time - time0) / (time1 - time0)) * (centre1 - centre0);
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        Point3 c = centre(ray.time);
        double root_distance;
        if (tl_cands) {
            double any;
            if (distance_to_sphere(ray, c, radius, min_dist, INF, &any)) note_candidate(any, min_dist, owner, 0);
        }
        if (!distance_to_sphere(ray, c, radius, min_dist, max_dist, &root_distance)) return false;
        Vec3 outward_normal = (ray.at(root_distance) - c) / radius;
        double u, v;
        get_sphere_uv(outward_normal, &u, &v);
        *out = HitRecord(ray, root_distance, outward_normal, material, u, v, owner, 0);
        return true;
    }
    bool bounding_box(double t0, double t1, AABB* out) const override {
        AABB box0{centre(t0) - Vec3(radius, radius, radius), centre(t0) + Vec3(radius, radius, radius)};
        AABB box1{centre(t1) - Vec3(radius, radius, radius), centre(t1) + Vec3(radius, radius, radius)};
        *out = surrounding_box(box0, box1);
        return true;
    }
};

// XYRect / XZRect / YZRect, :192-382.  `axis` is the fixed axis: 2 = XY, 1 = XZ, 0 = YZ;
// (a, b) are the two free axes in the reference's order: XY -> (x, y), XZ -> (x, z), YZ -> (y, z).
struct AxisRect : Hittable {
    int axis;
    double a0, a1, b0, b1, k;
    const Material* material;
    bool facing_positive;
    int owner, sub;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
        double t = (k - ray.origin[axis]) / ray.direction[axis];
        if (t < min_dist || t > max_dist) {
            if (tl_cands && t >= min_dist) {  // candidate beyond max: record if inside the rectangle
                double a = ray.origin[ia] + t * ray.direction[ia], b = ray.origin[ib] + t * ray.direction[ib];
                if (!(a < a0 || a > a1 || b < b0 || b > b1)) note_candidate(t, min_dist, owner, sub);
            }
            return false;
        }
        double a = ray.origin[ia] + t * ray.direction[ia];
        double b = ray.origin[ib] + t * ray.direction[ib];
        if (a < a0 || a > a1 || b < b0 || b > b1) return false;
        if (t != t) return false;  // 0/0: the reference would return a NaN hit; declared not reproduced
        note_candidate(t, min_dist, owner, sub);
        double u = (a - a0) / (a1 - a0);
        double v = (b - b0) / (b1 - b0);
        Vec3 n(0, 0, 0);
        n[axis] = facing_positive ? 1.0 : -1.0;
        *out = HitRecord(ray, t, n, material, u, v, owner, sub);
        return true;
    }
    bool bounding_box(double, double, AABB* out) const override {  // :245-250, 309-314, 373-378
        int ia = axis == 0 ? 1 : 0, ib = axis == 2 ? 1 : 2;
        out->minimum[axis] = k - 0.0001;
        out->maximum[axis] = k + 0.0001;
        out->minimum[ia] = a0; out->maximum[ia] = a1;
        out->minimum[ib] = b0; out->maximum[ib] = b1;
        return true;
    }
};
static HittablePtr make_rect(int axis, double a0, double a1, double b0, double b1, double k,
                             const Material* m, bool facing_positive, int owner, int sub) {
    auto r = std::make_shared<AxisRect>();
    r->axis = axis; r->a0 = a0; r->a1 = a1; r->b0 = b0; r->b1 = b1; r->k = k;
    r->material = m; r->facing_positive = facing_positive; r->owner = owner; r->sub = sub;
    return r;
}

struct Block : Hittable {  // :137-190
    Point3 minimum, maximum;
    HittableList sides;
    Block(Point3 mn, Point3 mx, const Material* m, int owner) : minimum(mn), maximum(mx) {
        sides.items = {
            make_rect(2, mn.x, mx.x, mn.y, mx.y, mn.z, m, true, owner, 0),
            make_rect(2, mn.x, mx.x, mn.y, mx.y, mx.z, m, false, owner, 1),
            make_rect(1, mn.x, mx.x, mn.z, mx.z, mn.y, m, true, owner, 2),
            make_rect(1, mn.x, mx.x, mn.z, mx.z, mx.y, m, false, owner, 3),
            make_rect(0, mn.y, mx.y, mn.z, mx.z, mn.x, m, true, owner, 4),
            make_rect(0, mn.y, mx.y, mn.z, mx.z, mx.x, m, false, owner, 5),
        };
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        return sides.hit(ray, min_dist, max_dist, out);
    }
    bool bounding_box(double, double, AABB* out) const override {
        out->minimum = minimum;
        out->maximum = maximum;
        return true;
    }
};

struct ConstantMedium : Hittable {  // :384-460
    HittablePtr boundary;
    const Material* phase_function;
    double neg_inv_density;
    int owner;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        tl_cand_mute++;
        HitRecord hit1, hit2;
        bool ok = boundary->hit(ray, -INF, INF, &hit1) && boundary->hit(ray, hit1.distance + 0.0001, INF, &hit2);
        tl_cand_mute--;
        if (!ok) return false;
        if (hit1.distance < min_dist) hit1.distance = min_dist;
        if (hit2.distance > max_dist) hit2.distance = max_dist;
        if (hit1.distance >= hit2.distance) return false;
        if (hit1.distance < 0.0) hit1.distance = 0.0;
        double ray_length = ray.direction.length();
        double distance_inside_boundary = (hit2.distance - hit1.distance) * ray_length;
        double xi = tl_rng.medium_xi(hash_combine(ray.ctx, (uint32_t)owner));
        double hit_distance = neg_inv_density * std::log(xi);
        if (hit_distance > distance_inside_boundary) return false;
        double distance = hit1.distance + hit_distance / ray_length;
        HitRecord h;
        h.distance = distance;
        h.intersection = ray.at(distance);
        h.normal = Vec3(1, 0, 0);
        h.front_face = true;
        h.material = phase_function;
        h.surface_u = 0.0;
        h.surface_v = 0.0;
        h.owner = owner;
        h.sub = 0;
        note_candidate(distance, min_dist, owner, 0);
        *out = h;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB* out) const override { return boundary->bounding_box(t0, t1, out); }
};

struct Triangle : Hittable {  // :462-565
    Point3 point;
    Vec3 vec1, vec2, normal;
    AABB bbox;
    const Material* material;
    int owner, sub;
    Triangle(Point3 a, Point3 b, Point3 c, const Vec3* supplied_normal, const Material* m, int owner_, int sub_)
        : point(a), vec1(b - a), vec2(c - a), material(m), owner(owner_), sub(sub_) {
        Vec3 possible_normal = cross(vec1, vec2).unit_vector();
        if (supplied_normal) {  // with_normal, :486-509
            normal = dot(*supplied_normal, possible_normal) < 0.0 ? -*supplied_normal : *supplied_normal;
        } else {
            normal = possible_normal;
        }
        // vertices_to_bbox, :512-534 (unpadded)
        bbox.minimum = Vec3(INF, INF, INF);
        bbox.maximum = Vec3(-INF, -INF, -INF);
        const Point3 vs[3] = {a, b, c};
        for (auto& vtx : vs) {
            bbox.minimum = Vec3(std::fmin(bbox.minimum.x, vtx.x), std::fmin(bbox.minimum.y, vtx.y), std::fmin(bbox.minimum.z, vtx.z));
            bbox.maximum = Vec3(std::fmax(bbox.maximum.x, vtx.x), std::fmax(bbox.maximum.y, vtx.y), std::fmax(bbox.maximum.z, vtx.z));
        }
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        Vec3 s;
        if (!line_plane_collision(ray.origin, ray.direction, point, vec1, vec2, &s)) return false;
        double distance = s[0];
        bool inside = !(s[1] < 0.0 || s[2] < 0.0 || s[1] + s[2] > 1.0);
        if (inside) note_candidate(distance, min_dist, owner, sub);
        if (distance < min_dist || distance > max_dist) return false;
        if (!inside) return false;
        *out = HitRecord(ray, distance, normal, material, s[2], s[1], owner, sub);
        return true;
    }
    bool bounding_box(double, double, AABB* out) const override {
        *out = bbox;
        return true;
    }
};

struct Plane : Hittable {  // :567-629
    Point3 point;
    Vec3 vec1, vec2, normal;
    double uv_repeat;
    const Material* material;
    int owner;
    Plane(Point3 a, Point3 b, Point3 c, double rep, const Material* m, int owner_)
        : point(a), uv_repeat(rep), material(m), owner(owner_) {
        vec1 = (b - a).unit_vector();
        normal = cross(vec1, c - a).unit_vector();
        vec2 = cross(normal, vec1);
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.prims++;
        Vec3 s;
        if (!line_plane_collision(ray.origin, ray.direction, point, vec1, vec2, &s)) return false;
        double distance = s[0];
        note_candidate(distance, min_dist, owner, 0);
        if (distance < min_dist || distance > max_dist) return false;
        double u = std::fmod(s[1], uv_repeat) / uv_repeat;   // Rust `%` on f64 is fmod
        double v = std::fmod(-s[2], uv_repeat) / uv_repeat;
        u = u < 0.0 ? 1.0 + u : u;
        v = v < 0.0 ? 1.0 + v : v;
        *out = HitRecord(ray, distance, normal, material, u, v, owner, 0);
        return true;
    }
    bool bounding_box(double, double, AABB*) const override { return false; }
};

// ---------------------------------------------------------------- src/transforms.rs
struct Translate : Hittable {  // :7-51
    HittablePtr original;
    Vec3 offset;
    int ctx_id = -1;  // config object index, or -1 for the transforms inside a spotlight
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        Ray moved(ray.origin - offset, ray.direction, ray.time,
                  ctx_id >= 0 ? hash_combine(ray.ctx, (uint32_t)ctx_id) : ray.ctx);
        HitRecord h;
        if (!original->hit(moved, min_dist, max_dist, &h)) return false;
        h.intersection = h.intersection + offset;
        *out = h;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB* out) const override {
        AABB bb;
        if (!original->bounding_box(t0, t1, &bb)) return false;
        out->minimum = bb.minimum + offset;
        out->maximum = bb.maximum + offset;
        return true;
    }
};

// RotateX / RotateY / RotateZ, :53-324.  axis 0 rotates (y, z) as RotateX does, axis 1
// rotates (x, z) as RotateY, axis 2 rotates (x, y) as RotateZ.  In the reference all three
// share one pattern on a pair (p, q) of coordinates:
//   to object space:  p' = c*p - s*q ; q' = s*p + c*q
//   back to world:    p  = c*p' + s*q' ; q = -s*p' + c*q'
// with (p, q) = (z, y) for X (:104-105), (x, z) for Y (:195-196), (y, x) for Z (:286-287).
struct Rotate : Hittable {
    HittablePtr original;
    int axis;
    double sin_theta, cos_theta;
    bool has_bbox = false;
    AABB bbox;
    int ctx_id = -1;
    void pq(int* p, int* q) const {
        if (axis == 0) { *p = 2; *q = 1; }
        else if (axis == 1) { *p = 0; *q = 2; }
        else { *p = 1; *q = 0; }
    }
    Rotate(const HittablePtr& orig, int axis_, double radians) : original(orig), axis(axis_) {
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        AABB b;
        if (original->bounding_box(0.0, 1.0, &b)) {  // TIME_MIN, TIME_MAX
            int p, q;
            pq(&p, &q);
            Point3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++)
                    for (int k = 0; k < 3; k++) {  // 27 combos incl. the extrapolating index 2, as written
                        Vec3 corner((double)i * b.maximum.x + (1.0 - (double)i) * b.minimum.x,
                                    (double)j * b.maximum.y + (1.0 - (double)j) * b.minimum.y,
                                    (double)k * b.maximum.z + (1.0 - (double)k) * b.minimum.z);
                        Vec3 tester = corner;
                        tester[p] = cos_theta * corner[p] + sin_theta * corner[q];
                        tester[q] = -sin_theta * corner[p] + cos_theta * corner[q];
                        for (int c = 0; c < 3; c++) {
                            mn[c] = std::fmin(mn[c], tester[c]);
                            mx[c] = std::fmax(mx[c], tester[c]);
                        }
                    }
            bbox.minimum = mn;
            bbox.maximum = mx;
            has_bbox = true;
        }
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        int p, q;
        pq(&p, &q);
        Vec3 origin = ray.origin, direction = ray.direction;
        origin[p] = cos_theta * ray.origin[p] - sin_theta * ray.origin[q];
        origin[q] = sin_theta * ray.origin[p] + cos_theta * ray.origin[q];
        direction[p] = cos_theta * ray.direction[p] - sin_theta * ray.direction[q];
        direction[q] = sin_theta * ray.direction[p] + cos_theta * ray.direction[q];
        Ray rotated(origin, direction, ray.time, ctx_id >= 0 ? hash_combine(ray.ctx, (uint32_t)ctx_id) : ray.ctx);
        HitRecord h;
        if (!original->hit(rotated, min_dist, max_dist, &h)) return false;
        Point3 ip = h.intersection;
        Vec3 n = h.normal;
        h.intersection[p] = cos_theta * ip[p] + sin_theta * ip[q];
        h.intersection[q] = -sin_theta * ip[p] + cos_theta * ip[q];
        h.normal[p] = cos_theta * n[p] + sin_theta * n[q];
        h.normal[q] = -sin_theta * n[p] + cos_theta * n[q];
        *out = h;
        return true;
    }
    bool bounding_box(double, double, AABB* out) const override {
        if (has_bbox) *out = bbox;
        return has_bbox;
    }
};

struct Spotlight : Hittable {  // :631-717 (new_primitive)
    Point3 minimum, maximum;
    HittableList panes;
    Spotlight(Point3 mn, Point3 mx, const Material* dark, const Material* light, int owner)
        : minimum(mn), maximum(mx) {
        panes.items = {
            make_rect(2, mn.x, mx.x, mn.y, mx.y, mn.z, dark, true, owner, 0),
            make_rect(2, mn.x, mx.x, mn.y, mx.y, mx.z, dark, false, owner, 1),
            make_rect(1, mn.x, mx.x, mn.z, mx.z, mn.y, light, true, owner, 2),
            make_rect(0, mn.y, mx.y, mn.z, mx.z, mn.x, dark, true, owner, 3),
            make_rect(0, mn.y, mx.y, mn.z, mx.z, mx.x, dark, false, owner, 4),
        };
    }
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        return panes.hit(ray, min_dist, max_dist, out);
    }
    bool bounding_box(double, double, AABB* out) const override {
        out->minimum = minimum;
        out->maximum = maximum;
        return true;
    }
};

static double signum(double x) {  // f64::signum: +-1 for +-0, NaN for NaN
    if (x != x) return x;
    return std::signbit(x) ? -1.0 : 1.0;
}

static HittablePtr make_spotlight(Point3 looking_from, Point3 looking_at, double length, double width,
                                  const Material* dark, const Material* light, int owner) {  // :665-701
    HittablePtr spot = std::make_shared<Spotlight>(Point3(-width / 2.0, -length, -width / 2.0),
                                                   Point3(width / 2.0, 0, width / 2.0), dark, light, owner);
    Vec3 direction = looking_at - looking_from;
    Vec3 direction_x(direction.x, 0, 0), direction_z(0, 0, direction.z);
    double theta1 = direction.length() != 0.0 ? -std::acos(dot(Vec3(0, 1, 0), direction.unit_vector())) : 0.0;
    double theta2 = (direction_x.x != 0.0 || direction_z.z != 0.0)
                        ? std::acos(dot(Vec3(1, 0, 0), (direction_x + direction_z).unit_vector())) * -signum(direction_z.z)
                        : 0.0;
    HittablePtr rz = std::make_shared<Rotate>(spot, 2, theta1);
    HittablePtr ry = std::make_shared<Rotate>(rz, 1, theta2);
    auto tr = std::make_shared<Translate>();
    tr->original = ry;
    tr->offset = looking_from;
    return tr;
}

// Oracle-only transparent wrapper that stamps which `world` entry a hit came through.
struct WorldEntry : Hittable {
    HittablePtr inner;
    int index;
    bool has_xform = false;
    bool hit(const Ray& ray, double min_dist, double max_dist, HitRecord* out) const override {
        tl_count.leaves++;
        if (has_xform) tl_count.xforms++;
        if (!inner->hit(ray, min_dist, max_dist, out)) return false;
        out->world_index = index;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB* out) const override { return inner->bounding_box(t0, t1, out); }
};

// ---------------------------------------------------------------- src/camera.rs
struct Camera {  // :11-76
    Point3 origin, lower_left_corner;
    Vec3 horizontal, vertical, u, v, w;
    double lens_radius, start_time, end_time;
    Camera() {}
    Camera(Point3 look_from, Point3 look_at, Vec3 direction_up, double vertical_fov, double aspect_ratio,
           double aperture, double focus_dist, double start_time_, double end_time_) {
        double theta = vertical_fov * (PI / 180.0);  // f64::to_radians
        double viewport_height = 2.0 * std::tan(theta / 2.0);
        double viewport_width = aspect_ratio * viewport_height;
        w = (look_from - look_at).unit_vector();
        u = cross(direction_up, w).unit_vector();
        v = cross(w, u);
        origin = look_from;
        horizontal = focus_dist * viewport_width * u;
        vertical = focus_dist * viewport_height * v;
        lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - focus_dist * w;
        lens_radius = aperture / 2.0;
        start_time = start_time_;
        end_time = end_time_;
    }
    Ray find_ray(double s, double t, RngStream& rng) const {  // :67-75
        Vec3 rd = lens_radius * random_in_unit_disc(rng);
        Vec3 offset = u * rd.x + v * rd.y;
        Point3 o = origin + offset;
        Vec3 d = lower_left_corner + s * horizontal + t * vertical - origin - offset;
        double time = rng.range(start_time, end_time);  // gen_range(start..=end), distribution-equivalent
        return Ray(o, d, time);
    }
};

struct Sky {  // src/config.rs:405-421, src/camera.rs:78-86
    bool gradient = false;
    Colour colour, col1, col2;
    Vec3 unit_dir;
    Colour operator()(const Ray& ray) const {
        if (!gradient) return colour;
        double gradient_pos = dot(unit_dir, ray.direction.unit_vector());
        double t = 0.5 * (gradient_pos + 1.0);
        return 1.0 * ((1.0 - t) * col1 + t * col2);
    }
};

// ---------------------------------------------------------------- src/hitting.rs:11-31
static Colour cast_ray(const Ray& ray, const Hittable& world, const Sky& sky, uint32_t bounces, uint32_t depth) {
    if (bounces == 0) return Colour(0, 0, 0);
    tl_rng.begin_bounce(depth);
    tl_count.rays++;
    HitRecord hit;
    if (world.hit(ray, 0.001, INF, &hit)) {
        Colour emitted = hit.material->emitted(hit);
        Ray new_ray(Point3(), Vec3(1, 0, 0), 0);
        Colour attenuation;
        if (hit.material->scatter(ray, hit, &new_ray, &attenuation)) {
            return emitted + coeff(attenuation, cast_ray(new_ray, world, sky, bounces - 1, depth + 1));
        }
        return emitted;
    }
    return sky(ray);
}

// ---------------------------------------------------------------- src/config.rs:19-63
struct OracleScene {
    rt::Scene cfg;
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::shared_ptr<Material>> materials;  // config materials, then 2 per spotlight
    std::vector<HittablePtr> objects;
    HittablePtr world;
    Camera camera;
    Sky sky;
};

static Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }

static OracleScene* build_scene(rt::Scene&& cfg_in) {
    std::unique_ptr<OracleScene> S(new OracleScene);
    S->cfg = std::move(cfg_in);
    const rt::Scene& cfg = S->cfg;
    for (auto& t : cfg.textures) {
        if (t.type == rt::TextureDef::SolidColour)
            S->textures.push_back(std::make_shared<SolidColour>(v3(t.colour)));
        else
            S->textures.push_back(std::make_shared<ImageTexture>(&cfg.images[(size_t)t.image]));
    }
    S->materials.resize(cfg.materials.size());
    for (int mi : cfg.material_order) {
        const rt::MaterialDef& m = cfg.materials[(size_t)mi];
        std::shared_ptr<Material> out;
        switch (m.type) {
            case rt::MaterialDef::Lambertian: {
                auto p = std::make_shared<Lambertian>();
                p->albedo = S->textures[(size_t)m.texture];
                out = p;
                break;
            }
            case rt::MaterialDef::Metal: {
                auto p = std::make_shared<Metal>();
                p->albedo = v3(m.albedo);
                p->fuzz = m.fuzz;
                out = p;
                break;
            }
            case rt::MaterialDef::Dielectric: {
                auto p = std::make_shared<Dielectric>();
                p->index_of_refraction = m.index_of_refraction;
                out = p;
                break;
            }
            case rt::MaterialDef::DiffuseLight: {
                auto p = std::make_shared<DiffuseLight>();
                p->emit = S->textures[(size_t)m.texture];
                out = p;
                break;
            }
            case rt::MaterialDef::Isotropic: {
                auto p = std::make_shared<Isotropic>();
                p->albedo = S->textures[(size_t)m.texture];
                out = p;
                break;
            }
            case rt::MaterialDef::Checkered: {
                auto p = std::make_shared<Checkered>();
                p->odd = S->materials[(size_t)m.odd];
                p->even = S->materials[(size_t)m.even];
                p->tile_density = m.tile_density;
                out = p;
                break;
            }
        }
        out->index = mi;
        S->materials[(size_t)mi] = out;
    }
    // Implicit spotlight materials (src/objects.rs:639-641): numbered after the config
    // materials, two per spotlight object in config order (dark, light).
    std::vector<int> spot_ordinal(cfg.objects.size(), -1);
    {
        int n = 0;
        for (size_t i = 0; i < cfg.objects.size(); i++)
            if (cfg.objects[i].type == rt::ObjectDef::Spotlight) spot_ordinal[i] = n++;
        S->materials.resize(cfg.materials.size() + 2 * (size_t)n);
    }
    S->objects.resize(cfg.objects.size());
    for (int oi : cfg.object_order) {
        const rt::ObjectDef& o = cfg.objects[(size_t)oi];
        const Material* mat = o.material >= 0 ? S->materials[(size_t)o.material].get() : nullptr;
        HittablePtr out;
        switch (o.type) {
            case rt::ObjectDef::Sphere: {
                auto p = std::make_shared<Sphere>();
                p->centre = v3(o.p0); p->radius = o.radius; p->material = mat; p->owner = oi;
                out = p;
                break;
            }
            case rt::ObjectDef::MovingSphere: {
                auto p = std::make_shared<MovingSphere>();
                p->centre0 = v3(o.p0); p->centre1 = v3(o.p1); p->time0 = o.time0; p->time1 = o.time1;
                p->radius = o.radius; p->material = mat; p->owner = oi;
                out = p;
                break;
            }
            case rt::ObjectDef::Block:
                out = std::make_shared<Block>(v3(o.p0), v3(o.p1), mat, oi);
                break;
            case rt::ObjectDef::Rect: {  // src/config.rs:247-279
                const double *c0 = o.p0, *c1 = o.p1;
                if (o.rect_axis == 0) out = make_rect(0, c0[1], c1[1], c0[2], c1[2], c0[0], mat, o.facing_forward, oi, 0);
                else if (o.rect_axis == 1) out = make_rect(1, c0[0], c1[0], c0[2], c1[2], c0[1], mat, o.facing_forward, oi, 0);
                else out = make_rect(2, c0[0], c1[0], c0[1], c1[1], c0[2], mat, o.facing_forward, oi, 0);
                break;
            }
            case rt::ObjectDef::Triangle:
                out = std::make_shared<Triangle>(v3(o.p0), v3(o.p1), v3(o.p2), nullptr, mat, oi, 0);
                break;
            case rt::ObjectDef::Plane:
                out = std::make_shared<Plane>(v3(o.p0), v3(o.p1), v3(o.p2), o.uv_repeat, mat, oi);
                break;
            case rt::ObjectDef::Mesh: {  // load_mesh, src/objects.rs:719-762
                const rt::MeshData& md = cfg.meshes[(size_t)o.mesh];
                std::vector<HittablePtr> tris;
                tris.reserve(md.triangles.size());
                for (size_t k = 0; k < md.triangles.size(); k++) {
                    const rt::MeshTriangle& t = md.triangles[k];
                    Vec3 n = v3(t.normal);
                    tris.push_back(std::make_shared<Triangle>(v3(t.p[0]), v3(t.p[1]), v3(t.p[2]),
                                                              t.has_normal ? &n : nullptr, mat, oi, (int)k));
                }
                out = bvh_from_vec(tris, 0.0, 1.0);  // TIME_MIN, TIME_MAX
                break;
            }
            case rt::ObjectDef::Spotlight: {
                size_t base = cfg.materials.size() + 2 * (size_t)spot_ordinal[(size_t)oi];
                auto dark = std::make_shared<Lambertian>();
                dark->albedo = std::make_shared<SolidColour>(Colour(0, 0, 0));
                dark->index = (int)base;
                auto light = std::make_shared<DiffuseLight>();
                light->emit = std::make_shared<SolidColour>(v3(o.p2));
                light->index = (int)base + 1;
                S->materials[base] = dark;
                S->materials[base + 1] = light;
                out = make_spotlight(v3(o.p0), v3(o.p1), o.length, o.width, dark.get(), light.get(), oi);
                break;
            }
            case rt::ObjectDef::ConstantMedium: {
                auto p = std::make_shared<ConstantMedium>();
                p->boundary = S->objects[(size_t)o.prototype];
                p->phase_function = mat;
                p->neg_inv_density = -1.0 / o.density;
                p->owner = oi;
                out = p;
                break;
            }
            case rt::ObjectDef::Translate: {
                auto p = std::make_shared<Translate>();
                p->original = S->objects[(size_t)o.prototype];
                p->offset = v3(o.p0);
                p->ctx_id = oi;
                out = p;
                break;
            }
            case rt::ObjectDef::RotateX:
            case rt::ObjectDef::RotateY:
            case rt::ObjectDef::RotateZ: {
                int axis = o.type == rt::ObjectDef::RotateX ? 0 : o.type == rt::ObjectDef::RotateY ? 1 : 2;
                auto p = std::make_shared<Rotate>(S->objects[(size_t)o.prototype], axis, o.degrees * (PI / 180.0));
                p->ctx_id = oi;
                out = p;
                break;
            }
        }
        S->objects[(size_t)oi] = out;
    }
    std::vector<HittablePtr> world;
    for (size_t k = 0; k < cfg.world.size(); k++) {
        auto e = std::make_shared<WorldEntry>();
        e->inner = S->objects[(size_t)cfg.world[k]];
        e->index = (int)k;
        {
            rt::ObjectDef::Type ty = cfg.objects[(size_t)cfg.world[k]].type;
            e->has_xform = ty == rt::ObjectDef::Translate || ty == rt::ObjectDef::RotateX || ty == rt::ObjectDef::RotateY ||
                           ty == rt::ObjectDef::RotateZ || ty == rt::ObjectDef::Spotlight;
        }
        world.push_back(e);
    }
    S->world = bvh_from_vec(world, cfg.camera.start_time, cfg.camera.end_time);
    const rt::CameraDef& c = cfg.camera;
    S->camera = Camera(v3(c.look_from), v3(c.look_at), v3(c.direction_up), c.vertical_fov, cfg.aspect(),
                       c.aperture, c.focus_dist, c.start_time, c.end_time);
    if (cfg.background.type == rt::BackgroundDef::PlainColour) {
        S->sky.gradient = false;
        S->sky.colour = v3(cfg.background.colour);
    } else {
        S->sky.gradient = true;
        S->sky.unit_dir = v3(cfg.background.direction).unit_vector();
        S->sky.col1 = v3(cfg.background.colour0);
        S->sky.col2 = v3(cfg.background.colour1);
    }
    return S.release();
}

static void colour_to_raw(Colour c, uint8_t* out) {  // src/main.rs:172-177
    const double ch[3] = {c.x, c.y, c.z};
    for (int k = 0; k < 3; k++) {
        double f = std::floor(255.0 * clamp(std::fabs(ch[k]), 0.0, 0.999));
        out[k] = f != f ? 0 : f <= 0.0 ? 0 : f >= 255.0 ? 255 : (uint8_t)f;  // `as u8` saturates, NaN -> 0
    }
}

static thread_local std::string tl_error;

}  // namespace orc

// =================================================================== C interface (ctypes)
using namespace orc;

extern "C" {

struct orc_ray {
    double o[3], d[3], time, tmin, tmax;
};
struct orc_hit {
    int32_t hit, owner, sub, world_index, front_face, material;
    double t, p[3], n[3], u, v;
    double second_t;  // nearest candidate distance from a *different* primitive (inf if none)
    uint32_t nodes, prims;
};

const char* orc_last_error() { return tl_error.c_str(); }

void* orc_load_scene(const char* path, const char* asset_root) {
    try {
        return build_scene(rt::load_scene_file(path, asset_root ? asset_root : ""));
    } catch (const std::exception& e) {
        tl_error = e.what();
        return nullptr;
    }
}
void* orc_load_scene_text(const char* text, const char* asset_root) {
    try {
        return build_scene(rt::load_scene_text(text, asset_root ? asset_root : ""));
    } catch (const std::exception& e) {
        tl_error = e.what();
        return nullptr;
    }
}
void orc_free_scene(void* s) { delete (OracleScene*)s; }

uint32_t orc_image_height(void* s, uint32_t width) { return ((OracleScene*)s)->cfg.image_height(width); }

// Hittable::hit of the world on explicit rays (directions are normalised as Ray::new does).
// Volume draws use the stream of (seed, pixel = ray index, sample 0, bounce 0).
void orc_trace(void* sv, const orc_ray* rays, uint64_t n, uint64_t seed, orc_hit* out) {
    OracleScene* S = (OracleScene*)sv;
    std::vector<Candidate> cands;
    for (uint64_t i = 0; i < n; i++) {
        Ray r(Vec3(rays[i].o), Vec3(rays[i].d), rays[i].time);
        tl_rng.begin_path(seed, (uint32_t)i, 0);
        tl_rng.begin_bounce(0);
        tl_count = Counters();
        HitRecord h;
        bool ok = S->world->hit(r, rays[i].tmin, rays[i].tmax, &h);
        Counters counted = tl_count;
        if (ok) {  // second pass: every candidate primitive along the ray, for the tie rule
            cands.clear();
            tl_cands = &cands;
            tl_exhaustive = true;
            HitRecord ignore;
            (void)S->world->hit(r, rays[i].tmin, INF, &ignore);
            tl_exhaustive = false;
            tl_cands = nullptr;
        }
        tl_count = counted;
        orc_hit& o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.hit = ok ? 1 : 0;
        o.owner = -1;
        o.world_index = -1;
        o.material = -1;
        o.second_t = INF;
        o.nodes = (uint32_t)tl_count.nodes;
        o.prims = (uint32_t)tl_count.prims;
        if (!ok) continue;
        o.owner = h.owner; o.sub = h.sub; o.world_index = h.world_index;
        o.front_face = h.front_face ? 1 : 0;
        o.material = h.material->index;
        o.t = h.distance;
        o.p[0] = h.intersection.x; o.p[1] = h.intersection.y; o.p[2] = h.intersection.z;
        o.n[0] = h.normal.x; o.n[1] = h.normal.y; o.n[2] = h.normal.z;
        o.u = h.surface_u; o.v = h.surface_v;
        for (auto& c : cands)
            if ((c.owner != h.owner || c.sub != h.sub) && c.t < o.second_t) o.second_t = c.t;
    }
}

// Camera rays exactly as the row loop builds them (src/main.rs:132-135) for sample index
// `sample` of every pixel of rows [j0, j1); row j is the reference's bottom-up row index.
void orc_primary_rays(void* sv, uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t sample,
                      uint64_t seed, orc_ray* out) {
    OracleScene* S = (OracleScene*)sv;
    for (uint32_t j = j0; j < j1; j++)
        for (uint32_t i = 0; i < W; i++) {
            tl_rng.begin_path(seed, j * W + i, sample);
            double u = ((double)i + tl_rng.camera.range(0.0, 1.0)) / (double)(W - 1);
            double v = ((double)j + tl_rng.camera.range(0.0, 1.0)) / (double)(H - 1);
            Ray r = S->camera.find_ray(u, v, tl_rng.camera);
            orc_ray& o = out[(size_t)(j - j0) * W + i];
            o.o[0] = r.origin.x; o.o[1] = r.origin.y; o.o[2] = r.origin.z;
            o.d[0] = r.direction.x; o.d[1] = r.direction.y; o.d[2] = r.direction.z;
            o.time = r.time; o.tmin = 0.001; o.tmax = INF;
        }
}

// The row loop of src/main.rs:117-154 for rows [j0, j1) and sample indices
// [s0, s0 + spp): writes, per pixel (row-major from row j0), the *sum* of radiance over
// those samples into sum_rgb (3 doubles, may be null) and, when rgb8 is non-null, the
// quantised pixel computed from sum / spp exactly as :138-142,172-177.
// Returns the number of world.hit queries issued by cast_ray ("rays").
uint64_t orc_render(void* sv, uint32_t W, uint32_t H, uint32_t j0, uint32_t j1, uint32_t s0, uint32_t spp,
                    uint32_t max_bounces, uint64_t seed, int nthreads, double* sum_rgb, uint8_t* rgb8) {
    OracleScene* S = (OracleScene*)sv;
    if (nthreads < 1) nthreads = 1;
    std::atomic<uint32_t> next_row(0);
    std::atomic<uint64_t> total_rays(0);
    uint32_t nrows = j1 - j0;
    auto worker = [&]() {
        tl_count = Counters();
        for (;;) {
            uint32_t k = next_row.fetch_add(1);
            if (k >= nrows) break;
            uint32_t j = j1 - 1 - k;  // rows are handed out top of the image first (.rev(), :120)
            for (uint32_t i = 0; i < W; i++) {
                Colour colour(0, 0, 0);
                for (uint32_t s = s0; s < s0 + spp; s++) {
                    tl_rng.begin_path(seed, j * W + i, s);
                    double u = ((double)i + tl_rng.camera.range(0.0, 1.0)) / (double)(W - 1);
                    double v = ((double)j + tl_rng.camera.range(0.0, 1.0)) / (double)(H - 1);
                    Ray r = S->camera.find_ray(u, v, tl_rng.camera);
                    colour = colour + cast_ray(r, *S->world, S->sky, max_bounces, 0);
                }
                size_t px = (size_t)(j - j0) * W + i;
                if (sum_rgb) {
                    sum_rgb[px * 3 + 0] = colour.x; sum_rgb[px * 3 + 1] = colour.y; sum_rgb[px * 3 + 2] = colour.z;
                }
                if (rgb8) {
                    Colour mean = colour / (double)spp;
                    Colour gamma(std::sqrt(mean.x), std::sqrt(mean.y), std::sqrt(mean.z));
                    colour_to_raw(gamma, &rgb8[px * 3]);
                }
            }
        }
        total_rays += tl_count.rays;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return total_rays.load();
}

// Mean nodes visited / primitives tested per world.hit over a render (single thread).
void orc_traversal_stats(void* sv, uint32_t W, uint32_t H, uint32_t spp, uint32_t max_bounces, uint64_t seed,
                         uint32_t pixel_stride, double* out5) {
    OracleScene* S = (OracleScene*)sv;
    tl_count = Counters();
    for (uint32_t j = 0; j < H; j += pixel_stride)
        for (uint32_t i = 0; i < W; i += pixel_stride)
            for (uint32_t s = 0; s < spp; s++) {
                tl_rng.begin_path(seed, j * W + i, s);
                double u = ((double)i + tl_rng.camera.range(0.0, 1.0)) / (double)(W - 1);
                double v = ((double)j + tl_rng.camera.range(0.0, 1.0)) / (double)(H - 1);
                Ray r = S->camera.find_ray(u, v, tl_rng.camera);
                (void)cast_ray(r, *S->world, S->sky, max_bounces, 0);
            }
    double r = tl_count.rays ? (double)tl_count.rays : 1.0;
    out5[0] = (double)tl_count.rays;
    out5[1] = (double)tl_count.nodes / r;   // interior BVH nodes box-tested per ray
    out5[2] = (double)tl_count.prims / r;   // primitive hit() calls per ray
    out5[3] = (double)tl_count.leaves / r;  // world-BVH leaves (object instances) visited per ray
    out5[4] = (double)tl_count.xforms / r;  // of which behind a transform chain
}

// ---- known-answer hooks for the reference's own unit tests (src/math.rs:319-356)
void orc_kat_cross(const double* a, const double* b, double* out) {
    Vec3 r = cross(Vec3(a), Vec3(b));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_refract(const double* dir, const double* normal, double eta, double* out) {
    Vec3 r = refract(Vec3(dir), Vec3(normal), eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_kat_line_plane(const double* la, const double* lab, const double* p0, const double* p01,
                       const double* p02, double* out) {
    Vec3 r;
    if (!line_plane_collision(Vec3(la), Vec3(lab), Vec3(p0), Vec3(p01), Vec3(p02), &r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
void orc_kat_reflect(const double* dir, const double* normal, double* out) {
    Vec3 r = reflect(Vec3(dir), Vec3(normal));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_unit_vectors(uint64_t seed, uint32_t n, double* out) {  // random_unit_vector draws
    for (uint32_t i = 0; i < n; i++) {
        RngStream s(seed, DOMAIN_MATERIAL, i, 0, 0);
        Vec3 v = random_unit_vector(s);
        out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
    }
}
void orc_kat_colour_to_raw(const double* c, uint8_t* out) { colour_to_raw(Vec3(c), out); }
void orc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
void orc_uniforms(uint64_t seed, uint32_t domain, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t n, double* out) {
    RngStream s(seed, domain, pixel, sample, bounce);
    for (uint32_t i = 0; i < n; i++) out[i] = s.next();
}
}
