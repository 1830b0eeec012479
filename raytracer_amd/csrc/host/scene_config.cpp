// Scene file loader: see scene_config.hpp.  Follows reference src/config.rs:19-421 for
// what is accepted, which names must resolve, and the error texts.
#include "scene_config.hpp"

#include <cmath>
#include <fstream>
#include <set>
#include <sstream>

#include "json5.hpp"

namespace rt {

uint32_t Scene::image_height(uint32_t width) const {
    // f64::round rounds half away from zero, as std::round does.
    return (uint32_t)std::round((double)width / aspect());
}

namespace {

std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string join_path(const std::string& root, const std::string& rel) {
    if (root.empty() || (!rel.empty() && rel[0] == '/')) return rel;
    return root.back() == '/' ? root + rel : root + "/" + rel;
}

// Field access with serde's deny_unknown_fields + "missing field" behaviour.
class Fields {
public:
    Fields(const JValue& v, const std::string& what) : v_(v), what_(what) {
        if (!v.is(JValue::Object))
            throw SceneError(what + ": expected an object, found " + JValue::kind_name(v.kind) +
                             " (line " + std::to_string(v.line) + ")");
    }
    const JValue& get(const char* key) {
        used_.insert(key);
        const JValue* m = v_.find(key);
        if (!m) throw SceneError(what_ + ": missing field `" + key + "`");
        return *m;
    }
    const JValue* opt(const char* key) {
        used_.insert(key);
        const JValue* m = v_.find(key);
        if (m && m->is(JValue::Null)) return nullptr;
        return m;
    }
    double number(const char* key) { return as_number(get(key), key); }
    bool boolean(const char* key) {
        const JValue& m = get(key);
        if (!m.is(JValue::Bool)) bad(key, "a boolean", m);
        return m.b;
    }
    std::string string(const char* key) {
        const JValue& m = get(key);
        if (!m.is(JValue::String)) bad(key, "a string", m);
        return m.str;
    }
    void vec(const char* key, double* out, size_t n) {
        const JValue& m = get(key);
        if (!m.is(JValue::Array) || m.arr.size() != n)
            bad(key, ("an array of " + std::to_string(n) + " numbers").c_str(), m);
        for (size_t i = 0; i < n; i++) out[i] = as_number(m.arr[i], key);
    }
    void done() {
        for (auto& kv : v_.obj)
            if (!used_.count(kv.first))
                throw SceneError(what_ + ": unknown field `" + kv.first + "`");
    }
    const std::string& what() const { return what_; }

private:
    double as_number(const JValue& m, const char* key) {
        if (!m.is(JValue::Number)) bad(key, "a number", m);
        return m.num;
    }
    [[noreturn]] void bad(const char* key, const char* expected, const JValue& m) {
        throw SceneError(what_ + ": field `" + key + "` must be " + expected + ", found " +
                         JValue::kind_name(m.kind) + " (line " + std::to_string(m.line) + ")");
    }
    const JValue& v_;
    std::string what_;
    std::set<std::string> used_;
};

template <class T>
int index_of(const std::vector<T>& list, const std::string& name) {
    for (size_t i = 0; i < list.size(); i++)
        if (list[i].name == name) return (int)i;
    return -1;
}

struct PendingObject {
    std::string material, prototype;
};

}  // namespace

Scene load_scene_text(const std::string& text, const std::string& asset_root) {
    JValue root_v = json5_parse(text);
    Scene sc;
    Fields root(root_v, "scene");

    {  // camera (src/config.rs:447-461)
        Fields f(root.get("camera"), "camera");
        f.vec("lookFrom", sc.camera.look_from, 3);
        f.vec("lookAt", sc.camera.look_at, 3);
        f.vec("directionUp", sc.camera.direction_up, 3);
        sc.camera.vertical_fov = f.number("fieldOfView");
        f.vec("aspectRatio", sc.camera.aspect_ratio, 2);
        sc.camera.aperture = f.number("aperture");
        sc.camera.focus_dist = f.number("distanceToFocus");
        sc.camera.start_time = f.number("startTime");
        sc.camera.end_time = f.number("endTime");
        f.done();
        // Camera::new panics otherwise (src/camera.rs:36-38).
        if (sc.camera.start_time < 0.0 || sc.camera.end_time > 1.0 ||
            sc.camera.start_time > sc.camera.end_time)
            throw SceneError("Camera must have 0 <= start_time <= end_time <= 1");
    }
    {  // background (src/config.rs:434-445)
        Fields f(root.get("background"), "background");
        std::string type = f.string("type");
        if (type == "plainColour") {
            sc.background.type = BackgroundDef::PlainColour;
            f.vec("colour", sc.background.colour, 3);
        } else if (type == "gradient") {
            sc.background.type = BackgroundDef::Gradient;
            f.vec("direction", sc.background.direction, 3);
            f.vec("colour0", sc.background.colour0, 3);
            f.vec("colour1", sc.background.colour1, 3);
        } else {
            throw SceneError("background: unknown variant `" + type +
                             "`, expected `plainColour` or `gradient`");
        }
        f.done();
    }
    {  // textures (src/config.rs:65-96, 463-470)
        const JValue& tv = root.get("textures");
        Fields(tv, "textures");
        for (auto& kv : tv.obj) {
            Fields f(kv.second, "texture " + kv.first);
            TextureDef t;
            t.name = kv.first;
            std::string type = f.string("type");
            if (type == "solidColour") {
                t.type = TextureDef::SolidColour;
                f.vec("colour", t.colour, 3);
            } else if (type == "imageTexture") {
                t.type = TextureDef::ImageTexture;
                t.filename = f.string("filename");
                // De-duplicate decoded images by file name.
                for (size_t k = 0; k < sc.textures.size(); k++)
                    if (sc.textures[k].type == TextureDef::ImageTexture &&
                        sc.textures[k].filename == t.filename)
                        t.image = sc.textures[k].image;
                if (t.image < 0) {
                    sc.images.push_back(load_image_rgb8(join_path(asset_root, t.filename)));
                    t.image = (int)sc.images.size() - 1;
                }
            } else {
                throw SceneError("texture " + kv.first + ": unknown variant `" + type +
                                 "`, expected `solidColour` or `imageTexture`");
            }
            f.done();
            sc.textures.push_back(std::move(t));
        }
    }
    {  // materials (src/config.rs:98-174, 472-491)
        const JValue& mv = root.get("materials");
        Fields(mv, "materials");
        std::vector<std::pair<std::string, std::string>> checker_names;
        for (auto& kv : mv.obj) {
            Fields f(kv.second, "material " + kv.first);
            MaterialDef m;
            m.name = kv.first;
            std::string type = f.string("type");
            std::string tex, odd, even;
            if (type == "lambertian") {
                m.type = MaterialDef::Lambertian;
                tex = f.string("texture");
            } else if (type == "metal") {
                m.type = MaterialDef::Metal;
                m.fuzz = f.number("fuzz");
                f.vec("albedo", m.albedo, 3);
            } else if (type == "dielectric") {
                m.type = MaterialDef::Dielectric;
                m.index_of_refraction = f.number("indexOfRefraction");
            } else if (type == "diffuseLight") {
                m.type = MaterialDef::DiffuseLight;
                tex = f.string("emit");
            } else if (type == "isotropic") {
                m.type = MaterialDef::Isotropic;
                tex = f.string("albedo");
            } else if (type == "checkered") {
                m.type = MaterialDef::Checkered;
                odd = f.string("odd");
                even = f.string("even");
                m.tile_density = f.number("tileDensity");
            } else {
                throw SceneError("material " + kv.first + ": unknown variant `" + type + "`");
            }
            f.done();
            if (m.type == MaterialDef::Lambertian || m.type == MaterialDef::DiffuseLight ||
                m.type == MaterialDef::Isotropic) {
                m.texture = index_of(sc.textures, tex);
                if (m.texture < 0) throw SceneError("Texture " + tex + " does not exist");
            }
            checker_names.emplace_back(odd, even);
            sc.materials.push_back(std::move(m));
        }
        // Checkered materials wait for their odd/even parts (fixpoint, src/config.rs:143-159);
        // a part that never appears makes the material "impossible to construct".
        std::vector<char> built(sc.materials.size(), 0);
        size_t remaining = sc.materials.size();
        while (remaining) {
            bool progress = false;
            for (size_t i = 0; i < sc.materials.size(); i++) {
                if (built[i]) continue;
                MaterialDef& m = sc.materials[i];
                if (m.type == MaterialDef::Checkered) {
                    int o = index_of(sc.materials, checker_names[i].first);
                    int e = index_of(sc.materials, checker_names[i].second);
                    if (o < 0 || e < 0 || !built[o] || !built[e]) continue;
                    m.odd = o;
                    m.even = e;
                }
                built[i] = 1;
                sc.material_order.push_back((int)i);
                remaining--;
                progress = true;
            }
            if (!progress) {
                for (size_t i = 0; i < sc.materials.size(); i++)
                    if (!built[i])
                        throw SceneError("Material " + sc.materials[i].name +
                                         " is impossible to construct");
            }
        }
    }
    {  // objects (src/config.rs:176-403, 493-569)
        const JValue& ov = root.get("objects");
        Fields(ov, "objects");
        std::vector<PendingObject> pending;
        for (auto& kv : ov.obj) {
            Fields f(kv.second, "object " + kv.first);
            ObjectDef o;
            PendingObject p;
            o.name = kv.first;
            std::string type = f.string("type");
            if (type == "sphere") {
                o.type = ObjectDef::Sphere;
                f.vec("centre", o.p0, 3);
                o.radius = f.number("radius");
                p.material = f.string("material");
            } else if (type == "movingSphere") {
                o.type = ObjectDef::MovingSphere;
                f.vec("centre0", o.p0, 3);
                f.vec("centre1", o.p1, 3);
                o.time0 = f.number("time0");
                o.time1 = f.number("time1");
                o.radius = f.number("radius");
                p.material = f.string("material");
            } else if (type == "block") {
                o.type = ObjectDef::Block;
                f.vec("corner0", o.p0, 3);
                f.vec("corner1", o.p1, 3);
                p.material = f.string("material");
            } else if (type == "rect") {
                o.type = ObjectDef::Rect;
                f.vec("corner0", o.p0, 3);
                f.vec("corner1", o.p1, 3);
                o.facing_forward = f.boolean("facingForward");
                p.material = f.string("material");
            } else if (type == "triangle") {
                o.type = ObjectDef::Triangle;
                f.vec("point0", o.p0, 3);
                f.vec("point1", o.p1, 3);
                f.vec("point2", o.p2, 3);
                p.material = f.string("material");
            } else if (type == "plane") {
                o.type = ObjectDef::Plane;
                f.vec("point0", o.p0, 3);
                f.vec("point1", o.p1, 3);
                f.vec("point2", o.p2, 3);
                o.uv_repeat = f.number("uvRepeat");
                p.material = f.string("material");
            } else if (type == "mesh") {
                o.type = ObjectDef::Mesh;
                std::string filename = f.string("filename");
                std::string object_name;  // default "" (src/config.rs:323)
                if (const JValue* n = f.opt("objectName")) {
                    if (!n->is(JValue::String))
                        throw SceneError("object " + kv.first + ": field `objectName` must be a string");
                    object_name = n->str;
                }
                p.material = f.string("material");
                for (size_t k = 0; k < sc.meshes.size(); k++)
                    if (sc.meshes[k].filename == filename && sc.meshes[k].object_name == object_name)
                        o.mesh = (int)k;
                if (o.mesh < 0) {
                    MeshData md = load_obj_mesh(join_path(asset_root, filename), object_name);
                    md.filename = filename;
                    sc.meshes.push_back(std::move(md));
                    o.mesh = (int)sc.meshes.size() - 1;
                }
            } else if (type == "spotlight") {
                o.type = ObjectDef::Spotlight;
                f.vec("lookFrom", o.p0, 3);
                f.vec("lookAt", o.p1, 3);
                o.length = f.number("length");
                o.width = f.number("width");
                f.vec("light", o.p2, 3);
            } else if (type == "constantMedium") {
                o.type = ObjectDef::ConstantMedium;
                p.prototype = f.string("boundary");
                p.material = f.string("phaseFunction");
                o.density = f.number("density");
            } else if (type == "translate") {
                o.type = ObjectDef::Translate;
                p.prototype = f.string("prototype");
                f.vec("offset", o.p0, 3);
            } else if (type == "rotateX" || type == "rotateY" || type == "rotateZ") {
                o.type = type == "rotateX" ? ObjectDef::RotateX
                         : type == "rotateY" ? ObjectDef::RotateY : ObjectDef::RotateZ;
                p.prototype = f.string("prototype");
                o.degrees = f.number("degrees");
            } else {
                throw SceneError("object " + kv.first + ": unknown variant `" + type + "`");
            }
            f.done();
            if (o.type == ObjectDef::Rect) {
                // Axis rule of src/config.rs:247-279: x tested first, then y, then z.
                if (o.p0[0] == o.p1[0]) o.rect_axis = 0;
                else if (o.p0[1] == o.p1[1]) o.rect_axis = 1;
                else if (o.p0[2] == o.p1[2]) o.rect_axis = 2;
                else
                    throw SceneError(
                        "Rectangles are 2d; corner0 and corner1 must be equal along one axis");
            }
            sc.objects.push_back(std::move(o));
            pending.push_back(std::move(p));
        }
        // Dependency fixpoint (src/config.rs:186-401): an object whose prototype/boundary is
        // not built yet goes back into the queue; no progress => impossible to construct.
        std::vector<char> built(sc.objects.size(), 0);
        size_t remaining = sc.objects.size();
        while (remaining) {
            bool progress = false;
            for (size_t i = 0; i < sc.objects.size(); i++) {
                if (built[i]) continue;
                ObjectDef& o = sc.objects[i];
                const PendingObject& p = pending[i];
                bool needs_proto = o.type == ObjectDef::ConstantMedium ||
                                   o.type == ObjectDef::Translate || o.type == ObjectDef::RotateX ||
                                   o.type == ObjectDef::RotateY || o.type == ObjectDef::RotateZ;
                if (needs_proto) {
                    int proto = index_of(sc.objects, p.prototype);
                    if (proto < 0 || !built[proto]) continue;
                    o.prototype = proto;
                }
                bool has_material = o.type == ObjectDef::ConstantMedium ||
                                    (o.type != ObjectDef::Spotlight && !needs_proto);
                if (has_material) {
                    o.material = index_of(sc.materials, p.material);
                    if (o.material < 0)
                        throw SceneError("Material " + p.material + " does not exist");
                }
                built[i] = 1;
                sc.object_order.push_back((int)i);
                remaining--;
                progress = true;
            }
            if (!progress) {
                for (size_t i = 0; i < sc.objects.size(); i++)
                    if (!built[i])
                        throw SceneError("Object " + sc.objects[i].name +
                                         " is impossible to construct");
            }
        }
    }
    {  // world (src/config.rs:26-35)
        const JValue& wv = root.get("world");
        if (!wv.is(JValue::Array)) throw SceneError("scene: field `world` must be an array");
        for (auto& e : wv.arr) {
            if (!e.is(JValue::String)) throw SceneError("scene: `world` entries must be strings");
            int idx = index_of(sc.objects, e.str);
            if (idx < 0) throw SceneError("Object " + e.str + " does not exist");
            sc.world.push_back(idx);
        }
        // BVHNode::from_vec panics on an empty world (src/hitting.rs:164).
        if (sc.world.empty()) throw SceneError("BVHNode cannot be created from empty slice");
    }
    root.done();
    return sc;
}

Scene load_scene_file(const std::string& path, const std::string& asset_root) {
    return load_scene_text(read_file(path), asset_root);
}

}  // namespace rt
