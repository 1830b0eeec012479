// Wavefront .obj reader for `mesh` objects.
//
// Stands in for wavefront_obj 9.0.0 (Cargo.lock:828-829) as it is used by load_mesh at
// reference src/objects.rs:719-762: objects are delimited by `o <name>` lines (a file
// without any `o` line yields one object named ""), the object is chosen by exact name,
// polygons are fan-triangulated, points and lines are ignored, and a triangle's normal is
// the `vn` of the first of its corners that carries one (src/objects.rs:742).
// The crate is not vendored under /root/reference, so this follows its published
// behaviour; parity for it is unpinned (no reference test covers meshes).
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "scene_config.hpp"

namespace rt {

namespace {

struct ObjObject {
    std::string name;
    size_t v_min = 0, vn_min = 0;  // global index of this object's first v / vn
    size_t v_count = 0, vn_count = 0;
    std::vector<MeshTriangle> tris;
};

struct Corner {
    long v = 0;
    bool has_n = false;
    long n = 0;
};

}  // namespace

MeshData parse_obj_mesh(const std::string& text, const std::string& object_name,
                        const std::string& path) {
    std::vector<double> v, vn;  // global pools, 3 doubles each
    std::vector<ObjObject> objects;
    auto current = [&]() -> ObjObject& {
        if (objects.empty()) objects.emplace_back();  // implicit object named ""
        return objects.back();
    };
    std::istringstream in(text);
    std::string line, tok;
    std::vector<Corner> cs;
    int lineno = 0;
    auto fail = [&](const std::string& msg) {
        throw SceneError(path + ":" + std::to_string(lineno) + ": " + msg);
    };
    while (std::getline(in, line)) {
        lineno++;
        // Continuation lines and comments.
        while (!line.empty() && line.back() == '\r') line.pop_back();
        while (!line.empty() && line.back() == '\\') {
            line.pop_back();
            std::string more;
            if (!std::getline(in, more)) break;
            lineno++;
            line += " " + more;
        }
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        // (scanned in place: a stream per line costs 2 s on a million-triangle file)
        const char* p = line.c_str();
        auto skip_space = [&] { while (*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r') p++; };
        auto token_end = [](const char* q) { while (*q && *q != ' ' && *q != '\t' && *q != '\v' && *q != '\f' && *q != '\r') q++; return q; };
        skip_space();
        const char* tag_end = token_end(p);
        if (tag_end == p) continue;
        const size_t tag_len = (size_t)(tag_end - p);
        auto tag_is = [&](const char* t) { return std::strlen(t) == tag_len && std::memcmp(p, t, tag_len) == 0; };
        const bool is_o = tag_is("o"), is_v = tag_is("v"), is_vn = tag_is("vn"), is_f = tag_is("f");
        p = tag_end;
        auto three_numbers = [&](double* out) {
            for (int k = 0; k < 3; k++) {
                char* end = nullptr;
                out[k] = std::strtod(p, &end);
                if (end == p) return false;
                p = end;
            }
            return true;
        };
        if (is_o) {
            ObjObject o;
            std::string rest(p);
            size_t a = rest.find_first_not_of(" \t");
            size_t b = rest.find_last_not_of(" \t");
            o.name = a == std::string::npos ? "" : rest.substr(a, b - a + 1);
            o.v_min = v.size() / 3;
            o.vn_min = vn.size() / 3;
            objects.push_back(std::move(o));
        } else if (is_v) {
            double xyz[3];
            if (!three_numbers(xyz)) fail("bad vertex");
            v.insert(v.end(), xyz, xyz + 3);
            current().v_count++;
        } else if (is_vn) {
            double xyz[3];
            if (!three_numbers(xyz)) fail("bad normal");
            vn.insert(vn.end(), xyz, xyz + 3);
            current().vn_count++;
        } else if (is_f) {
            ObjObject& o = current();
            cs.clear();
            for (;;) {
                skip_space();
                const char* q = token_end(p);
                if (q == p) break;
                tok.assign(p, (size_t)(q - p));  // (a copy, so that a number never runs on into the next token)
                p = q;
                Corner c;
                // v, v/vt, v//vn, v/vt/vn
                const char* tp = tok.c_str();
                char* end = nullptr;
                c.v = std::strtol(tp, &end, 10);
                if (end == tp) fail("bad face index `" + tok + "`");
                if (*end == '/') {
                    tp = end + 1;
                    if (*tp != '/') {
                        (void)std::strtol(tp, &end, 10);  // vt, unused
                    } else {
                        end = const_cast<char*>(tp);
                    }
                    if (*end == '/') {
                        tp = end + 1;
                        long n = std::strtol(tp, &end, 10);
                        if (end != tp) {
                            c.has_n = true;
                            c.n = n;
                        }
                    }
                }
                cs.push_back(c);
            }
            if (cs.size() < 3) continue;  // points / lines written as faces are not triangles
            auto vertex = [&](long idx, double* out) {
                long g = idx > 0 ? idx - 1 : (long)(v.size() / 3) + idx;
                if (idx == 0 || g < (long)o.v_min || g >= (long)(o.v_min + o.v_count))
                    fail("face vertex index " + std::to_string(idx) + " outside its object");
                std::memcpy(out, &v[(size_t)g * 3], 3 * sizeof(double));
            };
            auto normal = [&](long idx, double* out) {
                long g = idx > 0 ? idx - 1 : (long)(vn.size() / 3) + idx;
                if (idx == 0 || g < (long)o.vn_min || g >= (long)(o.vn_min + o.vn_count))
                    fail("face normal index " + std::to_string(idx) + " outside its object");
                std::memcpy(out, &vn[(size_t)g * 3], 3 * sizeof(double));
            };
            for (size_t k = 1; k + 1 < cs.size(); k++) {
                const Corner* tri[3] = {&cs[0], &cs[k], &cs[k + 1]};
                MeshTriangle t;
                for (int c = 0; c < 3; c++) vertex(tri[c]->v, t.p[c]);
                for (int c = 0; c < 3; c++) {
                    if (tri[c]->has_n) {
                        t.has_normal = true;
                        normal(tri[c]->n, t.normal);
                        break;
                    }
                }
                o.tris.push_back(t);
            }
        }
        // vt, g, s, usemtl, mtllib, l, p, ...: irrelevant to load_mesh.
    }
    if (objects.empty()) objects.emplace_back();
    for (auto& o : objects) {
        if (o.name == object_name) {
            MeshData md;
            md.filename = path;
            md.object_name = object_name;
            md.triangles = std::move(o.tris);
            if (md.triangles.empty())  // BVHNode::from_vec panics on an empty list (src/hitting.rs:164)
                throw SceneError("BVHNode cannot be created from empty slice (mesh " + path + ")");
            return md;
        }
    }
    std::string names = "[";
    for (size_t i = 0; i < objects.size(); i++)
        names += (i ? ", \"" : "\"") + objects[i].name + "\"";
    names += "]";
    // Error text of src/objects.rs:730-735.
    throw SceneError("No object with name " + object_name + " in file " + path +
                     "\nObject names are " + names);
}

MeshData load_obj_mesh(const std::string& path, const std::string& object_name) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError("cannot open mesh file " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_obj_mesh(ss.str(), object_name, path);
}

}  // namespace rt
