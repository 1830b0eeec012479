// Scene description: the JSON5 scene format of the reference, parsed and name-resolved.
//
// Mirrors the serde types at reference src/config.rs:423-569 (field names, camelCase
// type tags, deny_unknown_fields) and the name-resolution / dependency rules of
// build_textures / build_materials / build_hittables (src/config.rs:65-403).  This is
// boundary code (the scene-file contract); it is consumed both by the flattener that
// feeds the GPU and by the CPU oracle under oracle/.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

struct SceneError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct CameraDef {  // src/config.rs:447-461
    double look_from[3], look_at[3], direction_up[3];
    double vertical_fov;  // "fieldOfView", degrees
    double aspect_ratio[2];
    double aperture;
    double focus_dist;  // "distanceToFocus"
    double start_time, end_time;
};

struct BackgroundDef {  // src/config.rs:434-445
    enum Type { PlainColour, Gradient } type;
    double colour[3];                            // PlainColour
    double direction[3], colour0[3], colour1[3];  // Gradient
};

struct ImageRGB8 {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;  // row-major, top row first, 3 bytes per texel
};

struct TextureDef {  // src/config.rs:463-470
    enum Type { SolidColour, ImageTexture } type;
    std::string name;
    double colour[3] = {0, 0, 0};
    std::string filename;
    int image = -1;  // index into Scene::images
};

struct MaterialDef {  // src/config.rs:472-491
    enum Type { Lambertian, Metal, Dielectric, DiffuseLight, Isotropic, Checkered } type;
    std::string name;
    int texture = -1;  // lambertian.texture / diffuseLight.emit / isotropic.albedo
    double albedo[3] = {0, 0, 0};  // metal
    double fuzz = 0;               // metal
    double index_of_refraction = 0;
    int odd = -1, even = -1;  // checkered
    double tile_density = 0;
};

struct MeshTriangle {
    double p[3][3];
    bool has_normal = false;  // first corner that carries a vn index (src/objects.rs:742)
    double normal[3] = {0, 0, 0};
};

struct MeshData {
    std::string filename, object_name;
    std::vector<MeshTriangle> triangles;
};

struct ObjectDef {  // src/config.rs:493-569
    enum Type {
        Sphere, MovingSphere, Block, Rect, Triangle, Plane, Mesh, Spotlight,
        ConstantMedium, Translate, RotateX, RotateY, RotateZ
    } type;
    std::string name;
    // p0: centre / centre0 / corner0 / point0 / lookFrom / offset
    // p1: centre1 / corner1 / point1 / lookAt
    // p2: point2 / light (spotlight colour)
    double p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
    double radius = 0, time0 = 0, time1 = 0;
    double uv_repeat = 0, length = 0, width = 0, density = 0, degrees = 0;
    bool facing_forward = false;
    int rect_axis = -1;  // Rect only: 0 = YZ rect (x fixed), 1 = XZ, 2 = XY (src/config.rs:247-279)
    int material = -1;   // also constantMedium.phaseFunction
    int prototype = -1;  // translate/rotate prototype, constantMedium.boundary
    int mesh = -1;       // index into Scene::meshes
};

struct Scene {
    CameraDef camera;
    BackgroundDef background;
    std::vector<ImageRGB8> images;
    std::vector<TextureDef> textures;
    std::vector<MaterialDef> materials;
    std::vector<ObjectDef> objects;
    std::vector<MeshData> meshes;
    std::vector<int> world;           // object indices, in `world` order
    std::vector<int> material_order;  // dependencies-first construction order
    std::vector<int> object_order;
    double aspect() const { return camera.aspect_ratio[0] / camera.aspect_ratio[1]; }
    // src/main.rs:82: (width / aspect).round(), round half away from zero.
    uint32_t image_height(uint32_t width) const;
};

// Reads and resolves a scene file.  `asset_root` is prepended to relative `filename`s in
// the scene (the reference opens them relative to the process CWD, src/objects.rs:724,
// src/textures.rs:33 - pass "" for that behaviour).  Throws SceneError / Json5Error.
Scene load_scene_file(const std::string& path, const std::string& asset_root = "");
Scene load_scene_text(const std::string& json5_text, const std::string& asset_root = "");

// .obj reader standing in for wavefront_obj 9.0.0 as called at src/objects.rs:719-762.
MeshData load_obj_mesh(const std::string& path, const std::string& object_name);
MeshData parse_obj_mesh(const std::string& text, const std::string& object_name,
                        const std::string& path_for_errors);

// Baseline JPEG / PPM(P6) / PNG-less image reader standing in for image 0.23.12
// (src/textures.rs:32-36): decodes to RGB8.
ImageRGB8 load_image_rgb8(const std::string& path);
ImageRGB8 decode_jpeg(const uint8_t* data, size_t size);           // baseline, extended and progressive Huffman JPEG
ImageRGB8 decode_jpeg_baseline(const uint8_t* data, size_t size);  // (the name earlier rounds used: the same decoder)
ImageRGB8 decode_png(const uint8_t* data, size_t size);

}  // namespace rt
