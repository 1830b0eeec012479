// Flattener: resolved scene description -> the SoA arrays of rt_scene_desc.
//
// Replaces the object-graph construction of config::load_config (reference
// src/config.rs:19-63, build_hittables :176-403) and BVHNode::from_vec
// (src/hitting.rs:147-197) with a linearised two-level BVH plus instance records.
#pragma once
#include <vector>

#include "../../../include/rt_mi355x.h"
#include "scene_config.hpp"

namespace rt {

struct FlatScene {
    std::vector<rt_node> nodes;
    std::vector<rt_object> objects;
    std::vector<rt_xform> xforms;
    std::vector<rt_sphere> spheres;
    std::vector<rt_moving_sphere> moving_spheres;
    std::vector<rt_rect> rects;
    std::vector<rt_triangle> triangles;
    std::vector<rt_plane> planes;
    std::vector<rt_medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels;
    std::vector<int32_t> top_list;
    int32_t tlas_root = -1;
    int32_t stack_depth = 0;
    rt_camera camera;
    rt_sky sky;
    rt_scene_desc desc() const;
};

// Throws SceneError for constructs the GPU path does not take (a constant medium whose
// boundary is itself a constant medium).
FlatScene flatten_scene(const Scene& scene);

}  // namespace rt
