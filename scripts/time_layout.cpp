// Times the host-side traversal-layout build of rt_upload_scene (tree_layout.hpp) without a GPU:
//   hipcc -O2 -std=c++17 -Iinclude scripts/time_layout.cpp -Lraytracer_amd/lib -lrt_mi355x -Wl,-rpath,$PWD/raytracer_amd/lib -o /tmp/time_layout
//   /tmp/time_layout <scene.json5> <asset root>
#include <chrono>
#include <cstdio>
#include <vector>

#include "../include/rt_mi355x.h"
#include "../raytracer_amd/csrc/device/rt_device.h"
#include "../raytracer_amd/csrc/tree_layout.hpp"

int main(int argc, char** argv) {
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a) { return std::chrono::duration<double, std::milli>(clock::now() - a).count(); };
    rt_scene* sc = nullptr;
    auto t0 = clock::now();
    if (rt_scene_load_file(argv[1], argc > 2 ? argv[2] : "", &sc) != RT_OK) {
        std::printf("%s\n", rt_last_global_error());
        return 1;
    }
    std::printf("load + flatten %.1f ms\n", ms(t0));
    const rt_scene_desc& d = *rt_scene_get_desc(sc);
    using namespace rt_layout;
    const float pad = box_pad(d);
    for (int rep = 0; rep < 2; rep++) {
        t0 = clock::now();
        std::vector<rtd::DNode2> v2;
        LeafRanks ranks;
        ranks.object.assign(d.n_objects, 0);
        ranks.triangle.assign(d.n_triangles, 0);
        if (d.tlas_root >= 0) build_nodes2(d, d.tlas_root, pad, v2, ranks);
        for (uint64_t i = 0; i < d.n_objects; i++)
            if (d.objects[i].kind == RT_OBJ_BLAS) build_nodes2(d, d.objects[i].first, pad, v2, ranks);
        const double t2 = ms(t0);
        t0 = clock::now();
        std::vector<rtd::WideNode4> v4;
        int need = 0;
        int32_t r = -1;
        for (uint64_t i = 0; i < d.n_objects; i++)
            if (d.objects[i].kind == RT_OBJ_BLAS) build_sah4(d, d.objects[i].first, pad, v4, &r, &need);
        const double t4 = ms(t0);
        t0 = clock::now();
        std::vector<rtd::DNode4> packed;
        quantize_nodes4(v4, packed);
        std::printf("nodes2 %.1f ms (%zu), sah4 %.1f ms (%zu nodes, stack need %d), quantise %.1f ms\n", t2, v2.size(), t4, v4.size(), need, ms(t0));
    }
    return 0;
}
