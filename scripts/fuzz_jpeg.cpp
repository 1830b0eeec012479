// Host-side sanitizer run of the baseline JPEG decoder (image textures are untrusted input):
//   g++ -std=c++17 -g -fsanitize=address,undefined -Iraytracer_amd/csrc/host scripts/fuzz_jpeg.cpp \
//       raytracer_amd/csrc/host/image_io.cpp -o /tmp/fuzz_jpeg && /tmp/fuzz_jpeg examples/globe.jpg
// Every truncation of the first 1 KiB and 2000 random corruptions (use a small picture: every case decodes all of its blocks) must end in a picture or a SceneError.
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <vector>

#include "scene_config.hpp"

int main(int argc, char** argv) {
    std::ifstream f(argc > 1 ? argv[1] : "examples/globe.jpg", std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string good = ss.str();
    if (good.size() < 100) return 2;
    size_t errors = 0, pictures = 0;
    auto run = [&](const std::vector<uint8_t>& d) {
        try {
            rt::decode_jpeg_baseline(d.data(), d.size());
            pictures++;
        } catch (const rt::SceneError&) {
            errors++;
        }
    };
    for (size_t cut = 0; cut < 1024 && cut < good.size(); cut++) run(std::vector<uint8_t>(good.begin(), good.begin() + (long)cut));
    std::mt19937 rng(7);
    const size_t header = 700;  // tables and headers of globe.jpg end before this offset
    std::vector<uint8_t> small(good.begin(), good.begin() + (long)std::min<size_t>(good.size(), 6000));
    for (int k = 0; k < 2000; k++) {
        std::vector<uint8_t> d = small;
        const int n = 1 + (int)(rng() % 3);
        for (int j = 0; j < n; j++) d[rng() % header] = (uint8_t)rng();
        run(d);
    }
    std::printf("%zu errors, %zu pictures, no fault\n", errors, pictures);
    return 0;
}
