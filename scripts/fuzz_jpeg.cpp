// Host-side sanitizer run of the texture decoders (image textures are untrusted input):
//   g++ -std=c++17 -g -fsanitize=address,undefined -Iraytracer_amd/csrc/host scripts/fuzz_jpeg.cpp \
//       raytracer_amd/csrc/host/image_io.cpp -lz -o /tmp/fuzz_image && /tmp/fuzz_image small.jpg small_progressive.jpg small.png
// For every file (JPEG or PNG, by its signature; use small pictures - every case decodes the whole of it): every truncation of
// the first 1 KiB and 3000 random corruptions, anywhere in the file and concentrated in its headers, must end in a picture
// or a SceneError.
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <vector>

#include "scene_config.hpp"

int main(int argc, char** argv) {
    size_t errors = 0, pictures = 0;
    for (int a = 1; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string good = ss.str();
        if (good.size() < 100) return 2;
        const bool png = (unsigned char)good[0] == 0x89;
        auto run = [&](const std::vector<uint8_t>& d) {
            try {
                if (png) rt::decode_png(d.data(), d.size());
                else rt::decode_jpeg(d.data(), d.size());
                pictures++;
            } catch (const rt::SceneError&) {
                errors++;
            }
        };
        for (size_t cut = 0; cut < 1024 && cut < good.size(); cut++) run(std::vector<uint8_t>(good.begin(), good.begin() + (long)cut));
        std::mt19937 rng(7);
        const std::vector<uint8_t> whole(good.begin(), good.end());
        const size_t header = std::min<size_t>(good.size(), 700);
        for (int k = 0; k < 3000; k++) {
            std::vector<uint8_t> d = whole;
            const int n = 1 + (int)(rng() % 3);
            for (int j = 0; j < n; j++) d[rng() % (k % 2 ? header : d.size())] = (uint8_t)rng();
            run(d);
        }
    }
    std::printf("%zu errors, %zu pictures, no fault\n", errors, pictures);
    return 0;
}
