// Output image encoders, chosen by file extension like image 0.23's ImageBuffer::save
// (reference src/main.rs:161-162): .png (zlib deflate), .ppm / .pnm (binary P6), .bmp (24-bit).
// Other formats the image crate knows (.jpg, .gif, .tiff, ...) are not encoded here: the call
// fails with a clear message instead of writing something else under that name.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

namespace {

void write_all(const std::string& path, const std::vector<uint8_t>& bytes) {
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path + ": " + std::strerror(errno));
    bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    ok = std::fclose(f) == 0 && ok;
    if (!ok) throw std::runtime_error("write error on " + path);
}

void be32(std::vector<uint8_t>& out, uint32_t v) {
    out.push_back((uint8_t)(v >> 24)); out.push_back((uint8_t)(v >> 16)); out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v);
}

void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

std::vector<uint8_t> encode_png(uint32_t w, uint32_t h, const uint8_t* rgb) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (w * 3 + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);  // filter type None
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("PNG: deflate failed");
    z.resize(bound);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, w);
    be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8-bit, truecolour RGB
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    return out;
}

std::vector<uint8_t> encode_ppm(uint32_t w, uint32_t h, const uint8_t* rgb) {
    std::string head = "P6\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    out.insert(out.end(), rgb, rgb + (size_t)w * h * 3);
    return out;
}

std::vector<uint8_t> encode_bmp(uint32_t w, uint32_t h, const uint8_t* rgb) {
    uint32_t stride = (w * 3 + 3) & ~3u, size = 54 + stride * h;
    std::vector<uint8_t> out(size, 0);
    auto le32 = [&](size_t at, uint32_t v) { out[at] = (uint8_t)v; out[at + 1] = (uint8_t)(v >> 8); out[at + 2] = (uint8_t)(v >> 16); out[at + 3] = (uint8_t)(v >> 24); };
    out[0] = 'B'; out[1] = 'M';
    le32(2, size); le32(10, 54); le32(14, 40); le32(18, w); le32(22, h);
    out[26] = 1; out[28] = 24;
    le32(34, stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* src = rgb + (size_t)(h - 1 - y) * w * 3;  // BMP rows run bottom-up
        uint8_t* dst = &out[54 + (size_t)y * stride];
        for (uint32_t x = 0; x < w; x++) { dst[3 * x] = src[3 * x + 2]; dst[3 * x + 1] = src[3 * x + 1]; dst[3 * x + 2] = src[3 * x]; }
    }
    return out;
}

}  // namespace

// rgb: width*height*3 bytes, top row first.
void save_image(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (ext == "png") write_all(path, encode_png(width, height, rgb));
    else if (ext == "ppm" || ext == "pnm") write_all(path, encode_ppm(width, height, rgb));
    else if (ext == "bmp") write_all(path, encode_bmp(width, height, rgb));
    else
        throw std::runtime_error("The image format could not be determined or is not supported by this build (." + ext +
                                 "): use .png, .ppm or .bmp");
}

}  // namespace rt
