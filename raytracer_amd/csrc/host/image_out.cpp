// Output image encoders, chosen by file extension like image 0.23's ImageBuffer::save
// (reference src/main.rs:161-162): .png (zlib deflate), .jpg / .jpeg (baseline JFIF, quality 75 - the
// image crate's default; its covers are .jpg), .ppm / .pnm (binary P6), .pam (P7), .bmp (24-bit), .tga (type 2),
// .tiff / .tif (one uncompressed strip), .ff / .farbfeld.  The formats the image crate knows that need a palette
// or a container of sizes (.gif, .ico) and the grey / bitmap PNM kinds (.pgm, .pbm) are not encoded here: the call
// fails with a clear message instead of writing something else under that name.
#include <zlib.h>

#include <algorithm>
#include <cmath>
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

// Truevision TGA, type 2 (uncompressed true colour), 24 bits, top-left origin, BGR.
std::vector<uint8_t> encode_tga(uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (w > 65535 || h > 65535) throw std::runtime_error("a .tga image holds at most 65535 x 65535 pixels");
    std::vector<uint8_t> out(18 + (size_t)w * h * 3, 0);
    out[2] = 2;
    out[12] = (uint8_t)w; out[13] = (uint8_t)(w >> 8);
    out[14] = (uint8_t)h; out[15] = (uint8_t)(h >> 8);
    out[16] = 24;
    out[17] = 0x20;  // rows run top-down
    for (size_t i = 0; i < (size_t)w * h; i++) {
        out[18 + 3 * i] = rgb[3 * i + 2]; out[18 + 3 * i + 1] = rgb[3 * i + 1]; out[18 + 3 * i + 2] = rgb[3 * i];
    }
    return out;
}

// farbfeld: "farbfeld", width and height as big-endian u32, then 16-bit big-endian RGBA (an 8-bit value v becomes v * 257).
std::vector<uint8_t> encode_farbfeld(uint32_t w, uint32_t h, const uint8_t* rgb) {
    std::vector<uint8_t> out;
    out.reserve(16 + (size_t)w * h * 8);
    const char* magic = "farbfeld";
    out.insert(out.end(), magic, magic + 8);
    be32(out, w);
    be32(out, h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        for (int c = 0; c < 3; c++) { out.push_back(rgb[3 * i + c]); out.push_back(rgb[3 * i + c]); }
        out.push_back(0xff); out.push_back(0xff);
    }
    return out;
}

// PAM (P7), RGB, one byte a sample.
std::vector<uint8_t> encode_pam(uint32_t w, uint32_t h, const uint8_t* rgb) {
    std::string head = "P7\nWIDTH " + std::to_string(w) + "\nHEIGHT " + std::to_string(h) + "\nDEPTH 3\nMAXVAL 255\nTUPLTYPE RGB\nENDHDR\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    out.insert(out.end(), rgb, rgb + (size_t)w * h * 3);
    return out;
}

// Baseline TIFF 6.0, little-endian, one uncompressed RGB strip.
std::vector<uint8_t> encode_tiff(uint32_t w, uint32_t h, const uint8_t* rgb) {
    const uint64_t pixels = (uint64_t)w * h * 3;
    if (pixels > 0xfff00000ull) throw std::runtime_error("the image is too large for a .tiff file (4 GB)");
    const uint32_t n_entries = 10, data_at = 8, bits_at = data_at + (uint32_t)pixels + ((uint32_t)pixels & 1u),
                   ifd_at = bits_at + 6 + 2;  // (three shorts, padded to a word boundary)
    std::vector<uint8_t> out((size_t)ifd_at + 2 + n_entries * 12 + 4, 0);
    auto le16 = [&](size_t at, uint32_t v) { out[at] = (uint8_t)v; out[at + 1] = (uint8_t)(v >> 8); };
    auto le32 = [&](size_t at, uint32_t v) { le16(at, v & 0xffffu); le16(at + 2, v >> 16); };
    out[0] = 'I'; out[1] = 'I';
    le16(2, 42);
    le32(4, ifd_at);
    std::memcpy(&out[data_at], rgb, (size_t)pixels);
    le16(bits_at, 8); le16(bits_at + 2, 8); le16(bits_at + 4, 8);
    size_t at = ifd_at;
    le16(at, n_entries);
    at += 2;
    auto entry = [&](uint32_t tag, uint32_t type, uint32_t count, uint32_t value) {  // type 3 = SHORT, 4 = LONG
        le16(at, tag); le16(at + 2, type); le32(at + 4, count);
        if (type == 3 && count == 1) le16(at + 8, value); else le32(at + 8, value);
        at += 12;
    };
    entry(256, 4, 1, w);              // ImageWidth
    entry(257, 4, 1, h);              // ImageLength
    entry(258, 3, 3, bits_at);        // BitsPerSample -> 8, 8, 8
    entry(259, 3, 1, 1);              // Compression: none
    entry(262, 3, 1, 2);              // PhotometricInterpretation: RGB
    entry(273, 4, 1, data_at);        // StripOffsets
    entry(277, 3, 1, 3);              // SamplesPerPixel
    entry(278, 4, 1, h);              // RowsPerStrip
    entry(279, 4, 1, (uint32_t)pixels);  // StripByteCounts
    entry(284, 3, 1, 1);              // PlanarConfiguration: chunky
    le32(at, 0);                      // no further IFD
    return out;
}

// ---- baseline sequential JPEG (ITU T.81): YCbCr 4:4:4, the Annex K tables, quality as in libjpeg.
// Lossy, so there is nothing bit-exact to match in the reference's encoder (image 0.23 + its own
// tables: parity unpinned); the test decodes the file again and compares with the input.
struct JpegBits {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            const uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xff) out.push_back(0);  // byte stuffing
            n -= 8;
        }
    }
    void flush() {
        if (n > 0) put(0x7f, 8 - n);  // pad with ones
    }
};
struct JpegHuff {
    uint16_t code[256];
    uint8_t len[256];
    JpegHuff(const uint8_t* bits, const uint8_t* vals) {
        std::memset(len, 0, sizeof(len));
        std::memset(code, 0, sizeof(code));
        uint16_t c = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            for (int i = 0; i < bits[l - 1]; i++, k++) {
                code[vals[k]] = c++;
                len[vals[k]] = (uint8_t)l;
            }
            c = (uint16_t)(c << 1);
        }
    }
};
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQuantLuma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQuantChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                  99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void jpeg_segment(std::vector<uint8_t>& out, uint8_t marker, const std::vector<uint8_t>& body) {
    out.push_back(0xff);
    out.push_back(marker);
    const size_t len = body.size() + 2;
    out.push_back((uint8_t)(len >> 8));
    out.push_back((uint8_t)len);
    out.insert(out.end(), body.begin(), body.end());
}

std::vector<uint8_t> encode_jpeg(uint32_t w, uint32_t h, const uint8_t* rgb, int quality) {
    if (w > 65535 || h > 65535) throw std::runtime_error("Image too large for JPEG (65535 pixels per side)");
    quality = std::min(100, std::max(1, quality));
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t qt[2][64];
    for (int i = 0; i < 64; i++) {
        qt[0][i] = (uint8_t)std::min(255, std::max(1, (kQuantLuma[i] * scale + 50) / 100));
        qt[1][i] = (uint8_t)std::min(255, std::max(1, (kQuantChroma[i] * scale + 50) / 100));
    }
    std::vector<uint8_t> out{0xff, 0xd8};
    jpeg_segment(out, 0xe0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < 2; t++) {
        std::vector<uint8_t> body{(uint8_t)t};
        for (int i = 0; i < 64; i++) body.push_back(qt[t][kZigzag[i]]);
        jpeg_segment(out, 0xdb, body);
    }
    jpeg_segment(out, 0xc0, {8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1});
    auto dht = [&](uint8_t id, const uint8_t* bits, const uint8_t* vals, int n) {
        std::vector<uint8_t> body{id};
        body.insert(body.end(), bits, bits + 16);
        body.insert(body.end(), vals, vals + n);
        jpeg_segment(out, 0xc4, body);
    };
    dht(0x00, kDcLumaBits, kDcVals, 12);
    dht(0x10, kAcLumaBits, kAcLumaVals, 162);
    dht(0x01, kDcChromaBits, kDcVals, 12);
    dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    jpeg_segment(out, 0xda, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});

    const JpegHuff dc[2] = {JpegHuff(kDcLumaBits, kDcVals), JpegHuff(kDcChromaBits, kDcVals)};
    const JpegHuff ac[2] = {JpegHuff(kAcLumaBits, kAcLumaVals), JpegHuff(kAcChromaBits, kAcChromaVals)};
    double cosine[8][8];  // cosine[u][x] = c(u)/2 cos((2x+1) u pi / 16)
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            cosine[u][x] = (u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    JpegBits bits{out};
    int prev_dc[3] = {0, 0, 0};
    auto magnitude = [](int v, int* nbits, uint32_t* extra) {
        int a = v < 0 ? -v : v, n = 0;
        while (a) {
            n++;
            a >>= 1;
        }
        *nbits = n;
        *extra = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << n) - 1u);
    };
    for (uint32_t by = 0; by < h; by += 8)
        for (uint32_t bx = 0; bx < w; bx += 8) {
            double comp[3][64];
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {  // edge pixels repeat into the padding
                    const uint8_t* px = rgb + ((size_t)std::min(h - 1, by + y) * w + std::min(w - 1, bx + x)) * 3;
                    const double r = px[0], g = px[1], b = px[2];
                    comp[0][y * 8 + x] = 0.299 * r + 0.587 * g + 0.114 * b - 128.0;
                    comp[1][y * 8 + x] = -0.168736 * r - 0.331264 * g + 0.5 * b;
                    comp[2][y * 8 + x] = 0.5 * r - 0.418688 * g - 0.081312 * b;
                }
            for (int c = 0; c < 3; c++) {
                const int t = c == 0 ? 0 : 1;
                double tmp[64], freq[64];
                for (int y = 0; y < 8; y++)
                    for (int u = 0; u < 8; u++) {
                        double sum = 0;
                        for (int x = 0; x < 8; x++) sum += comp[c][y * 8 + x] * cosine[u][x];
                        tmp[y * 8 + u] = sum;
                    }
                for (int v = 0; v < 8; v++)
                    for (int u = 0; u < 8; u++) {
                        double sum = 0;
                        for (int y = 0; y < 8; y++) sum += tmp[y * 8 + u] * cosine[v][y];
                        freq[v * 8 + u] = sum;
                    }
                int q[64];
                for (int i = 0; i < 64; i++) q[i] = (int)std::lround(freq[kZigzag[i]] / qt[t][kZigzag[i]]);
                int nbits;
                uint32_t extra;
                magnitude(q[0] - prev_dc[c], &nbits, &extra);
                prev_dc[c] = q[0];
                bits.put(dc[t].code[nbits], dc[t].len[nbits]);
                if (nbits) bits.put(extra, nbits);
                int run = 0;
                for (int i = 1; i < 64; i++) {
                    if (q[i] == 0) {
                        run++;
                        continue;
                    }
                    for (; run >= 16; run -= 16) bits.put(ac[t].code[0xf0], ac[t].len[0xf0]);
                    magnitude(q[i], &nbits, &extra);
                    const int sym = run << 4 | nbits;
                    bits.put(ac[t].code[sym], ac[t].len[sym]);
                    bits.put(extra, nbits);
                    run = 0;
                }
                if (run) bits.put(ac[t].code[0x00], ac[t].len[0x00]);  // end of block
            }
        }
    bits.flush();
    out.push_back(0xff);
    out.push_back(0xd9);
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
    else if (ext == "jpg" || ext == "jpeg") write_all(path, encode_jpeg(width, height, rgb, 75));
    else if (ext == "tga") write_all(path, encode_tga(width, height, rgb));
    else if (ext == "tiff" || ext == "tif") write_all(path, encode_tiff(width, height, rgb));
    else if (ext == "ff" || ext == "farbfeld") write_all(path, encode_farbfeld(width, height, rgb));
    else if (ext == "pam") write_all(path, encode_pam(width, height, rgb));
    else
        throw std::runtime_error("The image format could not be determined or is not supported by this build (." + ext +
                                 "): use .png, .jpg, .ppm, .pam, .bmp, .tga, .tiff or .ff");
}

}  // namespace rt
