// Image input for `imageTexture` textures: baseline JPEG and binary PPM (P6).
//
// Stands in for image 0.23.12 + jpeg-decoder 0.1.20 (Cargo.lock:275-276,294-295) as used
// at reference src/textures.rs:32-36 (decode to RGB8).  Those crates are not vendored; a
// different IDCT / colour-conversion rounding can move a texel by +-1 LSB, which is far
// below the image-level tolerance - parity for it is unpinned.
// Decodes baseline sequential DCT (SOF0/SOF1, 8-bit, Huffman), 1 or 3 components, any
// sampling factors (chroma is upsampled by replication), restart intervals.
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "scene_config.hpp"

namespace rt {

namespace {

struct Huff {
    // Canonical Huffman decode tables (JPEG Annex F.2.2.3).
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // allocated size in 8x8 blocks (MCU padded)
    std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
    int dc_pred = 0;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill() {
        while (nbits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    int nx = p < end ? *p : 0xD9;
                    if (nx == 0x00) {
                        p++;
                    } else {  // a marker: feed zeros from here on
                        hit_marker = true;
                        p--;
                        byte = 0;
                    }
                }
            }
            acc |= (uint32_t)byte << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    void reset() {
        acc = 0;
        nbits = 0;
        hit_marker = false;
    }
};

void build_huff(Huff& h, const uint8_t* counts, const uint8_t* symbols, int nsym) {
    std::memcpy(h.vals, symbols, (size_t)nsym);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
        h.valptr[len] = k;
        h.mincode[len] = code;
        code += counts[len - 1];
        k += counts[len - 1];
        h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.present = true;
}

int decode_symbol(BitReader& br, const Huff& h) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.get(1);
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len])
            return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    throw SceneError("JPEG: bad Huffman code");
}

int extend(int v, int t) { return t && v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const double* in, uint8_t* out, int stride) {
    // Separable double-precision inverse DCT (reference-accurate, not fast; textures are
    // decoded once per scene).
    static double c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++)
                c[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
        init = true;
    }
    double tmp[64];
    for (int y = 0; y < 8; y++)      // rows: over u
        for (int x = 0; x < 8; x++) {
            double s = 0;
            for (int u = 0; u < 8; u++) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) {
            double s = 0;
            for (int v = 0; v < 8; v++) s += c[y][v] * tmp[v * 8 + x];
            int p = (int)std::lround(s + 128.0);
            out[y * stride + x] = (uint8_t)(p < 0 ? 0 : p > 255 ? 255 : p);
        }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

ImageRGB8 decode_jpeg_baseline(const uint8_t* d, size_t size) {
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) throw SceneError("JPEG: missing SOI marker");
    uint16_t qt[4][64] = {};
    Huff hdc[4], hac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, restart_interval = 0, hmax = 1, vmax = 1;
    bool adobe = false;
    int adobe_transform = -1;
    size_t i = 2;
    for (;;) {
        if (i + 4 > size) throw SceneError("JPEG: truncated file");
        if (d[i] != 0xFF) throw SceneError("JPEG: expected a marker");
        while (i + 2 < size && d[i + 1] == 0xFF) i++;  // fill bytes before a marker
        int m = d[i + 1];
        i += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) throw SceneError("JPEG: no image data");
        if (i + 2 > size) throw SceneError("JPEG: truncated file");
        size_t len = be16(d + i);
        if (len < 2 || i + len > size) throw SceneError("JPEG: bad segment length");
        const uint8_t* seg = d + i + 2;
        size_t seglen = len - 2;
        if (m == 0xDB) {  // DQT
            size_t k = 0;
            while (k < seglen) {
                int pq = seg[k] >> 4, tq = seg[k] & 15;
                k++;
                if (tq > 3 || pq > 1) throw SceneError("JPEG: bad quantisation table id");
                if (k + (pq ? 128u : 64u) > seglen) throw SceneError("JPEG: truncated quantisation table");
                for (int z = 0; z < 64; z++) {
                    if (pq) {
                        qt[tq][z] = be16(seg + k);
                        k += 2;
                    } else {
                        qt[tq][z] = seg[k++];
                    }
                }
            }
        } else if (m == 0xC4) {  // DHT
            size_t k = 0;
            while (k + 17 <= seglen) {
                int tc = seg[k] >> 4, th = seg[k] & 15;
                if (th > 3 || tc > 1) throw SceneError("JPEG: bad Huffman table id");
                const uint8_t* counts = seg + k + 1;
                int nsym = 0;
                for (int z = 0; z < 16; z++) nsym += counts[z];
                if (nsym > 256 || k + 17 + (size_t)nsym > seglen) throw SceneError("JPEG: bad Huffman table");
                build_huff(tc ? hac[th] : hdc[th], counts, seg + k + 17, nsym);
                k += 17 + (size_t)nsym;
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0 / SOF1 (Huffman, sequential)
            if (seglen < 6) throw SceneError("JPEG: truncated frame header");
            if (seg[0] != 8) throw SceneError("JPEG: only 8-bit precision is supported");
            height = be16(seg + 1);
            width = be16(seg + 3);
            int n = seg[5];
            if ((n != 1 && n != 3) || width == 0 || height == 0)
                throw SceneError("JPEG: unsupported component count or size");
            if (seglen < 6 + 3 * (size_t)n) throw SceneError("JPEG: truncated frame header");
            comps.resize((size_t)n);
            for (int c = 0; c < n; c++) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4;
                comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3)
                    throw SceneError("JPEG: bad component parameters");
                if (comps[c].h > hmax) hmax = comps[c].h;
                if (comps[c].v > vmax) vmax = comps[c].v;
            }
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw SceneError("JPEG: only baseline (SOF0/SOF1) files are supported");
        } else if (m == 0xDD) {
            if (seglen < 2) throw SceneError("JPEG: truncated restart interval");
            restart_interval = be16(seg);
        } else if (m == 0xEE && seglen >= 12 && std::memcmp(seg, "Adobe", 5) == 0) {
            adobe = true;
            adobe_transform = seg[11];
        } else if (m == 0xDA) {  // SOS
            if (comps.empty()) throw SceneError("JPEG: scan before frame header");
            if (seglen < 1) throw SceneError("JPEG: truncated scan header");
            int ns = seg[0];
            if (ns != (int)comps.size()) throw SceneError("JPEG: non-interleaved scans are not supported");
            if (seglen < 1 + 2 * (size_t)ns) throw SceneError("JPEG: truncated scan header");
            for (int s = 0; s < ns; s++) {
                int cid = seg[1 + 2 * s];
                int tab = seg[2 + 2 * s];
                if ((tab >> 4) > 3 || (tab & 15) > 3) throw SceneError("JPEG: bad Huffman table id in scan header");
                bool found = false;
                for (auto& c : comps)
                    if (c.id == cid) {
                        c.td = tab >> 4;
                        c.ta = tab & 15;
                        found = true;
                    }
                if (!found) throw SceneError("JPEG: scan names an unknown component");
            }
            i += len;
            break;
        }
        i += len;
    }
    if (comps.empty()) throw SceneError("JPEG: no frame header");

    int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
    for (auto& c : comps) {
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        if (!hdc[c.td].present || !hac[c.ta].present) throw SceneError("JPEG: missing Huffman table");
    }
    BitReader br{d + i, d + size};
    int mcu_count = 0;
    int next_rst = 0;
    for (int my = 0; my < mcus_y; my++) {
        for (int mx = 0; mx < mcus_x; mx++) {
            if (restart_interval && mcu_count && mcu_count % restart_interval == 0) {
                // Skip to the RSTn marker and reset the predictors.
                const uint8_t* p = br.p;
                while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) p++;
                if (p + 1 >= br.end) throw SceneError("JPEG: missing restart marker");
                (void)next_rst;
                br.p = p + 2;
                br.reset();
                for (auto& c : comps) c.dc_pred = 0;
            }
            for (auto& c : comps) {
                for (int by = 0; by < c.v; by++) {
                    for (int bx = 0; bx < c.h; bx++) {
                        double coef[64] = {};
                        int t = decode_symbol(br, hdc[c.td]);
                        if (t > 16) throw SceneError("JPEG: bad DC coefficient size");
                        int diff = t ? extend(br.get(t), t) : 0;
                        c.dc_pred += diff;
                        coef[0] = (double)c.dc_pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            int rs = decode_symbol(br, hac[c.ta]);
                            int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r == 15) {
                                    k += 16;
                                    continue;
                                }
                                break;  // EOB
                            }
                            k += r;
                            if (k > 63) throw SceneError("JPEG: coefficient index out of range");
                            coef[kZigzag[k]] = (double)extend(br.get(s), s) * qt[c.tq][k];
                            k++;
                        }
                        int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                        idct8x8(coef, &c.plane[(size_t)py * c.blocks_w * 8 + px], c.blocks_w * 8);
                    }
                }
            }
            mcu_count++;
        }
    }

    ImageRGB8 img;
    img.width = width;
    img.height = height;
    img.rgb.resize((size_t)width * height * 3);
    bool ycc = comps.size() == 3 && !(adobe && adobe_transform == 0);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            int s[3];
            for (size_t c = 0; c < comps.size(); c++) {
                const Component& cc = comps[c];
                int sx = x * cc.h / hmax, sy = y * cc.v / vmax;
                s[c] = cc.plane[(size_t)sy * cc.blocks_w * 8 + sx];
            }
            uint8_t* o = &img.rgb[((size_t)y * width + x) * 3];
            if (comps.size() == 1) {
                o[0] = o[1] = o[2] = (uint8_t)s[0];
            } else if (ycc) {
                double Y = s[0], cb = s[1] - 128.0, cr = s[2] - 128.0;
                double rgb[3] = {Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb};
                for (int k = 0; k < 3; k++) {
                    int p = (int)std::lround(rgb[k]);
                    o[k] = (uint8_t)(p < 0 ? 0 : p > 255 ? 255 : p);
                }
            } else {
                o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2];
            }
        }
    }
    return img;
}

namespace {

ImageRGB8 decode_ppm(const std::string& data, const std::string& path) {
    // Binary P6, maxval <= 255.
    size_t pos = 2;
    auto next_int = [&]() -> int {
        for (;;) {
            while (pos < data.size() && std::isspace((unsigned char)data[pos])) pos++;
            if (pos < data.size() && data[pos] == '#') {
                while (pos < data.size() && data[pos] != '\n') pos++;
            } else {
                break;
            }
        }
        int v = 0;
        bool any = false;
        while (pos < data.size() && std::isdigit((unsigned char)data[pos])) {
            v = v * 10 + (data[pos++] - '0');
            any = true;
        }
        if (!any) throw SceneError("PPM: bad header in " + path);
        return v;
    };
    ImageRGB8 img;
    img.width = next_int();
    img.height = next_int();
    int maxval = next_int();
    pos++;  // single whitespace after maxval
    if (maxval <= 0 || maxval > 255) throw SceneError("PPM: unsupported maxval in " + path);
    size_t n = (size_t)img.width * img.height * 3;
    if (data.size() < pos + n) throw SceneError("PPM: truncated data in " + path);
    img.rgb.assign(data.begin() + (long)pos, data.begin() + (long)(pos + n));
    if (maxval != 255)
        for (auto& b : img.rgb) b = (uint8_t)((int)b * 255 / maxval);
    return img;
}

}  // namespace

ImageRGB8 load_image_rgb8(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError("cannot open image file " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string data = ss.str();
    if (data.size() >= 2 && (uint8_t)data[0] == 0xFF && (uint8_t)data[1] == 0xD8)
        return decode_jpeg_baseline((const uint8_t*)data.data(), data.size());
    if (data.size() >= 2 && data[0] == 'P' && data[1] == '6') return decode_ppm(data, path);
    throw SceneError("unsupported image format (baseline JPEG and binary PPM are supported): " + path);
}

}  // namespace rt
