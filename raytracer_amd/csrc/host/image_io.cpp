// Image input for `imageTexture` textures: JPEG (baseline, extended and progressive Huffman), PNG, BMP and PNM.
//
// Stands in for image 0.23.12 (jpeg-decoder 0.1.20, png 0.16.8; Cargo.lock:275-276,294-295, 541-542) as the reference
// uses it at src/textures.rs:32-36: `image::io::Reader::open(filename)?.decode()?.into_rgb8()` - any format the crate
// decodes, converted to RGB8 (alpha dropped, grey replicated, 16-bit samples narrowed).  Those crates are not vendored and
// the reference holds no decoded texel, so parity with them is UNPINNED: the tests compare with PIL (PNG, BMP, PNM exact;
// JPEG within a few LSB - a different IDCT and colour-conversion rounding can move a texel by +-1, far below the image-level
// tolerance).  What is restated from the crates' published behaviour:
//   * JPEG chroma up-sampling as jpeg-decoder does it: h2v1 by the triangle filter (3 near + far + 2) >> 2, h2v2 by the
//     same filter in both directions ((3 t_near + t_far + 8) >> 4 over t = 3 near_row + far_row), other ratios by replication;
//   * 16-bit PNG samples to 8 bits by (x + 128) / 257 (the inverse of the 8 -> 16 bit replication x * 257);
//   * PNG gamma / colour-profile chunks are ignored, palette transparency is dropped.
// Texture files are untrusted input: every length, index and table of the formats is checked, and a malformed file is a
// SceneError, never a fault (tests/test_host_io.py, scripts/fuzz_jpeg.cpp).
#include <zlib.h>

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
    int own_w = 0, own_h = 0;        // blocks of the component's own extent (what a non-interleaved scan covers)
    std::vector<int16_t> coef;       // blocks_w * blocks_h * 64 quantised coefficients, natural (row-major) order
    std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8 samples after the inverse DCT
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
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int16_t clamp_coef(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

// ---- one scan of a JPEG frame (sequential or progressive) into the components' coefficient arrays
struct Scan {
    std::vector<Component*> comps;
    int ss = 0, se = 63, ah = 0, al = 0;
};

struct ScanDecoder {
    BitReader& br;
    const Huff* hdc;
    const Huff* hac;
    bool progressive;
    const Scan& scan;
    int eobrun = 0;

    void block_sequential(Component& c, int16_t* coef) {
        int t = decode_symbol(br, hdc[c.td]);
        if (t > 16) throw SceneError("JPEG: bad DC coefficient size");
        int diff = t ? extend(br.get(t), t) : 0;
        c.dc_pred += diff;
        coef[0] = clamp_coef(c.dc_pred);
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
            coef[kZigzag[k]] = clamp_coef(extend(br.get(s), s));
            k++;
        }
    }
    // Progressive scans, ITU T.81 Annex G: DC first / refinement, AC first / refinement with end-of-band runs.
    void block_dc(Component& c, int16_t* coef) {
        if (scan.ah == 0) {
            int t = decode_symbol(br, hdc[c.td]);
            if (t > 16) throw SceneError("JPEG: bad DC coefficient size");
            int diff = t ? extend(br.get(t), t) : 0;
            c.dc_pred += diff;
            coef[0] = clamp_coef(c.dc_pred * (1 << scan.al));
        } else if (br.get(1)) {
            coef[0] = (int16_t)(coef[0] | (1 << scan.al));
        }
    }
    void block_ac_first(Component& c, int16_t* coef) {
        if (eobrun > 0) {
            eobrun--;
            return;
        }
        for (int k = scan.ss; k <= scan.se;) {
            int rs = decode_symbol(br, hac[c.ta]);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += br.get(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > scan.se) throw SceneError("JPEG: coefficient index out of range");
            coef[kZigzag[k]] = clamp_coef(extend(br.get(s), s) * (1 << scan.al));
            k++;
        }
    }
    void block_ac_refine(Component& c, int16_t* coef) {
        const int p1 = 1 << scan.al, m1 = -1 * (1 << scan.al);
        int k = scan.ss;
        if (eobrun <= 0) {
            for (; k <= scan.se;) {
                int rs = decode_symbol(br, hac[c.ta]);
                int r = rs >> 4, s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1 << r);
                        if (r) eobrun += br.get(r);
                        break;  // the rest of the band is refined below, as part of the run
                    }
                } else {
                    if (s != 1) throw SceneError("JPEG: bad refinement coefficient");
                    value = br.get(1) ? p1 : m1;
                }
                // skip r zero-history coefficients, refining the non-zero ones met on the way
                while (k <= scan.se) {
                    int16_t& z = coef[kZigzag[k]];
                    if (z != 0) {
                        if (br.get(1) && (z & p1) == 0) z = (int16_t)(z >= 0 ? z + p1 : z + m1);
                    } else {
                        if (r == 0) {
                            if (value) z = (int16_t)value;
                            k++;
                            break;
                        }
                        r--;
                    }
                    k++;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= scan.se; k++) {
                int16_t& z = coef[kZigzag[k]];
                if (z != 0 && br.get(1) && (z & p1) == 0) z = (int16_t)(z >= 0 ? z + p1 : z + m1);
            }
            eobrun--;
        }
    }
    void block(Component& c, int16_t* coef) {
        if (!progressive) block_sequential(c, coef);
        else if (scan.ss == 0) block_dc(c, coef);
        else if (scan.ah == 0) block_ac_first(c, coef);
        else block_ac_refine(c, coef);
    }
};

// jpeg-decoder's chroma up-samplers (unpinned restatement, see the head of the file).
void upsample_h2v1_row(const uint8_t* in, int in_w, uint8_t* out) {
    if (in_w == 1) {
        out[0] = out[1] = in[0];
        return;
    }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < in_w - 1; i++) {
        int sample = 3 * in[i] + 2;
        out[i * 2] = (uint8_t)((sample + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((sample + in[i + 1]) >> 2);
    }
    out[(in_w - 1) * 2] = (uint8_t)((in[in_w - 1] * 3 + in[in_w - 2] + 2) >> 2);
    out[(in_w - 1) * 2 + 1] = in[in_w - 1];
}
void upsample_h2v2_row(const uint8_t* near, const uint8_t* far, int in_w, uint8_t* out) {
    if (in_w == 1) {
        out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
        return;
    }
    int t0 = 3 * near[0] + far[0], t1 = 3 * near[1] + far[1];
    out[0] = (uint8_t)((t0 * 4 + 8) >> 4);
    out[1] = (uint8_t)((t0 * 3 + t1 + 8) >> 4);
    for (int i = 2; i < in_w; i++) {
        int t2 = 3 * near[i] + far[i];
        out[i * 2 - 2] = (uint8_t)((t1 * 3 + t0 + 8) >> 4);
        out[i * 2 - 1] = (uint8_t)((t1 * 3 + t2 + 8) >> 4);
        t0 = t1;
        t1 = t2;
    }
    out[in_w * 2 - 2] = (uint8_t)((t1 * 3 + t0 + 8) >> 4);
    out[in_w * 2 - 1] = (uint8_t)((t1 * 4 + 8) >> 4);
}

}  // namespace

ImageRGB8 decode_jpeg(const uint8_t* d, size_t size) {
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) throw SceneError("JPEG: missing SOI marker");
    uint16_t qt[4][64] = {};
    Huff hdc[4], hac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, restart_interval = 0, hmax = 1, vmax = 1;
    bool adobe = false, progressive = false, have_frame = false, any_scan = false;
    int adobe_transform = -1;
    size_t i = 2;
    for (;;) {
        if (i + 4 > size) {
            if (any_scan) break;  // (a file that ends without EOI after its last scan is decoded as far as it goes)
            throw SceneError("JPEG: truncated file");
        }
        if (d[i] != 0xFF) {
            if (any_scan) {  // entropy-coded bytes left over after a scan: skip to the next marker
                i++;
                continue;
            }
            throw SceneError("JPEG: expected a marker");
        }
        while (i + 2 < size && d[i + 1] == 0xFF) i++;  // fill bytes before a marker
        int m = d[i + 1];
        i += 2;
        if (m == 0x00 && any_scan) continue;  // a stuffed 0xFF inside leftover scan data
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) {
            if (any_scan) break;
            throw SceneError("JPEG: no image data");
        }
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
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // SOF0 / SOF1 (sequential) / SOF2 (progressive), Huffman
            if (have_frame) throw SceneError("JPEG: more than one frame header");
            if (seglen < 6) throw SceneError("JPEG: truncated frame header");
            if (seg[0] != 8) throw SceneError("JPEG: only 8-bit precision is supported");
            progressive = m == 0xC2;
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
            if ((uint64_t)width * height > (1ull << 28)) throw SceneError("JPEG: image too large");
            const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
            for (auto& c : comps) {
                c.blocks_w = mcus_x * c.h;
                c.blocks_h = mcus_y * c.v;
                c.own_w = ((width * c.h + hmax - 1) / hmax + 7) / 8;
                c.own_h = ((height * c.v + vmax - 1) / vmax + 7) / 8;
                c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
            }
            have_frame = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            throw SceneError("JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
        } else if (m == 0xDD) {
            if (seglen < 2) throw SceneError("JPEG: truncated restart interval");
            restart_interval = be16(seg);
        } else if (m == 0xEE && seglen >= 12 && std::memcmp(seg, "Adobe", 5) == 0) {
            adobe = true;
            adobe_transform = seg[11];
        } else if (m == 0xDA) {  // SOS: one scan
            if (!have_frame) throw SceneError("JPEG: scan before frame header");
            if (seglen < 1) throw SceneError("JPEG: truncated scan header");
            int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size()) throw SceneError("JPEG: bad component count in scan header");
            if (seglen < 1 + 2 * (size_t)ns + 3) throw SceneError("JPEG: truncated scan header");
            Scan scan;
            for (int s = 0; s < ns; s++) {
                int cid = seg[1 + 2 * s];
                int tab = seg[2 + 2 * s];
                if ((tab >> 4) > 3 || (tab & 15) > 3) throw SceneError("JPEG: bad Huffman table id in scan header");
                Component* found = nullptr;
                for (auto& c : comps)
                    if (c.id == cid) found = &c;
                if (!found) throw SceneError("JPEG: scan names an unknown component");
                for (Component* seen : scan.comps)
                    if (seen == found) throw SceneError("JPEG: scan names a component twice");
                found->td = tab >> 4;
                found->ta = tab & 15;
                scan.comps.push_back(found);
            }
            scan.ss = seg[1 + 2 * ns];
            scan.se = seg[2 + 2 * ns];
            scan.ah = seg[3 + 2 * ns] >> 4;
            scan.al = seg[3 + 2 * ns] & 15;
            if (!progressive) {
                scan.ss = 0; scan.se = 63; scan.ah = 0; scan.al = 0;
            } else {
                if (scan.ss > scan.se || scan.se > 63 || scan.al > 13 || scan.ah > 13 || (scan.ss == 0 && scan.se != 0) ||
                    (scan.ss > 0 && ns != 1))
                    throw SceneError("JPEG: bad spectral selection / successive approximation in scan header");
            }
            for (Component* c : scan.comps) {
                const bool needs_dc = !progressive || (scan.ss == 0 && scan.ah == 0);
                const bool needs_ac = !progressive || scan.ss > 0;
                if ((needs_dc && !hdc[c->td].present) || (needs_ac && !hac[c->ta].present)) throw SceneError("JPEG: missing Huffman table");
                c->dc_pred = 0;
            }
            i += len;
            // entropy-coded segment
            BitReader br{d + i, d + size};
            ScanDecoder dec{br, hdc, hac, progressive, scan};
            const bool interleaved = scan.comps.size() > 1;
            const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
            const int units_x = interleaved ? mcus_x : scan.comps[0]->own_w, units_y = interleaved ? mcus_y : scan.comps[0]->own_h;
            int unit_count = 0;
            for (int uy = 0; uy < units_y; uy++) {
                for (int ux = 0; ux < units_x; ux++) {
                    if (restart_interval && unit_count && unit_count % restart_interval == 0) {
                        // Skip to the RSTn marker and reset the predictors.
                        const uint8_t* p = br.p;
                        while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) p++;
                        if (p + 1 >= br.end) throw SceneError("JPEG: missing restart marker");
                        br.p = p + 2;
                        br.reset();
                        for (Component* c : scan.comps) c->dc_pred = 0;
                        dec.eobrun = 0;
                    }
                    if (interleaved) {
                        for (Component* c : scan.comps)
                            for (int by = 0; by < c->v; by++)
                                for (int bx = 0; bx < c->h; bx++)
                                    dec.block(*c, &c->coef[((size_t)(uy * c->v + by) * c->blocks_w + (ux * c->h + bx)) * 64]);
                    } else {
                        Component* c = scan.comps[0];
                        dec.block(*c, &c->coef[((size_t)uy * c->blocks_w + ux) * 64]);
                    }
                    unit_count++;
                }
            }
            // on to the next marker: what the bit reader has not consumed
            i = (size_t)(br.p - d);
            any_scan = true;
            if (!progressive) break;  // (baseline files with one scan per component are rare; further scans are read below)
            continue;
        }
        i += len;
    }
    if (!have_frame || !any_scan) throw SceneError("JPEG: no frame header");
    if (!progressive) {
        // a sequential file may still hold one scan per component: read the scans that follow the first
        bool more = false;
        for (size_t k = i; k + 1 < size; k++)
            if (d[k] == 0xFF && d[k + 1] == 0xDA) more = true;
        if (more) throw SceneError("JPEG: sequential files with several scans are not supported");
    }

    // dequantise, inverse DCT
    for (auto& c : comps) {
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        for (int by = 0; by < c.blocks_h; by++)
            for (int bx = 0; bx < c.blocks_w; bx++) {
                const int16_t* q = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
                double coef[64];
                for (int k = 0; k < 64; k++) coef[kZigzag[k]] = (double)q[kZigzag[k]] * qt[c.tq][k];
                idct8x8(coef, &c.plane[(size_t)by * 8 * c.blocks_w * 8 + bx * 8], c.blocks_w * 8);
            }
        std::vector<int16_t>().swap(c.coef);
    }

    // chroma up-sampling to the luma grid (full planes of width x height samples)
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const Component& cc = comps[ci];
        const int stride = cc.blocks_w * 8;
        const int cw = (width * cc.h + hmax - 1) / hmax, ch = (height * cc.v + vmax - 1) / vmax;  // the component's own size
        std::vector<uint8_t>& out = full[ci];
        out.resize((size_t)width * height);
        if (cc.h == hmax && cc.v == vmax) {
            for (int y = 0; y < height; y++) std::memcpy(&out[(size_t)y * width], &cc.plane[(size_t)y * stride], (size_t)width);
        } else if (cc.h * 2 == hmax && cc.v == vmax) {
            std::vector<uint8_t> row((size_t)cw * 2);
            for (int y = 0; y < height; y++) {
                upsample_h2v1_row(&cc.plane[(size_t)y * stride], cw, row.data());
                std::memcpy(&out[(size_t)y * width], row.data(), (size_t)width);
            }
        } else if (cc.h * 2 == hmax && cc.v * 2 == vmax) {
            std::vector<uint8_t> row((size_t)cw * 2);
            for (int y = 0; y < height; y++) {
                const int near = y / 2;
                int far = (y & 1) ? near + 1 : near - 1;
                far = far < 0 ? 0 : far > ch - 1 ? ch - 1 : far;
                upsample_h2v2_row(&cc.plane[(size_t)near * stride], &cc.plane[(size_t)far * stride], cw, row.data());
                std::memcpy(&out[(size_t)y * width], row.data(), (size_t)width);
            }
        } else {
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) out[(size_t)y * width + x] = cc.plane[(size_t)(y * cc.v / vmax) * stride + x * cc.h / hmax];
        }
    }

    ImageRGB8 img;
    img.width = width;
    img.height = height;
    img.rgb.resize((size_t)width * height * 3);
    bool ycc = comps.size() == 3 && !(adobe && adobe_transform == 0);
    for (size_t px = 0; px < (size_t)width * height; px++) {
        uint8_t* o = &img.rgb[px * 3];
        if (comps.size() == 1) {
            o[0] = o[1] = o[2] = full[0][px];
        } else if (ycc) {
            double Y = full[0][px], cb = full[1][px] - 128.0, cr = full[2][px] - 128.0;
            double rgb[3] = {Y + 1.402 * cr, Y - 0.344136 * cb - 0.714136 * cr, Y + 1.772 * cb};
            for (int k = 0; k < 3; k++) {
                int p = (int)std::lround(rgb[k]);
                o[k] = (uint8_t)(p < 0 ? 0 : p > 255 ? 255 : p);
            }
        } else {
            o[0] = full[0][px]; o[1] = full[1][px]; o[2] = full[2][px];
        }
    }
    return img;
}

ImageRGB8 decode_jpeg_baseline(const uint8_t* d, size_t size) { return decode_jpeg(d, size); }

// ------------------------------------------------------------------ PNG (png 0.16.8 behind image 0.23.12)
// Every colour type and bit depth of the specification, interlaced or not: critical chunks IHDR / PLTE / IDAT / IEND (CRCs
// checked), the zlib stream inflated with zlib, the five row filters undone, samples widened / narrowed to RGB8.
ImageRGB8 decode_png(const uint8_t* d, size_t size) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || std::memcmp(d, sig, 8) != 0) throw SceneError("PNG: bad signature");
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> palette, idat;
    bool have_header = false, ended = false;
    size_t i = 8;
    while (!ended) {
        if (i + 12 > size) throw SceneError("PNG: truncated file");
        const uint32_t len = be32(d + i);
        if (len > size - i - 12) throw SceneError("PNG: bad chunk length");
        const uint8_t* type = d + i + 4;
        const uint8_t* body = d + i + 8;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(body + len)) throw SceneError("PNG: chunk checksum mismatch");
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (have_header || len != 13) throw SceneError("PNG: bad IHDR chunk");
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colour = body[9];
            interlace = body[12];
            const bool depth_ok = colour == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                                  : colour == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                  : (colour == 2 || colour == 4 || colour == 6) ? (depth == 8 || depth == 16)
                                                                                : false;
            if (width == 0 || height == 0 || !depth_ok || body[10] != 0 || body[11] != 0 || interlace > 1)
                throw SceneError("PNG: unsupported or invalid header");
            if ((uint64_t)width * height > (1ull << 28)) throw SceneError("PNG: image too large");
            have_header = true;
        } else if (!have_header) {
            throw SceneError("PNG: first chunk is not IHDR");
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) throw SceneError("PNG: bad palette");
            palette.assign(body, body + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            ended = true;
        } else if ((type[0] & 0x20) == 0) {
            throw SceneError("PNG: unknown critical chunk");
        }
        i += 12 + (size_t)len;
    }
    if (colour == 3 && palette.empty()) throw SceneError("PNG: palette image without a palette");
    const int channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : 4;
    const int bpp_bits = channels * depth;
    const size_t filter_bpp = (size_t)((bpp_bits + 7) / 8);
    // the sub-images of the stream: one for a non-interlaced file, up to seven Adam7 passes otherwise
    struct Pass { uint32_t x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlace) {
        passes.push_back({0, 0, 1, 1, width, height});
    } else {
        static const uint32_t px0[7] = {0, 4, 0, 2, 0, 1, 0}, py0[7] = {0, 0, 4, 0, 2, 0, 1};
        static const uint32_t pdx[7] = {8, 8, 4, 4, 2, 2, 1}, pdy[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int p = 0; p < 7; p++) {
            const uint32_t w = width > px0[p] ? (width - px0[p] + pdx[p] - 1) / pdx[p] : 0;
            const uint32_t h = height > py0[p] ? (height - py0[p] + pdy[p] - 1) / pdy[p] : 0;
            if (w && h) passes.push_back({px0[p], py0[p], pdx[p], pdy[p], w, h});
        }
    }
    size_t raw_size = 0;
    for (const Pass& p : passes) raw_size += (size_t)p.h * (1 + ((size_t)p.w * bpp_bits + 7) / 8);
    std::vector<uint8_t> raw(raw_size);
    {
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        if (inflateInit(&z) != Z_OK) throw SceneError("PNG: zlib initialisation failed");
        z.next_in = idat.data();
        z.avail_in = (uInt)idat.size();
        z.next_out = raw.data();
        z.avail_out = (uInt)raw.size();
        const int rc = inflate(&z, Z_FINISH);
        const size_t got = raw.size() - z.avail_out;
        inflateEnd(&z);
        if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size()) throw SceneError("PNG: corrupt or truncated image data");
    }
    ImageRGB8 img;
    img.width = (int)width;
    img.height = (int)height;
    img.rgb.assign((size_t)width * height * 3, 0);
    size_t pos = 0;
    for (const Pass& p : passes) {
        const size_t row_bytes = ((size_t)p.w * bpp_bits + 7) / 8;
        std::vector<uint8_t> prev(row_bytes, 0), cur(row_bytes);
        for (uint32_t y = 0; y < p.h; y++) {
            const int filter = raw[pos++];
            std::memcpy(cur.data(), &raw[pos], row_bytes);
            pos += row_bytes;
            for (size_t k = 0; k < row_bytes; k++) {
                const int a = k >= filter_bpp ? cur[k - filter_bpp] : 0, b = prev[k], c = k >= filter_bpp ? prev[k - filter_bpp] : 0;
                int add = 0;
                switch (filter) {
                    case 0: break;
                    case 1: add = a; break;
                    case 2: add = b; break;
                    case 3: add = (a + b) >> 1; break;
                    case 4: {
                        const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                        add = (pa <= pb && pa <= pc) ? a : (pb <= pc) ? b : c;
                        break;
                    }
                    default: throw SceneError("PNG: bad row filter");
                }
                cur[k] = (uint8_t)(cur[k] + add);
            }
            auto sample = [&](uint32_t x, int ch) -> int {  // channel ch of pixel x of the row, as an 8-bit value (palette: the index)
                if (depth == 8) return cur[(size_t)x * channels + ch];
                if (depth == 16) {
                    const int v = (cur[((size_t)x * channels + ch) * 2] << 8) | cur[((size_t)x * channels + ch) * 2 + 1];
                    return (v + 128) / 257;
                }
                const size_t bit = (size_t)x * depth;  // 1, 2, 4 bits: one channel
                const int v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1);
                return colour == 3 ? v : v * 255 / ((1 << depth) - 1);
            };
            for (uint32_t x = 0; x < p.w; x++) {
                uint8_t* o = &img.rgb[(((size_t)p.y0 + (size_t)y * p.dy) * width + p.x0 + (size_t)x * p.dx) * 3];
                if (colour == 3) {
                    const size_t idx = (size_t)sample(x, 0);
                    if (idx * 3 + 2 >= palette.size()) throw SceneError("PNG: palette index out of range");
                    o[0] = palette[idx * 3]; o[1] = palette[idx * 3 + 1]; o[2] = palette[idx * 3 + 2];
                } else if (colour == 0 || colour == 4) {
                    o[0] = o[1] = o[2] = (uint8_t)sample(x, 0);
                } else {
                    o[0] = (uint8_t)sample(x, 0); o[1] = (uint8_t)sample(x, 1); o[2] = (uint8_t)sample(x, 2);
                }
            }
            prev.swap(cur);
        }
    }
    return img;
}

namespace {

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// Windows BMP: uncompressed 24 / 32 bits per pixel and 8-bit palette images, bottom-up or top-down.
ImageRGB8 decode_bmp(const uint8_t* d, size_t size, const std::string& path) {
    if (size < 54) throw SceneError("BMP: truncated header in " + path);
    const uint32_t offset = le32(d + 10), header = le32(d + 14);
    if (header < 40 || 14 + (size_t)header > size) throw SceneError("BMP: unsupported header in " + path);
    const int32_t w = (int32_t)le32(d + 18), h_signed = (int32_t)le32(d + 22);
    const int bits = le16(d + 28);
    const uint32_t compression = le32(d + 30);
    const bool top_down = h_signed < 0;
    const int64_t h = top_down ? -(int64_t)h_signed : h_signed;
    if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > (1ull << 28)) throw SceneError("BMP: bad image size in " + path);
    if (!(bits == 24 || bits == 32 || bits == 8) || !(compression == 0 || (compression == 3 && bits == 32)))
        throw SceneError("BMP: only uncompressed 8 / 24 / 32-bit files are supported: " + path);
    const size_t stride = (((size_t)w * bits + 31) / 32) * 4;
    if ((size_t)offset > size || stride * (size_t)h > size - offset) throw SceneError("BMP: truncated pixel data in " + path);
    const uint8_t* pal = d + 14 + header;
    uint32_t colours = le32(d + 46);
    if (bits == 8) {
        if (colours == 0 || colours > 256) colours = 256;
        if (14 + (size_t)header + (size_t)colours * 4 > size) throw SceneError("BMP: truncated palette in " + path);
    }
    ImageRGB8 img;
    img.width = w;
    img.height = (int)h;
    img.rgb.resize((size_t)w * h * 3);
    for (int64_t y = 0; y < h; y++) {
        const uint8_t* row = d + offset + stride * (size_t)(top_down ? y : h - 1 - y);
        for (int32_t x = 0; x < w; x++) {
            uint8_t* o = &img.rgb[((size_t)y * w + x) * 3];
            if (bits == 8) {
                const uint32_t idx = row[x];
                if (idx >= colours) throw SceneError("BMP: palette index out of range in " + path);
                o[0] = pal[idx * 4 + 2]; o[1] = pal[idx * 4 + 1]; o[2] = pal[idx * 4];
            } else {
                const uint8_t* px = row + (size_t)x * (bits / 8);
                o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
            }
        }
    }
    return img;
}

// PNM: P6 / P5 (binary colour / grey) and P3 / P2 (their ASCII forms), maxval <= 255.
ImageRGB8 decode_pnm(const std::string& data, const std::string& path) {
    const int kind = data[1] - '0';
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
        long v = 0;
        bool any = false;
        while (pos < data.size() && std::isdigit((unsigned char)data[pos])) {
            v = v * 10 + (data[pos++] - '0');
            if (v > (1l << 30)) throw SceneError("PNM: number out of range in " + path);
            any = true;
        }
        if (!any) throw SceneError("PNM: bad header in " + path);
        return (int)v;
    };
    ImageRGB8 img;
    img.width = next_int();
    img.height = next_int();
    int maxval = next_int();
    if (img.width <= 0 || img.height <= 0 || (uint64_t)img.width * (uint64_t)img.height > (1ull << 28)) throw SceneError("PNM: bad image size in " + path);
    if (maxval <= 0 || maxval > 255) throw SceneError("PNM: unsupported maxval in " + path);
    const int channels = (kind == 6 || kind == 3) ? 3 : 1;
    const size_t n = (size_t)img.width * img.height * channels;
    std::vector<uint8_t> samples(n);
    if (kind == 6 || kind == 5) {
        pos++;  // single whitespace after maxval
        if (data.size() < pos + n) throw SceneError("PNM: truncated data in " + path);
        std::memcpy(samples.data(), data.data() + pos, n);
    } else {
        for (size_t k = 0; k < n; k++) {
            const int v = next_int();
            if (v > maxval) throw SceneError("PNM: sample above maxval in " + path);
            samples[k] = (uint8_t)v;
        }
    }
    if (maxval != 255)
        for (auto& b : samples) b = (uint8_t)((int)b * 255 / maxval);
    if (channels == 3) {
        img.rgb.swap(samples);
    } else {
        img.rgb.resize(n * 3);
        for (size_t k = 0; k < n; k++) img.rgb[3 * k] = img.rgb[3 * k + 1] = img.rgb[3 * k + 2] = samples[k];
    }
    return img;
}

}  // namespace

ImageRGB8 load_image_rgb8(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError("cannot open image file " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string data = ss.str();
    const uint8_t* d = (const uint8_t*)data.data();
    // (image::io::Reader::open picks the decoder by the file's extension, src/textures.rs:33; the content's signature is
    // what is looked at here, which accepts every file the reference does)
    if (data.size() >= 2 && d[0] == 0xFF && d[1] == 0xD8) return decode_jpeg(d, data.size());
    if (data.size() >= 8 && d[0] == 0x89 && d[1] == 'P' && d[2] == 'N' && d[3] == 'G') return decode_png(d, data.size());
    if (data.size() >= 2 && d[0] == 'B' && d[1] == 'M') return decode_bmp(d, data.size(), path);
    if (data.size() >= 3 && data[0] == 'P' && (data[1] == '6' || data[1] == '5' || data[1] == '3' || data[1] == '2')) return decode_pnm(data, path);
    throw SceneError("unsupported image format (JPEG, PNG, BMP and PNM are supported): " + path);
}

}  // namespace rt
