// TEST INFRASTRUCTURE - CPU oracle.  Not part of the product; see oracle/README.md.
//
// Counter-based random streams shared (as a specification, not as code) by the oracle and
// the GPU renderer, so that both can walk the *same* sample paths.
//
// The reference draws every random number from rand 0.8.0's thread_rng (ChaCha12, OS
// seeded; draw sites listed in SURVEY.md section 8a row a10), so it is not reproducible
// and only the *distributions* of its draws are part of its behaviour:
//   gen_range(a..b)  -> uniform f64 in [a, b)      (src/math.rs:238-240,251; src/main.rs:132,134;
//                                                   src/materials.rs:88; src/objects.rs:426)
//   gen_range(a..=b) -> uniform f64 in [a, b]      (src/camera.rs:73)
// This file pins a concrete generator for those draws: Philox4x32-10 (Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3", SC'11), keyed as
//   key     = { seed[31:0], seed[63:32] ^ domain }
//   counter = { pixel, sample, bounce, block }
// where `domain` is 0 for the camera stream, 1 for the material stream and
// 0x80000000|h for the volume (constant medium) stream of medium-instance hash h.  Each
// 128-bit block yields two doubles: u = (x >> 11) * 2^-53 from x = r0|r1<<32, r2|r3<<32.
#pragma once
#include <cstdint>

namespace orc {

inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum : uint32_t { DOMAIN_CAMERA = 0u, DOMAIN_MATERIAL = 1u, DOMAIN_MEDIUM = 0x80000000u };

inline uint32_t hash_combine(uint32_t a, uint32_t b) {
    return a ^ (b + 0x9E3779B9u + (a << 6) + (a >> 2));
}

struct RngStream {
    uint32_t key[2];
    uint32_t ctr[4];
    double buf[2];
    int have = 0;
    RngStream() : key{0, 0}, ctr{0, 0, 0, 0}, buf{0, 0} {}
    RngStream(uint64_t seed, uint32_t domain, uint32_t pixel, uint32_t sample, uint32_t bounce) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32) ^ domain;
        ctr[0] = pixel; ctr[1] = sample; ctr[2] = bounce; ctr[3] = 0;
        buf[0] = buf[1] = 0;
        have = 0;
    }
    double next() {  // uniform in [0, 1)
        if (have == 0) {
            uint32_t r[4];
            philox4x32_10(ctr, key, r);
            ctr[3]++;
            uint64_t a = (uint64_t)r[0] | ((uint64_t)r[1] << 32);
            uint64_t b = (uint64_t)r[2] | ((uint64_t)r[3] << 32);
            buf[0] = (double)(a >> 11) * 0x1.0p-53;
            buf[1] = (double)(b >> 11) * 0x1.0p-53;
            have = 2;
        }
        double v = buf[2 - have];
        have--;
        return v;
    }
    double range(double lo, double hi) { return lo + (hi - lo) * next(); }
};

}  // namespace orc
