// fqz_xxh.h — XXH64 on the device: the zstd Content_Checksum (RFC 8878 3.1.1: low 32 bits of XXH64 over the frame's
// content, seed 0).  The reference keeps this checksum on purpose (PERFORMANCE.md E033 "discarded due integrity
// requirements", README.md:87; encoder options compress.go:115-118).
//
// The hash is a serial chain over 32-byte stripes with four independent accumulators, so one frame is hashed by FOUR
// lanes (one accumulator each, 8 bytes per stripe) and a wave hashes 16 frames at once; frames are <= 64 KiB in our own
// payloads (FQZ-H2: one frame per group of four chunks), which is what makes the checksum affordable on a GPU: 2048
// dependent steps per frame instead of ~470 000 for a 15 MB stream.
#pragma once
#include "fqz_device.h"

#define XXP1 0x9E3779B185EBCA87ull
#define XXP2 0xC2B2AE3D27D4EB4Full
#define XXP3 0x165667B19E3779F9ull
#define XXP4 0x85EBCA77C2B2AE63ull
#define XXP5 0x27D4EB2F165667C5ull

__device__ __forceinline__ unsigned long long xx_rotl(unsigned long long x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ unsigned long long xx_round(unsigned long long acc, unsigned long long in) { return xx_rotl(acc + in * XXP2, 31) * XXP1; }
__device__ __forceinline__ unsigned long long xx_merge(unsigned long long acc, unsigned long long v) { return (acc ^ xx_round(0, v)) * XXP1 + XXP4; }
__device__ __forceinline__ unsigned long long xx_read64(const uint8_t *p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t xx_read32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// XXH64(p[0..len), seed 0) by the four lanes q = 0..3 of an aligned lane quad (all four must call it with the same p, len;
// lanes of quads that have nothing to hash pass len = 0).  The result is valid in every lane of the quad.
__device__ __forceinline__ unsigned long long xxh64_quad(const uint8_t *p, uint32_t len, uint32_t lane)
{
    const uint32_t q = lane & 3u, base = lane & ~3u;
    unsigned long long h;
    uint32_t done = 0;
    if (len >= 32) {
        unsigned long long v = q == 0 ? XXP1 + XXP2 : (q == 1 ? XXP2 : (q == 2 ? 0ull : 0ull - XXP1));
        const uint32_t stripes = len >> 5;
        const uint8_t *s = p + 8 * q;
        uint32_t i = 0;
        // sixteen stripes per trip: the kernel is a chain of memory round trips (a handful of waves per CU, every trip waits
        // for its loads), so the loads of 512 bytes per quad are in flight together
        for (; i + 16 <= stripes; i += 16) {
            unsigned long long x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = xx_read64(s + 32 * k);
#pragma unroll
            for (int k = 0; k < 16; k++) v = xx_round(v, x[k]);
            s += 512;
        }
        for (; i + 4 <= stripes; i += 4) {
            const unsigned long long a = xx_read64(s), b = xx_read64(s + 32), c = xx_read64(s + 64), d = xx_read64(s + 96);
            v = xx_round(v, a); v = xx_round(v, b); v = xx_round(v, c); v = xx_round(v, d);
            s += 128;
        }
        for (; i < stripes; i++) { v = xx_round(v, xx_read64(s)); s += 32; }
        done = stripes << 5;
        const unsigned long long v1 = __shfl(v, (int)base, WAVE), v2 = __shfl(v, (int)base + 1, WAVE), v3 = __shfl(v, (int)base + 2, WAVE),
                                 v4 = __shfl(v, (int)base + 3, WAVE);
        h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
        h = xx_merge(h, v1); h = xx_merge(h, v2); h = xx_merge(h, v3); h = xx_merge(h, v4);
    } else h = XXP5;
    h += (unsigned long long)len;
    // the last < 32 bytes: every lane of the quad computes the same value
    const uint8_t *t = p + done;
    uint32_t rem = len - done;
    while (rem >= 8) { h ^= xx_round(0, xx_read64(t)); h = xx_rotl(h, 27) * XXP1 + XXP4; t += 8; rem -= 8; }
    if (rem >= 4) { h ^= (unsigned long long)xx_read32(t) * XXP1; h = xx_rotl(h, 23) * XXP2 + XXP3; t += 4; rem -= 4; }
    while (rem) { h ^= (unsigned long long)(*t++) * XXP5; h = xx_rotl(h, 11) * XXP1; rem--; }
    h ^= h >> 33; h *= XXP2; h ^= h >> 29; h *= XXP3; h ^= h >> 32;
    return h;
}
