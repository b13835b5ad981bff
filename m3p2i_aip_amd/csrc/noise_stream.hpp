// noise_stream.hpp -- the in-kernel random stream of sampling_method = 'random' / mppi_mode = 'simple'
// (mppi.py:129-131, :340, :481 draw from torch's global generator, which cannot be reproduced; the build
// defines its own counter-based stream so that results are shard-invariant: DESIGN.md section 5).
#pragma once
#include <hip/hip_runtime.h>

namespace m3 {

// ---- counter-based noise stream (spec: DESIGN.md "Noise stream"; mirrors the oracle) ----
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& x) {
    unsigned long long z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned rotl32(unsigned x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ unsigned xoshiro128pp(unsigned (&s)[4]) {
    const unsigned result = rotl32(s[0] + s[3], 7) + s[0];
    const unsigned t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}
// standard-normal pair for (seed, call, k, t, pair)
__device__ __forceinline__ void gauss_pair(unsigned long long seed, unsigned call, unsigned k,
                                           unsigned t, unsigned pair, float& z0, float& z1) {
    unsigned long long x = seed ^ (0xD1B54A32D192ED03ULL * (unsigned long long)(call + 1u));
    x ^= ((unsigned long long)k << 32) | ((unsigned long long)t << 8) | (unsigned long long)pair;
    const unsigned long long a = splitmix64(x), b = splitmix64(x);
    unsigned s[4] = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    const unsigned r0 = xoshiro128pp(s), r1 = xoshiro128pp(s);
    const float u0 = ((float)(r0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = (float)(r1 >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u0));
    const float ang = 6.28318530717958647692f * u1;
    z0 = rad * cosf(ang);
    z1 = rad * sinf(ang);
}

}  // namespace m3
