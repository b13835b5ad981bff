// Host stand-in for <hip/hip_runtime.h>, only for tests/native/planar_dyn_host.cpp: the product's device header
// csrc/planar_dyn.hpp compiled by g++ with ONE lane per "wavefront" (test infrastructure, not product code).
#pragma once
#include <cmath>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
// v_med3_f32(x, -lim, lim), lim >= 0: the clamp
#define __builtin_amdgcn_fmed3f(x, lo, hi) std::fmin(std::fmax((x), (lo)), (hi))
// a wavefront of one lane
#define __builtin_amdgcn_ballot_w64(p) ((p) ? 1ull : 0ull)
