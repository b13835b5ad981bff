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
// v_med3_f32(x, -lim, lim), lim >= 0: the clamp, written as the spec writes it (compare and assign: the sign of a zero
// bound is the bound's; fmin / fmax leave that to the implementation)
static inline float m3_host_med3(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }
#define __builtin_amdgcn_fmed3f(x, lo, hi) m3_host_med3((x), (lo), (hi))
// a wavefront of one lane
#define __builtin_amdgcn_ballot_w64(p) ((p) ? 1ull : 0ull)
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
// lane index: the host build instantiates the one-lane-per-sample forms only (LPS = 1), which never read it
static const struct { unsigned x, y, z; } threadIdx = {0u, 0u, 0u};
// cross-lane operations of the sixteen-lanes-per-sample forms (never instantiated on the host)
int __builtin_amdgcn_update_dpp(int, int, int, int, int, bool);
int __builtin_amdgcn_ds_bpermute(int, int);
#define __builtin_amdgcn_readfirstlane(x) (x)      // a wavefront of one lane
