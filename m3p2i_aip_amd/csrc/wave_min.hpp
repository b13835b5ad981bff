// Minima of a rollout workgroup's trajectory costs (all / first half / second half of the samples), left behind by
// the rollout kernels for the multi-modal update with K > 8192: the beta ladder needs the three global minima
// before its first exp, and a launch of its own for them (k_mins) cost ~8 us + a launch gap on the update's critical
// path.  A rollout workgroup is ONE wavefront whose idle lanes have left, so the reduction goes through three LDS
// cells (ds_min_u32 on order-preserving keys: at most 64 serialised operations per cell, once per rollout) rather
// than DPP steps that would read the registers of lanes that no longer execute.  Plain stores, one row per
// workgroup: no atomics on a shared address (agent-scope atomics on ONE address retire at ~0.1-0.3 us each).
#pragma once
#include <hip/hip_runtime.h>

namespace m3 {

__device__ __forceinline__ unsigned wm_f2ord(float f) {
    const unsigned u = (unsigned)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float wm_ord2f(unsigned u) {
    return __int_as_float((int)((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u));
}

// every lane that is still running calls this once (lane 0 of a rollout workgroup always is); `mine` = the lane's
// cost counts (false: a shadow lane)
__device__ __forceinline__ void wave_min_store(float* dst /* [gridDim.x][3] */, float v, bool first_half, bool mine) {
    __shared__ unsigned s_wm[3];
    if (threadIdx.x == 0) { s_wm[0] = 0xff800000u; s_wm[1] = 0xff800000u; s_wm[2] = 0xff800000u; }   // +inf
    __syncthreads();
    if (mine) {
        const unsigned key = wm_f2ord(v);
        atomicMin(&s_wm[0], key);
        atomicMin(&s_wm[first_half ? 1 : 2], key);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dst[blockIdx.x * 3 + 0] = wm_ord2f(s_wm[0]);
        dst[blockIdx.x * 3 + 1] = wm_ord2f(s_wm[1]);
        dst[blockIdx.x * 3 + 2] = wm_ord2f(s_wm[2]);
    }
}

}  // namespace m3
