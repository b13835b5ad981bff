// spec_fma.hpp -- the fused operation and the reciprocal of the dynamics specs (DESIGN.md sections 2 and 3).
#pragma once
#include <hip/hip_runtime.h>

namespace m3 {

// Where a spec writes mad(a, b, c) the product and the sum are ONE operation with one rounding (IEEE 754
// fusedMultiplyAdd: v_fma_f32 here, fmaf in the oracle) -- every a*b + c of the dynamics.  Half the multiplies and
// adds of the solver and of the kinematics pair up; -ffp-contract=off stays, so nothing else is ever fused (the task
// costs follow torch's unfused arithmetic).
__device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// The specs' reciprocal (planar spec v1.6): a fixed sequence instead of an IEEE division -- the minimax bit-trick seed (5 % off at
// worst) and three Newton steps in residual form, 7 dependent operations where the division's expansion has ~12 and a
// quarter-rate v_rcp_f32.  Correctly rounded for > 99.9 % of inputs and within one unit in the last place for all of them
// (tests/test_dynamics_physics.py); x is positive and normal at every site.  Measured on the planar rollouts
// (profiles/r05/rcp_ab.txt): push -3.3 %, hybrid -3.2 %, the all-19-slot instance -4.3 %.
__device__ __forceinline__ float spec_rcp(float x) {
    float y = __uint_as_float(0x7EF311C7u - __float_as_uint(x));
    float r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    r = mad(-x, y, 1.0f); y = mad(y, r, y);
    return y;
}

}  // namespace m3
