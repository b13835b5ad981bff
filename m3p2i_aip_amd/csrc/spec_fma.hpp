// spec_fma.hpp -- the one fused operation of the dynamics specs (DESIGN.md sections 2 and 3).
#pragma once
#include <hip/hip_runtime.h>

namespace m3 {

// Where a spec writes mad(a, b, c) the product and the sum are ONE operation with one rounding (IEEE 754
// fusedMultiplyAdd: v_fma_f32 here, fmaf in the oracle) -- every a*b + c of the dynamics.  Half the multiplies and
// adds of the solver and of the kinematics pair up; -ffp-contract=off stays, so nothing else is ever fused (the task
// costs follow torch's unfused arithmetic).
__device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

}  // namespace m3
