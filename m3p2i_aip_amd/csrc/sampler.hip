// sampler.hip -- init-time noise sampler on the device.
//
// The reference builds its Halton-spline noise once per planner (mppi.py:458-483 ->
// mppi_utils.py generate_gaussian_halton_samples + bspline): K x nu series of n_knots = T // 4
// Gaussian Halton values, each fitted with a FITPACK smoothing spline (k = 2, s = 0.5) and evaluated
// at T points -- K*nu independent little least-squares problems, ~30 us each through scipy on the
// host (~4 s at K = 64000).  Here: one thread per series runs the same algorithm
// (spline_fit.hpp, bit-identical to scipy's FITPACK) and writes its T values straight into the
// time-major noise buffer [T][K_local][nu] the rollout kernel reads.
#include "m3_internal.hpp"
#include "spline_fit.hpp"

namespace m3 {

__global__ __launch_bounds__(64) void k_spline_noise(const float* __restrict__ knots /*[Kl][nu][n_knots]*/,
                                                     float* __restrict__ noise /*[T][Kl][nu]*/, int Kl, int nu,
                                                     int n_knots, int T, int degree, double smoothing) {
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;   // series = (sample, control dim)
    if (sidx >= Kl * nu) return;
    double y[SF_MAX_M];
    const float* src = knots + (size_t)sidx * n_knots;
    for (int i = 0; i < n_knots; ++i) y[i] = (double)src[i];
    // series (k, j) -> noise[t][k][j]: element stride Kl*nu, offset k*nu + j == sidx
    spline_fit_eval<float>(y, n_knots, degree, smoothing, T, noise + sidx, Kl * nu);
}

void launch_spline_noise(const float* knots, float* noise, int Kl, int nu, int n_knots, int T, int degree,
                         double smoothing, hipStream_t s) {
    const int n = Kl * nu;
    hipLaunchKernelGGL(k_spline_noise, dim3((n + 63) / 64), dim3(64), 0, s, knots, noise, Kl, nu, n_knots, T,
                       degree, smoothing);
}

}  // namespace m3
