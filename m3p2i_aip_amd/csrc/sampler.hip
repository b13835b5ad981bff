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
#include "noise_stream.hpp"
#include "spline_fit.hpp"

#include <hipcub/hipcub.hpp>

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

// ---- the knots themselves on the device (optional: m3_set_noise_halton) ------------------------
// Gaussian Halton values of samples k0 .. k0 + n: van der Corput radical inverse of the index k + 1 in the
// c-th prime (mppi_utils.py:69-96, the in-tree use_ghalton=False branch), accumulated in binary32 digit by
// digit from the least significant one with the scale 1 / base^d carried in binary64 and rounded per digit --
// the arithmetic of sampling.radical_inverse, bit for bit -- then sqrt(2) * erfinv(2u - 1) (mppi_utils.py:
// 99-104).  erfinv is the rational approximation + two Newton steps torch's CPU kernel uses, on the device's
// erff / expf / logf: those differ from glibc's in the last ulp, so the Gaussian values agree with the host
// sampler to ~1e-6 relative, not bit for bit (which is why the planner's default keeps the host knots,
// pinned by golden G8; MPPIConfig.device_knots opts in).
__device__ __forceinline__ float dev_erfinv(float y) {
    const float a[4] = {0.886226899f, -1.645349621f, 0.914624893f, -0.140543331f};
    const float b[4] = {-2.118377725f, 1.442710462f, -0.329097515f, 0.012229801f};
    const float c[4] = {-1.970840454f, -1.624906493f, 3.429567803f, 1.641345311f};
    const float d[2] = {3.543889200f, 1.637067800f};
    const float ya = fabsf(y);
    if (ya > 1.0f) return __builtin_nanf("");
    if (ya == 1.0f) return copysignf(__builtin_inff(), y);
    float x;
    if (ya <= 0.7f) {
        const float z = y * y;
        const float num = ((a[3] * z + a[2]) * z + a[1]) * z + a[0];
        const float dem = (((b[3] * z + b[2]) * z + b[1]) * z + b[0]) * z + 1.0f;
        x = y * num / dem;
    } else {
        const float z = sqrtf(-logf((1.0f - ya) / 2.0f));
        const float num = ((c[3] * z + c[2]) * z + c[1]) * z + c[0];
        const float dem = (d[1] * z + d[0]) * z + 1.0f;
        x = copysignf(num, y) / dem;
    }
    const float two_over_sqrt_pi = 1.1283791670955126f;
    x = x - (erff(x) - y) / (two_over_sqrt_pi * expf(-x * x));
    x = x - (erff(x) - y) / (two_over_sqrt_pi * expf(-x * x));
    return x;
}

// perm == nullptr: the plain van der Corput radical inverse (the reference's in-tree branch, mppi_utils.py:69-87).
// Otherwise the GENERALIZED Halton sequence: digit d of base primes[c] is replaced by perm[perm_off[c] + d]
// (pi_b(0) = 0, so the infinitely many leading zero digits still contribute nothing) -- the structure of the
// branch the reference's planner actually takes (ghalton.GeneralizedHalton, mppi_utils.py:89-95) with a published,
// formula-defined permutation set (m3_api.hip: Faure 1992) in place of ghalton's unpinnable EA_PERMS table.
__global__ void k_halton_knots(float* __restrict__ knots /*[n][ncol]*/, int k0, int n, int ncol, const int* __restrict__ primes,
                               const int* __restrict__ perm, const int* __restrict__ perm_off) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n * ncol) return;
    const int c = o % ncol;
    unsigned rem = (unsigned)(k0 + o / ncol) + 1u;      // Halton indices start at 1
    const unsigned base = (unsigned)primes[c];
    const int* pc = perm ? perm + perm_off[c] : nullptr;
    float acc = 0.0f;
    double scale = 1.0;
    while (rem > 0u) {
        scale /= (double)base;
        const unsigned d = rem % base;
        acc += (float)scale * (float)(pc ? (unsigned)pc[d] : d);
        rem /= base;
    }
    const float u = 2.0f * acc - 1.0f;
    knots[o] = 1.41421356237309515f * dev_erfinv(u);
}

void launch_halton_knots(float* knots, int k0, int n, int ncol, const int* primes_dev, const int* perm_dev,
                         const int* perm_off_dev, hipStream_t s) {
    const int total = n * ncol;
    hipLaunchKernelGGL(k_halton_knots, dim3((total + 255) / 256), dim3(256), 0, s, knots, k0, n, ncol, primes_dev, perm_dev,
                       perm_off_dev);
}

// ---- wavefront order of the samples (point_env rollout) ---------------------------------
// The rollout kernel picks, per substep and per WAVE, the leanest dynamics instance that covers
// every lane's nearby pairs (planar_dyn.hpp), so a wave costs the UNION of its 64 samples' contact
// situations.  Consecutive Halton indices are spread over the whole action space by construction
// -- every wave got a sample that meets the dyn-obs, one that meets the box, one at the obstacle.
// The noise is fixed after init, so the samples are assigned to wavefronts by a sort instead:
// key = position of the sample's path centroid relative to the mean path (sum_t cumsum_t(scale *
// delta)) projected on the axis along which the scene's objects (box, dyn-obs, obstacle) are spread
// (principal axis of their positions; in the reference's point_env they sit in one row).  Samples of a wave
// then meet the same object, or none.  Lane placement cannot change a sample's numbers (all
// arithmetic is lane-local; tests/test_full_size_properties.py).  Measured (bench.py, order off ->
// on): push K=2000 0.191 -> 0.171 ms, hybrid K=4000 0.231 -> 0.210, north-star K=10000 0.208 -> 0.190.
// Two ways to apply the order: as an indirection (lane slot -> sample; noise rows gathered once so
// the loads stay coalesced, but the stores into the sample-indexed outputs become scattered 8-16 B
// pieces: HBM write traffic 2.8x), or -- for generated noise, whose row labels carry no meaning -- by
// RELABELLING the samples once (m3_relabel_samples: rows permuted in place, samples with a role of
// their own fixed): index order is then wavefront order, everything coalesced, another -5 %.
// Keys tried and measured worse: direction (angle) of the centroid, of the mid-horizon and end
// displacement, radius, Morton cells, sectors x radius (0 .. -5 %); the mid-horizon displacement
// on the same axis is equivalent.
__global__ void k_order_keys(const float* __restrict__ noise /*[T][Kl][nu]*/, int Kl, int T, int nu, float s0,
                             float s1, int half_local, OrderScene os, float* __restrict__ keys,
                             int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    float bx = os.bx, by = os.by, dx = os.dx, dy = os.dy;
    if (os.sim_root) {
        bx = os.sim_root[(size_t)os.sim_box * 13 + 0]; by = os.sim_root[(size_t)os.sim_box * 13 + 1];
        dx = os.sim_root[(size_t)os.sim_dyn * 13 + 0]; dy = os.sim_root[(size_t)os.sim_dyn * 13 + 1];
    }
    // principal axis of the three object positions
    const float mx = (bx + dx + os.ox) * (1.0f / 3.0f), my = (by + dy + os.oy) * (1.0f / 3.0f);
    const float cxx = (bx - mx) * (bx - mx) + (dx - mx) * (dx - mx) + (os.ox - mx) * (os.ox - mx);
    const float cyy = (by - my) * (by - my) + (dy - my) * (dy - my) + (os.oy - my) * (os.oy - my);
    const float cxy = (bx - mx) * (by - my) + (dx - mx) * (dy - my) + (os.ox - mx) * (os.oy - my);
    const float th = 0.5f * atan2f(2.0f * cxy, cxx - cyy);
    const float ex = cosf(th), ey = sinf(th);
    float cx = 0.0f, cy = 0.0f, sx = 0.0f, sy = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float* d = noise + ((size_t)t * Kl + i) * nu;
        cx += s0 * d[0]; cy += s1 * d[1];
        sx += cx; sy += cy;
    }
    // the two modes of a multi-modal planner follow different means: keep them in separate waves
    keys[i] = (sx * ex + sy * ey) + (i >= half_local ? 1e9f : 0.0f);
    idx[i] = i;
}

// noise rows gathered into wavefront order once, so that the rollout's noise loads stay coalesced
// (its stores remain scattered: the outputs are indexed by sample for everything downstream)
__global__ void k_gather_noise(const float* __restrict__ noise, const int* __restrict__ order,
                               float* __restrict__ sorted, int Kl, int T, int nu) {
    const size_t n = (size_t)T * Kl * nu;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(o % nu);
        const size_t r = o / nu;
        const int slot = (int)(r % Kl);
        const int t = (int)(r / Kl);
        sorted[o] = noise[((size_t)t * Kl + order[slot]) * nu + j];
    }
}

// relabelling: samples with a role of their own (k = 0 and K/2 carry the best trajectories, K - 1 is
// the null / zero-noise sample) keep their index: each is swapped back to its own slot
__global__ __launch_bounds__(256) void k_fix_specials(int* __restrict__ order, int Kl, int s0, int s1, int s2) {
    __shared__ int s_pos;
    const int sp[3] = {s0, s1, s2};
    for (int q = 0; q < 3; ++q) {
        const int sidx = sp[q];
        if (sidx < 0 || sidx >= Kl) continue;   // (uniform)
        if (threadIdx.x == 0) s_pos = -1;
        __syncthreads();
        for (int p = threadIdx.x; p < Kl; p += blockDim.x)
            if (order[p] == sidx) s_pos = p;
        __syncthreads();
        if (threadIdx.x == 0 && s_pos >= 0) {
            const int other = order[sidx];
            order[sidx] = sidx;
            order[s_pos] = other;
        }
        __syncthreads();
    }
}
// pending suction forces [4][Kl] follow their samples
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ order, float* __restrict__ dst,
                              int Kl, int rows) {
    const int n = rows * Kl;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x)
        dst[o] = src[(o / Kl) * Kl + order[o % Kl]];
}

size_t wave_order_temp_bytes(int Kl) {
    size_t n = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, n, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, Kl);
    return n;
}

// scratch: keys_in [Kl] | keys_out [Kl] | idx_in [Kl] (floats / ints), then the radix sort's own storage
hipError_t launch_wave_order(const float* noise, int Kl, int T, int nu, float s0, float s1, int half_local,
                             const OrderScene& os, void* scratch, size_t temp_bytes, int* order, float* noise_sorted,
                             const int* specials, hipStream_t s) {
    float* keys_in = (float*)scratch;
    float* keys_out = keys_in + Kl;
    int* idx_in = (int*)(keys_out + Kl);
    void* temp = (void*)(idx_in + Kl);
    hipLaunchKernelGGL(k_order_keys, dim3((Kl + 255) / 256), dim3(256), 0, s, noise, Kl, T, nu, s0, s1, half_local,
                       os, keys_in, idx_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, order, Kl, 0, 32, s);
    if (e != hipSuccess) return e;
    if (specials) hipLaunchKernelGGL(k_fix_specials, dim3(1), dim3(256), 0, s, order, Kl, specials[0], specials[1], specials[2]);
    const size_t n = (size_t)T * Kl * nu;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gather_noise, dim3(blocks), dim3(256), 0, s, noise, order, noise_sorted, Kl, T, nu);
    return hipGetLastError();
}

void launch_gather_rows(const float* src, const int* order, float* dst, int Kl, int rows, hipStream_t s) {
    int blocks = (rows * Kl + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, s, src, order, dst, Kl, rows);
}

// ---- m3_sample_noise: the in-kernel stream of the rollout kernels, materialised (STEP mode) ----
__global__ __launch_bounds__(256) void k_sample_noise(const RolloutArgs a, float* __restrict__ out /* [T][Kl][nu] */) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.T * a.Kl) return;
    const int t = o / a.Kl, i = o % a.Kl, nu = a.nu;
    float z[M3_MAX_NU + 1];
    for (int p = 0; p < (nu + 1) / 2; ++p)
        gauss_pair(a.seed, a.call, (unsigned)(a.k0 + i), (unsigned)t, (unsigned)p, z[2 * p], z[2 * p + 1]);
    for (int j = 0; j < nu; ++j) {   // as rollout_point_kernel.hpp / rollout_panda.hip
        float acc;
        if (a.full_sigma) {
            acc = a.noise_mats[j * nu + 0] * z[0];
            for (int q = 1; q <= j; ++q) acc = acc + a.noise_mats[j * nu + q] * z[q];
        } else acc = z[j] * a.scale_tril[j];
        out[(size_t)o * nu + j] = a.noise_mu[j] + acc;
    }
}
void launch_sample_noise(const RolloutArgs& a, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sample_noise, dim3((a.T * a.Kl + 255) / 256), dim3(256), 0, s, a, out);
}

}  // namespace m3
