// update.hip -- importance-weight update of MPPI / M3P2I as three small kernels (gfx950).
//
//   k_weights : softmin weights with wavefront (DPP shuffle) + LDS reductions
//               _exp_util                      mppi.py:430-456
//               update_infinite_beta           m3p2i.py:24-44   (3 searches run in lock-step)
//               _multi_modal_exp_util          m3p2i.py:46-64
//               argmax / top-k                 mppi.py:493, 248; m3p2i.py:75-76
//               simple-mode weights            mppi.py:225-229
//   k_wsum    : weighted action sums + best / top-trajectory row gathers (per time step)
//               mppi.py:497-498, 252-254; m3p2i.py:77-83
//   k_finalize: mean update, per-mode means, simple-mode U update, Savitzky-Golay
//               mppi.py:502-503, 231, 245, 257-263; m3p2i.py:86-87
//
// The reference runs each beta-search pass as exp + sum kernels and a host sync
// (10-25 passes x 3 searches per command); here the searches stay on the device in one
// workgroup.  No MFMA: there is no dense contraction in this path (K x T*nu weighted sums
// are K-long dot products against ONE weight vector -> bandwidth-bound reductions).
#include "m3_internal.hpp"

namespace m3 {

constexpr int WT = 1024;  // threads of k_weights (16 wavefronts)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reductions of up to 3 values at once; result broadcast to every thread
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* lds /* >= N*16 */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = wave_sum(v[n]);
    __syncthreads();
    if (lane == 0)
        for (int n = 0; n < N; ++n) lds[n * 16 + wv] = v[n];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float s = 0.0f;
        for (int i = 0; i < nw; ++i) s += lds[n * 16 + i];
        v[n] = s;
    }
}
template <int N>
__device__ __forceinline__ void block_min(float (&v)[N], float* lds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = wave_min(v[n]);
    __syncthreads();
    if (lane == 0)
        for (int n = 0; n < N; ++n) lds[n * 16 + wv] = v[n];
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float s = lds[n * 16];
        for (int i = 1; i < nw; ++i) s = fminf(s, lds[n * 16 + i]);
        v[n] = s;
    }
}

// lexicographic (value, index) argmin over the block; "greater than (pv,pi)" filter gives
// the next-smallest element each round (no exclusion list).
struct VI { float v; int i; };
__device__ __forceinline__ bool vi_less(float av, int ai, float bv, int bi) {
    return (av < bv) || (av == bv && ai < bi);
}
__device__ __forceinline__ VI block_argmin(VI x, VI* lds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(x.v, o, 64);
        const int oi = __shfl_xor(x.i, o, 64);
        if (vi_less(ov, oi, x.v, x.i)) { x.v = ov; x.i = oi; }
    }
    __syncthreads();
    if (lane == 0) lds[wv] = x;
    __syncthreads();
    VI r = lds[0];
    for (int i = 1; i < nw; ++i)
        if (vi_less(lds[i].v, lds[i].i, r.v, r.i)) r = lds[i];
    return r;
}

// grid = 2 workgroups: 0 -> weights/info, 1 -> top-k (independent, runs concurrently)
__global__ __launch_bounds__(WT) void k_weights(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    __shared__ float s_beta[3];
    __shared__ int s_done[3];
    const int Kg = a.Kg, half = Kg / 2;
    const int tid = threadIdx.x;
    const float* J = a.Jall;
    const float INF = __builtin_inff();

    if (blockIdx.x == 1) {
        // top-k weights == k smallest trajectory costs (weights are monotone in J);
        // ties resolved towards the lower sample index.
        float pv = -INF;
        int pi = -1;
        const int n = (Kg < M3_TOPK) ? Kg : M3_TOPK;
        for (int r = 0; r < M3_TOPK; ++r) {
            if (r >= n) { if (tid == 0) a.top_idx[r] = a.top_idx[n - 1]; continue; }
            VI best = {INF, 0x7fffffff};
            for (int k = tid; k < Kg; k += WT) {
                const float v = J[k];
                if (vi_less(pv, pi, v, k) && vi_less(v, k, best.v, best.i)) { best.v = v; best.i = k; }
            }
            best = block_argmin(best, redvi);
            pv = best.v; pi = best.i;
            if (tid == 0) a.top_idx[r] = best.i;
        }
        return;
    }

    // ---- minima (all, first half, second half) ----
    float mn[3] = {INF, INF, INF};
    for (int k = tid; k < Kg; k += WT) {
        const float v = J[k];
        mn[0] = fminf(mn[0], v);
        if (k < half) mn[1] = fminf(mn[1], v); else mn[2] = fminf(mn[2], v);
    }
    block_min<3>(mn, red);

    float beta[3], eta[3];
    int iters[3] = {1, 1, 1};
    if (!a.multi_modal || a.mode_simple) {
        // single softmin: beta persists in info (panda adapts it), simple mode uses lambda
        const float b = a.mode_simple ? a.lambda_ : a.info->beta;
        float e[1] = {0.0f};
        const float nib = -1.0f / b;
        for (int k = tid; k < Kg; k += WT) e[0] += expf(nib * (J[k] - mn[0]));
        block_sum<1>(e, red);
        beta[0] = b; eta[0] = e[0];
        beta[1] = beta[2] = 1.0f; eta[1] = eta[2] = 0.0f;
    } else {
        // three on-the-fly beta searches in lock-step; each restarts at beta = 1
        // (beta_1/beta_2/beta are never written back: m3p2i.py:58-60)
        if (tid < 3) { s_beta[tid] = 1.0f; s_done[tid] = 0; }
        __syncthreads();
        eta[0] = eta[1] = eta[2] = 0.0f;
        for (int pass = 0; pass < 1000; ++pass) {
            const float b0 = s_beta[0], b1 = s_beta[1], b2 = s_beta[2];
            const int d0 = s_done[0], d1 = s_done[1], d2 = s_done[2];
            if (d0 && d1 && d2) break;
            float e[3] = {0.0f, 0.0f, 0.0f};
            const float n0 = -1.0f / b0, n1 = -1.0f / b1, n2 = -1.0f / b2;
            for (int k = tid; k < Kg; k += WT) {
                const float v = J[k];
                if (!d0) e[0] += expf(n0 * (v - mn[0]));
                if (k < half) { if (!d1) e[1] += expf(n1 * (v - mn[1])); }
                else { if (!d2) e[2] += expf(n2 * (v - mn[2])); }
            }
            block_sum<3>(e, red);
            __syncthreads();
            if (tid < 3 && !s_done[tid]) {
                const float et = e[tid];
                if (et > 10.0f) s_beta[tid] = s_beta[tid] * 0.9f;
                else if (et < 3.0f) s_beta[tid] = s_beta[tid] * 1.2f;
                else s_done[tid] = 1;
            }
            if (!d0) { eta[0] = e[0]; iters[0] = pass + 1; }
            if (!d1) { eta[1] = e[1]; iters[1] = pass + 1; }
            if (!d2) { eta[2] = e[2]; iters[2] = pass + 1; }
            __syncthreads();
        }
        beta[0] = s_beta[0]; beta[1] = s_beta[1]; beta[2] = s_beta[2];
        // a search that stopped by "found" keeps the beta that satisfied the bounds; one cut
        // off by the pass cap keeps its last evaluated beta -- recompute below is consistent
    }

    // ---- normalised weights, half sums, argmax ----
    // NOTE: when a search ended with `found`, s_beta was not changed after the last eta, so
    // exp(-(J-min)/beta) recomputed here equals the reference's returned exp_.
    const float i0 = 1.0f / eta[0], n0 = -1.0f / beta[0];
    float hs[2] = {0.0f, 0.0f};
    VI b0 = {INF, 0x7fffffff}, b1 = {INF, 0x7fffffff}, b2 = {INF, 0x7fffffff};
    for (int k = tid; k < Kg; k += WT) {
        const float v = J[k];
        const float wk = i0 * expf(n0 * (v - mn[0]));
        a.w[k] = wk;
        if (k < half) hs[0] += wk; else hs[1] += wk;
        // argmax of the weights, first index on ties (torch.argmax on CPU): key = -w
        if (vi_less(-wk, k, b0.v, b0.i)) { b0.v = -wk; b0.i = k; }
        if (a.multi_modal && !a.mode_simple) {
            if (k < half) {
                const float w1k = (1.0f / eta[1]) * expf((-1.0f / beta[1]) * (v - mn[1]));
                a.w1[k] = w1k;
                if (vi_less(-w1k, k, b1.v, b1.i)) { b1.v = -w1k; b1.i = k; }
            } else {
                const float w2k = (1.0f / eta[2]) * expf((-1.0f / beta[2]) * (v - mn[2]));
                a.w2[k - half] = w2k;
                if (vi_less(-w2k, k, b2.v, b2.i)) { b2.v = -w2k; b2.i = k; }
            }
        }
    }
    block_sum<2>(hs, red);
    b0 = block_argmin(b0, redvi);
    if (a.multi_modal && !a.mode_simple) {
        b1 = block_argmin(b1, redvi);
        b2 = block_argmin(b2, redvi);
    }
    if (tid == 0) {
        m3_info* f = a.info;
        f->eta = eta[0]; f->eta_1 = eta[1]; f->eta_2 = eta[2];
        f->iters = iters[0]; f->iters_1 = iters[1]; f->iters_2 = iters[2];
        f->best_idx = b0.i;
        f->best_idx_1 = (a.multi_modal && !a.mode_simple) ? b1.i : -1;
        f->best_idx_2 = (a.multi_modal && !a.mode_simple) ? b2.i : -1;
        f->wsum_push = hs[0]; f->wsum_pull = hs[1];
        f->pull_preference = hs[1] > hs[0];
        float nb = beta[0];
        if (!a.multi_modal && !a.mode_simple && a.env_type == M3_ENV_PANDA) {  // mppi.py:446-454
            if (eta[0] > 20.0f) nb = nb * 0.9f;
            else if (eta[0] < 10.0f) nb = nb * 1.2f;
        }
        if (!a.multi_modal && !a.mode_simple) f->beta = nb;
        f->beta_1 = beta[1]; f->beta_2 = beta[2];
        if (a.multi_modal && !a.mode_simple) f->beta = beta[0];
    }
}

void launch_weights(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_weights, dim3(2), dim3(WT), 0, s, a);
}

// one workgroup per time step t: sum_k w_k * actions[t][k][:] over the local shard, for the
// global weights and (multi-modal) the two per-mode weight sets; plus row gathers.
constexpr int ST = 256;
__global__ __launch_bounds__(ST) void k_wsum(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int Kl = a.Kl, k0 = a.k0, nu = a.nu, T = a.T, half = a.Kg / 2;
    const bool multi = a.multi_modal && !a.mode_simple;
    const float* act = a.actions + (size_t)t * Kl * nu;
    for (int j0 = 0; j0 < nu; ++j0) {
        float acc[3] = {0.0f, 0.0f, 0.0f};
        for (int i = tid; i < Kl; i += ST) {
            const int k = k0 + i;
            const float av = act[(size_t)i * nu + j0];
            acc[0] += a.w[k] * av;
            if (multi) {
                if (k < half) acc[1] += a.w1[k] * av;
                else acc[2] += a.w2[k - half] * av;
            }
        }
        block_sum<3>(acc, red);
        if (tid == 0) {
            a.reduce[reduce_off_psum(0, T, nu) + t * nu + j0] = acc[0];
            a.reduce[reduce_off_psum(1, T, nu) + t * nu + j0] = acc[1];
            a.reduce[reduce_off_psum(2, T, nu) + t * nu + j0] = acc[2];
        }
        __syncthreads();
    }
    // best rows (zero unless the owning rank) and top-k trajectories
    if (tid < 3 * nu) {
        const int which = tid / nu, j = tid % nu;
        const int gi = (which == 0) ? a.info->best_idx : (which == 1 ? a.info->best_idx_1 : a.info->best_idx_2);
        float v = 0.0f;
        const int li = gi - k0;
        if (gi >= 0 && li >= 0 && li < Kl) v = act[(size_t)li * nu + j];
        a.reduce[reduce_off_best(which, T, nu) + t * nu + j] = v;
    }
    if (tid >= 64 && tid < 64 + M3_TOPK * 2) {
        const int r = (tid - 64) >> 1, c = (tid - 64) & 1;
        const int li = a.top_idx[r] - k0;
        float v = 0.0f;
        if (li >= 0 && li < Kl) v = a.states[((size_t)t * Kl + li) * 4 + (c ? 2 : 0)];  // [0, 2]: mppi.py:253
        a.reduce[reduce_off_top(T, nu) + (r * T + t) * 2 + c] = v;
    }
}
void launch_wsum(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_wsum, dim3(a.T), dim3(ST), 0, s, a);
}

// Savitzky-Golay(9, 2, 'interp') as a fixed linear map: value at window position p of the
// quadratic least-squares fit over 9 samples (x = -4..4): c_p[i] = ca + cb*xp + cc*xp^2
__device__ __forceinline__ float sg_coef(int p, int i) {
    const float S0 = 9.0f, S2 = 60.0f, S4 = 708.0f, det = S0 * S4 - S2 * S2;
    const float xp = (float)(p - 4), xi = (float)(i - 4);
    const float ca = (S4 - S2 * xi * xi) / det;
    const float cb = xi / S2;
    const float cc = (S0 * xi * xi - S2) / det;
    return ca + cb * xp + cc * xp * xp;
}

__global__ __launch_bounds__(256) void k_finalize(const UpdateArgs a) {
    extern __shared__ float sm[];  // [T*nu] new plan
    const int T = a.T, nu = a.nu, n = T * nu, tid = threadIdx.x;
    const bool multi = a.multi_modal && !a.mode_simple;
    const float* ps = a.reduce + reduce_off_psum(0, T, nu);
    const float wtot = a.info->wsum_push + a.info->wsum_pull;
    for (int o = tid; o < n; o += blockDim.x) {
        const int t = o / nu, j = o % nu;
        float nv;
        if (a.mode_simple) {
            const int ts = (t + 1 == T) ? 0 : t + 1;  // rolled U
            const float u = a.mean[ts * nu + j];
            nv = u + (ps[o] - u * wtot);               // U += sum_k w_k (a_k - U): mppi.py:231
        } else {
            const int ts = (t + 1 < T) ? t + 1 : T - 1;  // shifted mean
            nv = (1.0f - a.step_size_mean) * a.mean[ts * nu + j] + a.step_size_mean * ps[o];
        }
        sm[o] = nv;
    }
    __syncthreads();
    for (int o = tid; o < n; o += blockDim.x) {
        a.mean[o] = sm[o];
        if (multi) {
            a.mean1[o] = a.reduce[reduce_off_psum(1, T, nu) + o];  // m3p2i.py:82-83
            a.mean2[o] = a.reduce[reduce_off_psum(2, T, nu) + o];
            a.best1[o] = a.reduce[reduce_off_best(1, T, nu) + o];  // m3p2i.py:77-78
            a.best2[o] = a.reduce[reduce_off_best(2, T, nu) + o];
        } else if (!a.mode_simple) {
            a.best[o] = a.reduce[reduce_off_best(0, T, nu) + o];   // mppi.py:495
        }
    }
    // returned plan: clone(mean) (halton) or U[:u_per_command] (simple), then the filter
    const int rows = a.mode_simple ? a.u_per_command : T;
    for (int o = tid; o < n; o += blockDim.x) {
        const int t = o / nu, j = o % nu;
        float v = 0.0f;
        if (t < rows) {
            if (a.filter_u && rows >= 9) {
                int p, base;
                if (t < 4) { p = t; base = 0; }
                else if (t >= rows - 4) { p = 8 - (rows - 1 - t); base = rows - 9; }
                else { p = 4; base = t - 4; }
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 9; ++i) acc += sg_coef(p, i) * sm[(base + i) * nu + j];
                v = acc;
            } else {
                v = sm[o];
            }
        }
        a.action_out[o] = v;
    }
    for (int o = tid; o < M3_TOPK * T * 2; o += blockDim.x)
        a.top_trajs[o] = a.reduce[reduce_off_top(T, nu) + o];
}
void launch_finalize(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), a.T * a.nu * sizeof(float), s, a);
}

}  // namespace m3
