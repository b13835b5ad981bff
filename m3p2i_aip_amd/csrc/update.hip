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
#include <cstdlib>
#include "m3_internal.hpp"

namespace m3 {

constexpr int WT_MAX = 1024;  // max threads of k_weights (16 wavefronts); 256 for small K

// exp for the softmin weights: v_exp_f32 on x*log2(e) (2 instructions, ~2 ulp + the argument
// rounding, i.e. <= ~5e-6 relative at |x| = 88) instead of the ~40-instruction correctly
// rounded expf.  The bar on the weights is 1e-3 and the same function is used for eta and
// for the weights, so they still sum to one.
__device__ __forceinline__ float m3_exp(float x) { return __expf(x); }

// ---- wavefront (64-lane) reductions on the DPP cross-lane path ---------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~100+ cycles each, six
// dependent steps per reduction); the update kernels are nothing but chains of such
// reductions (20+20 top-k rounds, up to ~25 beta-search passes), so they were latency-bound
// on it.  DPP row operations are ordinary VALU instructions: butterfly inside each 16-lane
// row with quad_perm / row_half_mirror / row_mirror, then row_bcast:15 / row_bcast:31 fold
// the four rows into lane 63, which v_readlane broadcasts.
#define M3_DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define M3_DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define M3_DPP_HALF_MIRROR 0x141
#define M3_DPP_MIRROR 0x140
#define M3_DPP_BCAST15 0x142
#define M3_DPP_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL,
                                                       ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<M3_DPP_XOR1, 0xF>(0.0f, v);
    v += dpp_f<M3_DPP_XOR2, 0xF>(0.0f, v);
    v += dpp_f<M3_DPP_HALF_MIRROR, 0xF>(0.0f, v);
    v += dpp_f<M3_DPP_MIRROR, 0xF>(0.0f, v);
    v += dpp_f<M3_DPP_BCAST15, 0xA>(0.0f, v);
    v += dpp_f<M3_DPP_BCAST31, 0xC>(0.0f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
    const float INF = __builtin_inff();
    v = fminf(v, dpp_f<M3_DPP_XOR1, 0xF>(INF, v));
    v = fminf(v, dpp_f<M3_DPP_XOR2, 0xF>(INF, v));
    v = fminf(v, dpp_f<M3_DPP_HALF_MIRROR, 0xF>(INF, v));
    v = fminf(v, dpp_f<M3_DPP_MIRROR, 0xF>(INF, v));
    v = fminf(v, dpp_f<M3_DPP_BCAST15, 0xA>(INF, v));
    v = fminf(v, dpp_f<M3_DPP_BCAST31, 0xC>(INF, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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
constexpr int TOPK_RPT = 8;  // costs per thread held in registers by a top-k workgroup
constexpr int WEIGHTS_LDS_MAX = 15000;  // costs staged in LDS by k_weights (60 KB, within the
                                        // 64 KB a launch gets without a function attribute)
int weights_lds_floats(int Kg) { return Kg < WEIGHTS_LDS_MAX ? Kg : WEIGHTS_LDS_MAX; }
__device__ __forceinline__ bool vi_less(float av, int ai, float bv, int bi) {
    return (av < bv) || (av == bv && ai < bi);
}
// (value, index) argmin as a min over 64-bit keys: the float is mapped to an order-preserving
// unsigned (sign flip) in the high word, the index sits in the low word, so one unsigned
// 64-bit min is the lexicographic (value, index) min.  Same DPP butterfly as wave_sum.
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = (unsigned)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __int_as_float((int)((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void key_min_step(unsigned& hi, unsigned& lo) {
    const unsigned ohi = dpp_u<CTRL, ROW_MASK>(0xffffffffu, hi);
    const unsigned olo = dpp_u<CTRL, ROW_MASK>(0xffffffffu, lo);
    const bool take = (ohi < hi) || (ohi == hi && olo < lo);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}
__device__ __forceinline__ VI wave_argmin(VI x) {
    unsigned hi = f2ord(x.v), lo = (unsigned)x.i;
    key_min_step<M3_DPP_XOR1, 0xF>(hi, lo);
    key_min_step<M3_DPP_XOR2, 0xF>(hi, lo);
    key_min_step<M3_DPP_HALF_MIRROR, 0xF>(hi, lo);
    key_min_step<M3_DPP_MIRROR, 0xF>(hi, lo);
    key_min_step<M3_DPP_BCAST15, 0xA>(hi, lo);
    key_min_step<M3_DPP_BCAST31, 0xC>(hi, lo);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    return VI{ord2f(hi), (int)lo};
}
__device__ __forceinline__ VI block_argmin(VI x, VI* lds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    x = wave_argmin(x);
    __syncthreads();
    if (lane == 0) lds[wv] = x;
    __syncthreads();
    VI r = lds[0];
    for (int i = 1; i < nw; ++i)
        if (vi_less(lds[i].v, lds[i].i, r.v, r.i)) r = lds[i];
    return r;
}

// grid = 1 + n_cand workgroups: 0 -> weights/info, 1.. -> top-k stage A (run concurrently).
// JR = costs per thread held in registers by workgroup 0.
template <int JR>
__global__ __launch_bounds__(WT_MAX) void k_weights(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    __shared__ float s_beta[3];
    __shared__ int s_done[3];
    const int Kg = a.Kg, half = Kg / 2;
    const int tid = threadIdx.x;
    const int WT = blockDim.x;
    const float* J = a.Jall;
    const float INF = __builtin_inff();

    if (blockIdx.x >= 1) {
        // top-k weights == k smallest trajectory costs (weights are monotone in J); ties
        // resolved towards the lower sample index.  Stage A (these workgroups, running beside
        // workgroup 0): each workgroup owns TOPK_RPT*blockDim consecutive costs, held in
        // REGISTERS (re-reading J from L2 every round costs ~1 us per round: measured 47 us
        // for K = 2000); every wave extracts the sorted top-k of its registers with shuffles
        // only, wave 0 merges the waves' candidates and writes the workgroup's k candidates.
        // Stage B (merge across workgroups) runs in k_wsum.
        __shared__ VI cand[16 * M3_TOPK];
        const int lane = tid & 63, wv = tid >> 6, nw = WT >> 6;
        const int base = (blockIdx.x - 1) * WT * TOPK_RPT;
        float rv[TOPK_RPT];
#pragma unroll
        for (int e = 0; e < TOPK_RPT; ++e) {
            const int k = base + e * WT + tid;
            rv[e] = (k < Kg) ? J[k] : INF;
        }
        unsigned used = 0u;
        for (int r = 0; r < M3_TOPK; ++r) {
            VI best = {INF, 0x7fffffff};
            int be = -1;
#pragma unroll
            for (int e = 0; e < TOPK_RPT; ++e) {
                const int k = base + e * WT + tid;
                if (!((used >> e) & 1u) && k < Kg && vi_less(rv[e], k, best.v, best.i)) {
                    best.v = rv[e]; best.i = k; be = e;
                }
            }
            const VI win = wave_argmin(best);
            if (be >= 0 && win.i == best.i) used |= 1u << be;
            if (lane == 0) cand[wv * M3_TOPK + r] = win;
        }
        __syncthreads();
        if (wv == 0) {
            float pv = -INF;
            int pi = -1;
            for (int r = 0; r < M3_TOPK; ++r) {
                VI best = {INF, 0x7fffffff};
                for (int c = lane; c < nw * M3_TOPK; c += 64) {
                    const VI x = cand[c];
                    if (vi_less(pv, pi, x.v, x.i) && vi_less(x.v, x.i, best.v, best.i)) best = x;
                }
                best = wave_argmin(best);
                pv = best.v; pi = best.i;
                if (lane == 0) a.cand[(blockIdx.x - 1) * M3_TOPK + r] = best;
            }
        }
        return;
    }

    // The costs are read ONCE into registers (thread t holds J[t + e*blockDim], e < JR): the
    // min pass, the up-to-~25 beta-search passes x 3 searches and the final weight pass are
    // then pure VALU + wave reductions instead of an L2 round trip per element per pass
    // (K = 64000 multi-modal: 207 us with the costs re-read from L2 beyond a 60 KB LDS stage).
    // Costs beyond JR*blockDim (K > 65536) fall back to memory.
    float jr[JR];
#pragma unroll
    for (int e_ = 0; e_ < JR; ++e_) {
        const int k = e_ * WT + tid;
        jr[e_] = (k < Kg) ? J[k] : INF;
    }
    // second tier: the next a.lds_floats costs live in LDS (1024 threads x 48 registers +
    // 15000 LDS floats cover K = 64000 without touching L2 again); third tier: memory
    extern __shared__ __attribute__((aligned(16))) float sJ[];
    const int lds0 = JR * WT;
    const int lds1 = (Kg < lds0 + a.lds_floats) ? Kg : lds0 + a.lds_floats;
    for (int k = lds0 + tid; k < lds1; k += WT) sJ[k - lds0] = J[k];
    __syncthreads();
#define FOR_J(...)                                                                   \
    _Pragma("unroll") for (int e_ = 0; e_ < JR; ++e_) {                              \
        if (e_ * WT >= Kg) break; /* wave-uniform */                                 \
        const int k = e_ * WT + tid;                                                 \
        if (k < Kg) { const float v = jr[e_]; __VA_ARGS__ }                          \
    }                                                                                \
    for (int k = lds0 + tid; k < lds1; k += WT) { const float v = sJ[k - lds0]; __VA_ARGS__ } \
    for (int k = (lds1 > lds0 ? lds1 : lds0) + tid; k < Kg; k += WT) { const float v = J[k]; __VA_ARGS__ }

    // ---- minima (all, first half, second half) ----
    float mn[3] = {INF, INF, INF};
    FOR_J({
        mn[0] = fminf(mn[0], v);
        if (k < half) mn[1] = fminf(mn[1], v); else mn[2] = fminf(mn[2], v);
    })
    block_min<3>(mn, red);

    float beta[3], eta[3];
    int iters[3] = {1, 1, 1};
    if (!a.multi_modal || a.mode_simple) {
        // single softmin: beta persists in info (panda adapts it), simple mode uses lambda
        const float b = a.mode_simple ? a.lambda_ : a.info->beta;
        float e[1] = {0.0f};
        const float nib = -1.0f / b;
        FOR_J({ e[0] += m3_exp(nib * (v - mn[0])); })
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
            FOR_J({
                if (!d0) e[0] += m3_exp(n0 * (v - mn[0]));
                if (k < half) { if (!d1) e[1] += m3_exp(n1 * (v - mn[1])); }
                else { if (!d2) e[2] += m3_exp(n2 * (v - mn[2])); }
            })
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
    FOR_J({
        const float wk = i0 * m3_exp(n0 * (v - mn[0]));
        a.w[k] = wk;
        if (k < half) hs[0] += wk; else hs[1] += wk;
        // argmax of the weights, first index on ties (torch.argmax on CPU): key = -w
        if (vi_less(-wk, k, b0.v, b0.i)) { b0.v = -wk; b0.i = k; }
        if (a.multi_modal && !a.mode_simple) {
            if (k < half) {
                const float w1k = (1.0f / eta[1]) * m3_exp((-1.0f / beta[1]) * (v - mn[1]));
                a.w1[k] = w1k;
                if (vi_less(-w1k, k, b1.v, b1.i)) { b1.v = -w1k; b1.i = k; }
            } else {
                const float w2k = (1.0f / eta[2]) * m3_exp((-1.0f / beta[2]) * (v - mn[2]));
                a.w2[k - half] = w2k;
                if (vi_less(-w2k, k, b2.v, b2.i)) { b2.v = -w2k; b2.i = k; }
            }
        }
    })
#undef FOR_J
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
        // multi-modal: the searched betas are locals in the reference (self.beta / beta_1 /
        // beta_2 are never written, m3p2i.py:58-60), so the persistent beta stays untouched;
        // the values found are reported for diagnostics only
        f->beta_1 = beta[1]; f->beta_2 = beta[2];
    }
}

int weights_threads(int Kg) { return Kg <= 8192 ? 256 : WT_MAX; }
int topk_workgroups(int Kg) {
    const int per = weights_threads(Kg) * TOPK_RPT;
    return (Kg + per - 1) / per;
}

void launch_weights(const UpdateArgs& a, hipStream_t s) {
    // few waves for small K: the block-wide reductions (2 barriers + a serial pass over the
    // waves' partials) dominate, not the K/threads elements per thread
    const int threads = weights_threads(a.Kg);
    UpdateArgs b = a;
    if (threads == 256) {
        b.lds_floats = 0;
        hipLaunchKernelGGL(k_weights<32>, dim3(1 + a.n_cand), dim3(256), 0, s, b);
    } else {
        const int rest = a.Kg - 48 * WT_MAX;
        b.lds_floats = rest <= 0 ? 0 : (rest < WEIGHTS_LDS_MAX ? rest : WEIGHTS_LDS_MAX);
        hipLaunchKernelGGL(k_weights<48>, dim3(1 + a.n_cand), dim3(WT_MAX), b.lds_floats * sizeof(float), s, b);
    }
}

// one workgroup per time step t: sum_k w_k * actions[t][k][:] over the local shard, for the
// global weights and (multi-modal) the two per-mode weight sets; plus row gathers.
constexpr int ST = 256;
__global__ __launch_bounds__(ST) void k_wsum(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    __shared__ int s_top[M3_TOPK];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int Kl = a.Kl, k0 = a.k0, nu = a.nu, T = a.T, half = a.Kg / 2;
    const bool multi = a.multi_modal && !a.mode_simple;
    const float* act = a.actions + (size_t)t * Kl * nu;
    // workgroup T: top-k stage B.  Wave 0 merges the stage-A candidates (registers + DPP
    // argmin rounds), then the whole workgroup gathers the top-k trajectories for every t
    // (mppi.py:252-254) -- in parallel with the T weighted-sum workgroups.
    if (t == T) {
        if (tid < 64) {
            const int lane = tid, nc = a.n_cand * M3_TOPK;
            VI rc[4];  // first 256 candidates in registers (covers K <= 98304)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = lane + 64 * e;
                rc[e] = (c < nc) ? a.cand[c] : VI{__builtin_inff(), 0x7fffffff};
            }
            float pv = -__builtin_inff();
            int pi = -1;
            for (int r = 0; r < M3_TOPK; ++r) {
                VI best = {__builtin_inff(), 0x7fffffff};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (vi_less(pv, pi, rc[e].v, rc[e].i) && vi_less(rc[e].v, rc[e].i, best.v, best.i)) best = rc[e];
                for (int c = lane + 256; c < nc; c += 64) {
                    const VI x = a.cand[c];
                    if (vi_less(pv, pi, x.v, x.i) && vi_less(x.v, x.i, best.v, best.i)) best = x;
                }
                best = wave_argmin(best);
                pv = best.v; pi = best.i;
                if (lane == 0) { s_top[r] = best.i; a.top_idx[r] = best.i; }
            }
        }
        __syncthreads();
        for (int o = tid; o < M3_TOPK * T * 2; o += ST) {
            const int c = o & 1, tt = (o >> 1) % T, r = (o >> 1) / T;
            const int li = s_top[r] - k0;
            float v = 0.0f;  // zero unless this rank owns the sample (summed by the all-reduce)
            if (li >= 0 && li < Kl) v = a.states[((size_t)tt * Kl + li) * 4 + (c ? 2 : 0)];  // [0, 2]
            a.reduce[reduce_off_top(T, nu) + o] = v;
        }
        return;
    }
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
}
void launch_wsum(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_wsum, dim3(a.T + 1), dim3(ST), 0, s, a);
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
