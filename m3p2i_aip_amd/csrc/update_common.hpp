// update_common.hpp -- what the update's translation units share (update.hip: the general multi-launch path;
// update_small.hip: k_update_small, the one-launch update of the reference's sizes; update_sharded.hip: the kernels
// of the sharding protocols and the three-launch large-K update): reductions, the top-k selection, the beta ladder
// and the three searches on it (search_body), the re-generated actions of the sharded protocols, the finalize.
// Device code only (header-inline); every translation unit has its own copy of the ladder's constant table
// (no relocatable device code in this build: a __constant__ cannot cross translation units) -- m3_create fills all.
#pragma once
#include "m3_internal.hpp"

namespace m3 {

constexpr int WT_MAX = 1024;       // threads of k_weights for large K (16 wavefronts); 256 for small K
constexpr int PREP_T = 256;        // threads of k_mins / a top-k stage-A workgroup
constexpr int PREP_RPT = 16;       // costs per thread held in registers there
// (LAD_S = 64 shrink-ladder points 0.9^j, LAD_G = 32 grow-ladder points 1.2^j, LAD_N: m3_internal.hpp)
constexpr int LAD_EL = 256;        // costs per k_ladder workgroup
constexpr int WEIGHTS_LDS_MAX = 32768;  // costs staged in LDS by k_weights (128 KB of the CU's 160 KB)

// exp for the softmin weights: v_exp_f32 on x*log2(e) (2 instructions, ~2 ulp + the argument
// rounding, i.e. <= ~5e-6 relative at |x| = 88) instead of the ~40-instruction correctly
// rounded expf.  The bar on the weights is 1e-3 and the same function is used for eta and
// for the weights, so they still sum to one.
__device__ __forceinline__ float m3_exp(float x) { return __expf(x); }

// an optimisation barrier for a value: the compiler must take it as given
__device__ __forceinline__ float uniform_f(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
// shard of global sample k (k < 2^24: exact in binary32; one multiply + a fix-up instead of an integer division)
__device__ __forceinline__ int shard_of(int k, int Kls, float inv_Kls) {
    int r = (int)((float)k * inv_Kls);
    r -= (r * Kls > k) ? 1 : 0;
    r += ((r + 1) * Kls <= k) ? 1 : 0;
    return r;
}
// trajectory cost of GLOBAL sample k: the contiguous array, or (shard_mix = 2) the head of its shard's
// gathered record
__device__ __forceinline__ float jcost(const UpdateArgs& a, int k) {
    if (!a.fast) return a.Jall[k];
    const int r = shard_of(k, a.Kls, 1.0f / (float)a.Kls);
    return a.records_all[(size_t)r * a.rec_len + (k - r * a.Kls)];
}

// ---- wavefront (64-lane) reductions on the DPP cross-lane path ---------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS-crossbar round trip, ~100+ cycles each, six
// dependent steps per reduction); the update kernels are chains of such reductions, so they
// were latency-bound on it.  DPP row operations are ordinary VALU instructions: butterfly
// inside each 16-lane row with quad_perm / row_half_mirror / row_mirror, then row_bcast:15 /
// row_bcast:31 fold the four rows into lane 63, which v_readlane broadcasts.
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

// ---- top-k: threshold filter + rank counting -------------------------------------------------
// top-k weights == k smallest costs (weights are monotone in J; ties towards the lower sample
// index).  Extracting the k minima by k argmin rounds cost ~1.4 us per round (29 + 15 us for the
// two stages at K = 2000).  Instead:
//   1. a threshold tau that is certainly >= the k-th smallest cost of the workgroup: every lane
//      takes the minimum of its registers, each wave radix-selects the k-th smallest of its 64
//      lane minima (32 ballot steps on order-preserving keys), tau = min over the waves -- the
//      wave that supplied it has >= k elements <= tau;
//   2. the few elements <= tau (typically 20..60 of 4096) are compacted into LDS;
//   3. every survivor counts how many survivors precede it in (cost, index) order -- its rank --
//      and the ones with rank < k write themselves to slot `rank`: sorted output, no rounds.
// Stage A does this per workgroup of 4096 costs, stage B over the workgroups' sorted lists
// (tau = smallest of the lists' k-th entries).  If the survivors do not fit the LDS list
// (massive ties, e.g. all costs equal) the old argmin-round code runs instead.
constexpr int TK_CAP = 1024;

__device__ __forceinline__ unsigned wave_kth_key(unsigned key, int kth) {  // kth: 1-based
    bool active = true;
    unsigned res = 0u;
    int need = kth;
    for (int bit = 31; bit >= 0; --bit) {
        const bool zero = !((key >> bit) & 1u);
        const int cnt = __builtin_popcountll(__ballot(active && zero));
        if (need <= cnt) {
            active = active && zero;
        } else {
            need -= cnt;
            active = active && !zero;
            res |= 1u << bit;
        }
    }
    return res;
}

// sorted top-k of list[0..n) by rank counting -> out[0..M3_TOPK) (padded when n < k).  Four
// lanes share one candidate (each scans a quarter of the list, partial ranks added with two
// quad-permute DPP steps); the scan is unrolled so several LDS reads are in flight.
// The list holds 64-bit keys (order-preserving cost bits : sample index) so that the compare is
// one unsigned 64-bit compare on one ds_read_b64 (a (float, int) pair compare made the compiler
// load the index lazily behind a branch: two dependent LDS round trips per element).
typedef unsigned long long tkey;
__device__ __forceinline__ tkey vi_key(float v, int i) { return ((tkey)f2ord(v) << 32) | (unsigned)i; }
__device__ __forceinline__ VI key_vi(tkey k) { return VI{ord2f((unsigned)(k >> 32)), (int)(unsigned)k}; }

__device__ __forceinline__ void topk_rank_emit(const tkey* list, int n, VI* out, int nt /* threads */) {
    const int tid = threadIdx.x, part = tid & 3, per = nt >> 2;
    for (int c0 = 0; c0 < n; c0 += per) {
        const int c = c0 + (tid >> 2);
        const bool valid = c < n;
        const tkey my = list[valid ? c : 0];
        unsigned rank = 0u;
#pragma unroll 4
        for (int q = part; q < n; q += 4) rank += (list[q] < my) ? 1u : 0u;
        rank += dpp_u<M3_DPP_XOR1, 0xF>(0u, rank);
        rank += dpp_u<M3_DPP_XOR2, 0xF>(0u, rank);
        if (valid && part == 0 && rank < (unsigned)M3_TOPK) out[rank] = key_vi(my);
    }
    for (int r = n + tid; r < M3_TOPK; r += nt) out[r] = VI{__builtin_inff(), 0x7fffffff};
}

// fallback stage A: k argmin rounds per wave over the registers, wave 0 merges the waves' lists.
// (The out-of-line fallbacks take scalars, not the argument struct: a struct passed by reference
// to a non-inlined function is copied to scratch, and a kernel that uses scratch at all pays
// ~3 us more per launch.)
template <int RPT>
__device__ __noinline__ void topk_stage_a_rounds(const float* J, int Kg, int kbase, int blk, VI* out) {
    constexpr int PREP_RPT = RPT;   // (shadows the namespace constant: rows per thread of THIS instance)
    __shared__ VI cand[16 * M3_TOPK];
    const int tid = threadIdx.x, WT = PREP_T;  // called by the first PREP_T threads
    const int lane = tid & 63, wv = tid >> 6, nw = WT >> 6;
    const float INF = __builtin_inff();
    const int base = blk * WT * PREP_RPT;
    float rv[PREP_RPT];
#pragma unroll
    for (int e = 0; e < PREP_RPT; ++e) {
        const int k = base + e * WT + tid;
        const float jv = J[min(k, Kg - 1)];
        rv[e] = (k < Kg) ? jv : INF;
    }
    unsigned used = 0u;
    for (int r = 0; r < M3_TOPK; ++r) {
        VI best = {INF, 0x7fffffff};
        int be = -1;
#pragma unroll
        for (int e = 0; e < PREP_RPT; ++e) {
            const int k = base + e * WT + tid;
            if (!((used >> e) & 1u) && k < Kg && vi_less(rv[e], kbase + k, best.v, best.i)) {
                best.v = rv[e]; best.i = kbase + k; be = e;
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
            if (lane == 0) out[r] = best;
        }
    }
}

// fallback stage B: wave 0 merges the stage-A lists with argmin rounds (registers + global tail)
__device__ __noinline__ void topk_stage_b_rounds(const VI* cands, int n_cand, VI* out) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int lane = tid, nc = n_cand * M3_TOPK;
        VI rc[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int c = lane + 64 * e;
            rc[e] = cands[min(c, nc - 1)];
            if (c >= nc) rc[e] = VI{__builtin_inff(), 0x7fffffff};
        }
        float pv = -__builtin_inff();
        int pi = -1;
        for (int r = 0; r < M3_TOPK; ++r) {
            VI best = {__builtin_inff(), 0x7fffffff};
#pragma unroll
            for (int e = 0; e < 6; ++e)
                if (vi_less(pv, pi, rc[e].v, rc[e].i) && vi_less(rc[e].v, rc[e].i, best.v, best.i)) best = rc[e];
            for (int c = lane + 384; c < nc; c += 64) {
                const VI x = cands[c];
                if (vi_less(pv, pi, x.v, x.i) && vi_less(x.v, x.i, best.v, best.i)) best = x;
            }
            best = wave_argmin(best);
            pv = best.v; pi = best.i;
            if (lane == 0) out[r] = best;
        }
    }
}

// top_idx + the top-k trajectories for every t (mppi.py:252-254; zero rows for samples of other
// ranks: summed by the all-reduce when sharded)
__device__ __forceinline__ void topk_finish(const UpdateArgs& a, const VI* top /* LDS, sorted */, int nt) {
    const int tid = threadIdx.x, T = a.T, Kl = a.Kl, k0 = a.k0;
    if (tid < M3_TOPK) {
        a.top_idx[tid] = top[tid].i;
        if (a.rec_topj) {  // sharded: the ranks' lists are merged after the collective
            a.rec_topj[tid] = top[tid].v;
            a.rec_topi[tid] = __int_as_float(top[tid].i);
        }
    }
    float2* dst = reinterpret_cast<float2*>(a.top_dst);
    const int total = M3_TOPK * T;
    if (a.regen) {
        // the global top-k is a subset of the union of the shards' own top-k lists, whose trajectories
        // came with the gathered records: find each winner in its owner's list, copy the row
        __shared__ int s_src[M3_TOPK];
        if (tid < M3_TOPK) {
            const int gi = top[tid].i;
            int off = -1;
            if (gi >= 0 && gi < a.Kg) {
                const float* rec = a.records_all + (size_t)(gi / a.Kls) * a.rec_len;
                for (int q = 0; q < M3_TOPK; ++q)
                    if (__float_as_int(rec[regen_off_topi(a.Kls) + q]) == gi) off = (int)((size_t)(gi / a.Kls) * a.rec_len + regen_off_trajs(a.Kls) + q * T * 2);   // < 2^25: K_global < 2^24 (m3_create)
            }
            s_src[tid] = off;
        }
        __syncthreads();
        for (int o = tid; o < total; o += nt) {
            const int r = o / T, tt = o - r * T;
            float2 v = make_float2(0.f, 0.f);
            if (s_src[r] >= 0) { v.x = a.records_all[s_src[r] + tt * 2]; v.y = a.records_all[s_src[r] + tt * 2 + 1]; }
            dst[o] = v;
        }
        return;
    }
    // one (x, vx, y, vy) row per (r, t); the rows were written by other CUs (HBM / remote-L2
    // latency per load), so a batch of independent loads is issued before the first is consumed
    const float4* st4 = reinterpret_cast<const float4*>(a.states);
    constexpr int UN = 4;
    for (int o0 = tid; o0 < total; o0 += UN * nt) {
        float4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int o = o0 + u * nt;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < total) {
                const int r = o / T, tt = o - r * T;
                const int li = top[r].i - k0;
                if (li >= 0 && li < Kl) v[u] = st4[(size_t)tt * Kl + li];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int o = o0 + u * nt;
            if (o < total) dst[o] = make_float2(v[u].x, v[u].z);  // states[..., [0, 2]]
        }
    }
}

// Nothing of the control update depends on the top-k selection (only top_idx / top_trajs,
// mppi.py:248-254), so it rides along as EXTRA workgroups of launches that exist anyway, on CUs
// the update does not use: stage A beside workgroup 0 of k_weights, stage B beside the sums of
// k_wsum (no extra launch, no extra stream; with K <= 4096 stage A is the whole selection).
template <int RPT = 16>
__device__ __forceinline__ void topk_stage_a(const UpdateArgs& a, int blk) {
    constexpr int PREP_RPT = RPT;   // rows of PREP_T costs per thread: 16 (4096 costs per workgroup) or 32
    __shared__ tkey flt[TK_CAP];
    __shared__ VI s_top[M3_TOPK];
    __shared__ unsigned s_tau[PREP_T / 64];
    __shared__ int s_cnt[PREP_T / 64];
    const int Kg = a.Kg, tid = threadIdx.x;
    if (tid >= PREP_T) return;  // launched with k_weights' block size: the first 4 waves work
    const int lane = tid & 63, wv = tid >> 6;
    const int base = blk * PREP_T * PREP_RPT;
    float rv[PREP_RPT];
    unsigned mk = 0xffffffffu;
#pragma unroll
    for (int e = 0; e < PREP_RPT; ++e) {
        const int k = base + e * PREP_T + tid;
        const float jv = a.Jall[min(k, Kg - 1)];  // unconditional: the 16 loads stay in flight together
        rv[e] = (k < Kg) ? jv : __builtin_inff();
        if (k < Kg) mk = min(mk, f2ord(rv[e]));
    }
    const unsigned tau_w = wave_kth_key(mk, M3_TOPK);
    if (lane == 0) s_tau[wv] = tau_w;
    __syncthreads();
    unsigned tau = s_tau[0];
#pragma unroll
    for (int w = 1; w < PREP_T / 64; ++w) tau = min(tau, s_tau[w]);
    // compaction without atomics: ballot masks per register row, wave totals through LDS,
    // position = waves before + rows before + lanes before (mbcnt)
    unsigned long long hit[PREP_RPT];
    int tot = 0;
#pragma unroll
    for (int e = 0; e < PREP_RPT; ++e) {
        const int k = base + e * PREP_T + tid;
        hit[e] = __ballot(k < Kg && f2ord(rv[e]) <= tau);
        tot += __builtin_popcountll(hit[e]);
    }
    if (lane == 0) s_cnt[wv] = tot;
    __syncthreads();
    int off = 0, n = 0;
#pragma unroll
    for (int w = 0; w < PREP_T / 64; ++w) {
        if (w < wv) off += s_cnt[w];
        n += s_cnt[w];
    }
#pragma unroll
    for (int e = 0; e < PREP_RPT; ++e) {
        if (hit[e] == 0ull) continue;
        const int pos = off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hit[e] >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned)hit[e], 0u));
        if (((hit[e] >> lane) & 1ull) && pos < TK_CAP) flt[pos] = vi_key(rv[e], a.kbase + base + e * PREP_T + tid);
        off += __builtin_popcountll(hit[e]);
    }
    __syncthreads();
    const bool single = a.n_cand == 1;  // K <= 4096: this workgroup's list is the final one
    VI* out = single ? s_top : a.cand + blk * M3_TOPK;
    if (n <= TK_CAP) topk_rank_emit(flt, n, out, PREP_T);
    else topk_stage_a_rounds<RPT>(a.Jall, a.Kg, a.kbase, blk, out);
    if (single) {
        __syncthreads();
        topk_finish(a, s_top, PREP_T);
    }
}

// stage B (n_cand > 1): merge the workgroups' sorted lists
__device__ __forceinline__ void topk_stage_b(const UpdateArgs& a) {
    __shared__ tkey flt[TK_CAP];
    __shared__ VI s_top[M3_TOPK];
    __shared__ VI s_arg[16];
    __shared__ int s_n;
    const int tid = threadIdx.x, nb = a.n_cand, nc = nb * M3_TOPK;
    VI tau = {__builtin_inff(), 0x7fffffff};
    for (int b = tid; b < nb; b += blockDim.x) {
        const VI x = a.cand[b * M3_TOPK + M3_TOPK - 1];
        if (vi_less(x.v, x.i, tau.v, tau.i)) tau = x;
    }
    if (tid == 0) s_n = 0;
    tau = block_argmin(tau, s_arg);
    const tkey tau_key = vi_key(tau.v, tau.i);
    // A second bound, tight when there are many lists: the 20th smallest of the lists' FIRST elements (every lane's
    // minimum over its lists -> per-wave radix select -> min over waves, as stage A does with the register minima):
    // at least 20 candidates lie at or below it.  The bound above alone (the smallest of the lists' LAST elements)
    // lets ~8 candidates per list through -- 2000 of 5120 at K = 1 M, beyond the LDS list, and the argmin rounds
    // that then ran took 280 us (k_wsum 369 -> 85 us at K = 1 M with this bound).
    __shared__ unsigned s_tau2[16];
    unsigned mk = 0xffffffffu;
    for (int b = tid; b < nb; b += blockDim.x) mk = min(mk, f2ord(a.cand[b * M3_TOPK].v));
    const unsigned tau2_w = wave_kth_key(mk, M3_TOPK);
    if ((tid & 63) == 0) s_tau2[tid >> 6] = tau2_w;
    __syncthreads();
    unsigned tau2 = s_tau2[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) tau2 = min(tau2, s_tau2[w]);
    const unsigned long long* cand64 = reinterpret_cast<const unsigned long long*>(a.cand);
    const int lane = tid & 63;
    for (int c0 = 0; c0 < nc; c0 += blockDim.x) {  // uniform trip count: ballots see whole waves
        const int c = c0 + tid;
        const unsigned long long raw = cand64[min(c, nc - 1)];  // {v: low word, i: high word}
        const tkey key = vi_key(__int_as_float((int)(unsigned)raw), (int)(unsigned)(raw >> 32));
        const bool hit = c < nc && key <= tau_key && (unsigned)(key >> 32) <= tau2;
        const unsigned long long m = __ballot(hit);
        if (m != 0ull) {  // one LDS atomic per wave and iteration that has survivors
            int first = 0;
            if (lane == 0) first = atomicAdd(&s_n, __builtin_popcountll(m));
            first = __builtin_amdgcn_readfirstlane(first);
            const int pos = first + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (hit && pos < TK_CAP) flt[pos] = key;
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n <= TK_CAP) topk_rank_emit(flt, n, s_top, blockDim.x);
    else topk_stage_b_rounds(a.cand, a.n_cand, s_top);
    __syncthreads();
    topk_finish(a, s_top, blockDim.x);
}
// shard_mix = 2: the global top-k is a subset of the union of the shards' own sorted top-k lists, which came
// with the gathered records: rank counting over the N x TOPK candidates (keys are unique: the index is part
// of the key), rows copied from the owners' records.  One workgroup.
__device__ __forceinline__ void topk_merge_records(const UpdateArgs& a) {
    __shared__ tkey s_key[MIX_MAX_RANKS * M3_TOPK];
    __shared__ int s_src[M3_TOPK];
    const int tid = threadIdx.x, nt = blockDim.x, N = a.n_ranks, nc = N * M3_TOPK, T = a.T;
    for (int c = tid; c < nc; c += nt) {
        const float* rec = a.records_all + (size_t)(c / M3_TOPK) * a.rec_len;
        s_key[c] = vi_key(rec[regen_off_topj(a.Kls) + c % M3_TOPK], __float_as_int(rec[regen_off_topi(a.Kls) + c % M3_TOPK]));
    }
    if (tid < M3_TOPK) s_src[tid] = 0;   // (records with duplicated keys -- never from real shards -- must not leave a slot unset)
    __syncthreads();
    for (int c = tid; c < nc; c += nt) {
        const tkey my = s_key[c];
        int rank = 0;
#pragma unroll 4
        for (int q = 0; q < nc; ++q) rank += (s_key[q] < my) ? 1 : 0;
        if (rank < M3_TOPK) {
            s_src[rank] = c;
            a.top_idx[rank] = (int)(unsigned)my;
        }
    }
    __syncthreads();
    for (int o = tid; o < M3_TOPK * T * 2; o += nt) {
        const int slot = o / (T * 2), c = s_src[slot];
        a.top_trajs[o] = a.records_all[(size_t)(c / M3_TOPK) * a.rec_len + regen_off_trajs(a.Kls) + (c % M3_TOPK) * T * 2 + o % (T * 2)];
    }
}

// ---------------------------------------------------------------------------------------
// k_ladder: partial eta sums of one workgroup's LAD_EL costs for every beta of both ladders and
// the three searches (all / first half / second half).  Thread = (ladder index j, element
// parity g): it loops over its half of the workgroup's costs (LDS broadcast reads) -- no
// reductions inside the loop, one LDS combine at the end.  lad[b][j][s].
// beta of ladder point j: 0.9^j (j < LAD_S), 1.2^(j - LAD_S + 1) -- formed by the same repeated f32 multiplication
// as the iterative search (=> identical bits), once, on the host (m3_create -> init_ladder_table): as a loop per
// use its back-edge was taken up to 63 times, ~1 us for the workgroups that need one value
static __constant__ float c_ladder_beta[LAD_N];
__device__ __forceinline__ float ladder_beta(int j) { return c_ladder_beta[j]; }
static inline int init_ladder_table_tu() {     // (this translation unit's copy)
    float t[LAD_N];
    float b = 1.0f;
    for (int j = 0; j < LAD_S; ++j) { t[j] = b; b = b * 0.9f; }
    b = 1.0f;
    for (int j = 0; j < LAD_G; ++j) { b = b * 1.2f; t[LAD_S + j] = b; }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_ladder_beta), t, sizeof(t)) == hipSuccess ? 0 : 1;
}

// ---------------------------------------------------------------------------------------
// Multi-modal search with K > 8192: in ONE workgroup the weights pass alone (3 exps, two stores and
// three running argmaxes per cost) is ~31 us of VALU work at K = 64000, and the minima another ~11.
// Both are embarrassingly parallel, so the path is split:
//   k_mins (existing) -> k_ladder (existing) -> k_search: minima from k_mins' partials, ladder
//   table, walk; iterative passes over J from memory only if a search left its ladder (rare)
//   -> k_apply_weights: every workgroup normalises 4096 costs, keeps its half sums and argmax
//   keys; the last one to finish (write-through partials + relaxed agent ticket) combines them in
//   workgroup order and fills m3_info.  Same values as k_weights; half sums in a different order.
constexpr int AP_T = 256, AP_RPT = 16;  // 4096 costs per workgroup (few workgroups: their tickets serialise on one
                                        // address, ~0.3 us each); <= 256 workgroups (K <= 1M)

struct SearchOut {   // device scratch, written by k_search
    float beta[3], eta[3], mn[3];
};
// The three beta searches (all K / mode 1 / mode 2, m3p2i.py:24-64) of a workgroup of any size: minima, the eta table
// on both beta ladders (mixed from the shards' tables when a.fast, summed from k_ladder's partials otherwise), the
// reference's rule on the table, passes over the costs only for a search that leaves its ladder or reverses.
// Every thread returns with the result; `publish`: thread 0 also writes a.srch and the diagnostics of m3_info.
// COHERENT: the partial tables were written by other workgroups of the SAME launch (k_ladder_search): read them with
// agent-scope loads (the per-XCD L2s are not coherent with each other inside a launch); `have_table` false (its wait
// gave up): every search runs the reference's iterative passes over the costs instead.
template <bool COHERENT = false>
__device__ __forceinline__ void search_body(const UpdateArgs& a, SearchOut& out, bool publish, bool have_table = true,
                                            const float* pre_mn = nullptr /* LDS: the three minima, already formed */) {
    __shared__ float red[3 * 16];
    __shared__ float s_beta[3], s_eta[3], s_mn[3];
    __shared__ int s_done[3], s_it[3];
    __shared__ float s_tab[LAD_N * 3];
    __shared__ float s_part[3 * LAD_N * 3];
    const int Kg = a.Kg, half = a.half_g - a.kbase, tid = threadIdx.x, WT = blockDim.x;
    const float INF = __builtin_inff();
    if (a.fast) {
        // the records carry every shard's minima m_r and its ladder sums relative to them:
        // eta(beta_j) = sum_r exp(-(m_r - m) / beta_j) eta_r(beta_j), m = min_r m_r  (rank order)
        const int N = a.n_ranks, om = regen_off_mins(a.Kls, a.T), ot = regen_off_table(a.Kls, a.T);
        if (tid < 3) {
            float m = INF;
            for (int r = 0; r < N; ++r) m = fminf(m, a.records_all[(size_t)r * a.rec_len + om + tid]);
            s_mn[tid] = m;
        }
        __syncthreads();
        for (int o = tid; o < LAD_N * 3; o += WT) {
            const int j = o / 3, sx = o - 3 * j;
            const float nib = -1.0f / ladder_beta(j), m = s_mn[sx];
            float t = 0.0f;
            for (int r = 0; r < N; ++r) {
                const float* rec = a.records_all + (size_t)r * a.rec_len;
                t += m3_exp(nib * (rec[om + sx] - m)) * rec[ot + o];
            }
            s_tab[o] = t;
        }
    } else {
    if (pre_mn) {
        if (tid < 3) s_mn[tid] = pre_mn[tid];
    } else if (a.n_mins <= 64) {
        if (tid < 3) {
            float m = INF;
            for (int b = 0; b < a.n_mins; ++b) m = fminf(m, a.part_min[b * 3 + tid]);
            s_mn[tid] = m;
        }
    } else {   // (the rollout workgroups' rows, wave_min.hpp: K / 64 of them)
        float mn[3] = {INF, INF, INF};
        for (int b = tid; b < a.n_mins; b += WT) {
            mn[0] = fminf(mn[0], a.part_min[b * 3 + 0]); mn[1] = fminf(mn[1], a.part_min[b * 3 + 1]); mn[2] = fminf(mn[2], a.part_min[b * 3 + 2]);
        }
        block_min<3>(mn, red);
        if (tid < 3) s_mn[tid] = tid == 0 ? mn[0] : (tid == 1 ? mn[1] : mn[2]);
        __syncthreads();
    }
    if constexpr (COHERENT) {
        // k_ladder_search's search workgroup (512 threads): the other workgroups' write-through stores are made visible
        // by ONE agent-scope acquire (L1 / L2 invalidate: ~3.5 us, once) instead of 72 000 L2-bypassing loads, whose
        // latency -- eight in flight per thread -- was 36 us here.  Thread = (float4 column of the 288-entry table,
        // seventh of the workgroups): 16 rows of 16 bytes in flight each, fixed order.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        constexpr int NT = LAD_N * 3, NQ = NT / 4, NSEG = 7;      // 72 x 7 = 504 of the 512 threads
        __shared__ float4 s_p4[NSEG * NQ];
        if (tid < NSEG * NQ) {
            const int q = tid % NQ, sg = tid / NQ;
            const int b0 = (int)(((long long)a.n_lad * sg) / NSEG), b1 = (int)(((long long)a.n_lad * (sg + 1)) / NSEG);
            const float4* src = reinterpret_cast<const float4*>(a.lad) + q;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b = b0; b < b1; b += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(b + u, b1 - 1) * NQ];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (b + u < b1) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            s_p4[sg * NQ + q] = acc;
        }
        __syncthreads();
        for (int q = tid; q < NQ; q += WT) {
            float4 t = s_p4[q];
            for (int g = 1; g < NSEG; ++g) { const float4 x = s_p4[g * NQ + q]; t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w; }
            s_tab[4 * q + 0] = t.x; s_tab[4 * q + 1] = t.y; s_tab[4 * q + 2] = t.z; s_tab[4 * q + 3] = t.w;
        }
    } else {   // ladder table (see k_weights): the workgroups' partial tables added in a fixed order
        const int NT = LAD_N * 3;
        const int nseg = (WT / NT) > 0 ? (WT / NT) : 1;
        auto ldp = [&](size_t o) -> float { return a.lad[o]; };
        for (int idx = tid; idx < nseg * NT; idx += WT) {   // (one trip when the workgroup has >= 288 threads)
            const int o = idx % NT, sg = idx / NT;
            const int b0 = (int)(((long long)a.n_lad * sg) / nseg), b1 = (int)(((long long)a.n_lad * (sg + 1)) / nseg);
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int b = b0;
            for (; b + 7 < b1; b += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += ldp((size_t)(b + u) * NT + o);
            }
            for (; b < b1; ++b) acc[0] += ldp((size_t)b * NT + o);
            s_part[sg * NT + o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
        __syncthreads();
        for (int q = tid; q < NT; q += WT) {
            float t = s_part[q];
            for (int g = 1; g < nseg; ++g) t += s_part[g * NT + q];
            s_tab[q] = t;
        }
    }
    }
    __syncthreads();
    if (tid < 3 && !have_table) { s_beta[tid] = 1.0f; s_eta[tid] = 0.0f; s_done[tid] = 0; s_it[tid] = 0; }
    if (tid < 3 && have_table) {  // the reference's rule on the table (m3p2i.py:35-51), as in k_weights
        const int s = tid;
        float b = 1.0f, et = s_tab[0 * 3 + s];
        int it = 1, done = 0;
        if (et > 10.0f) {
            int j = 0;
            for (;;) {
                b = b * 0.9f; ++j;
                if (j >= LAD_S) break;
                et = s_tab[j * 3 + s]; ++it;
                if (et > 10.0f) continue;
                if (et < 3.0f) b = b * 1.2f;
                else done = 1;
                break;
            }
        } else if (et < 3.0f) {
            int j = 0;
            for (;;) {
                b = b * 1.2f; ++j;
                if (j > LAD_G) break;
                et = s_tab[(LAD_S + j - 1) * 3 + s]; ++it;
                if (et < 3.0f) continue;
                if (et > 10.0f) b = b * 0.9f;
                else done = 1;
                break;
            }
        } else {
            done = 1;
        }
        s_beta[s] = b; s_eta[s] = et; s_done[s] = done; s_it[s] = it;
    }
    __syncthreads();
    const float m0 = s_mn[0], m1 = s_mn[1], m2 = s_mn[2];
    for (int pass = 0; pass < 1000; ++pass) {  // searches that left their ladder: passes over J in memory
        const float b0 = s_beta[0], b1 = s_beta[1], b2 = s_beta[2];
        const int d0 = s_done[0], d1 = s_done[1], d2 = s_done[2];
        if (d0 && d1 && d2) break;
        float e[3] = {0.0f, 0.0f, 0.0f};
        const float n0 = -1.0f / b0, n1 = -1.0f / b1, n2 = -1.0f / b2;
        for (int k = tid; k < Kg; k += WT) {
            const float v = jcost(a, k);
            if (!d0) e[0] += m3_exp(n0 * (v - m0));
            if (k < half) { if (!d1) e[1] += m3_exp(n1 * (v - m1)); }
            else { if (!d2) e[2] += m3_exp(n2 * (v - m2)); }
        }
        block_sum<3>(e, red);
        __syncthreads();
        if (tid < 3 && !s_done[tid]) {
            const float et = e[tid];
            s_eta[tid] = et;
            s_it[tid] = s_it[tid] + 1;
            if (et > 10.0f) s_beta[tid] = s_beta[tid] * 0.9f;
            else if (et < 3.0f) s_beta[tid] = s_beta[tid] * 1.2f;
            else s_done[tid] = 1;
        }
        __syncthreads();
    }
    for (int sx = 0; sx < 3; ++sx) { out.beta[sx] = s_beta[sx]; out.eta[sx] = s_eta[sx]; out.mn[sx] = s_mn[sx]; }
    if (publish && tid == 0) {
        SearchOut* o = a.srch;
        for (int s = 0; s < 3; ++s) { o->beta[s] = s_beta[s]; o->eta[s] = s_eta[s]; o->mn[s] = s_mn[s]; }
        m3_info* f = a.info;
        f->eta = s_eta[0]; f->eta_1 = s_eta[1]; f->eta_2 = s_eta[2];
        f->iters = s_it[0]; f->iters_1 = s_it[1]; f->iters_2 = s_it[2];
        f->beta_1 = s_beta[1]; f->beta_2 = s_beta[2];   // diagnostics; info->beta stays (m3p2i.py:58-60)
    }
}
constexpr int ST = 256;
constexpr int WS_BATCH = 8;     // loads in flight per thread and array
// samples per chunk: 8192, more when that would give more than 32 chunks per time step -- their
// arrival tickets share one address per time step and serialise (~0.3 us each)
__host__ __device__ inline int wsum_chunk_len(int Kl) {
    const int unit = 2048;   // WS_BATCH * ST
#ifndef M3_WSUM_MAX_CHUNKS
#define M3_WSUM_MAX_CHUNKS 32
#endif
    const int per32 = (((Kl + M3_WSUM_MAX_CHUNKS - 1) / M3_WSUM_MAX_CHUNKS) + unit - 1) / unit * unit;
#ifndef M3_WSUM_MIN_CHUNK
#define M3_WSUM_MIN_CHUNK 8192
#endif
    return per32 > M3_WSUM_MIN_CHUNK ? per32 : M3_WSUM_MIN_CHUNK;
}

template <bool SC1>
__device__ __forceinline__ void finalize_body(const UpdateArgs& a, float* sm);  // defined below

// The action the rollout formed for GLOBAL sample k at time step t (mppi.py:381-416 + :297-302, as in
// rollout_point.hip / rollout_panda.hip: same f32 operations, same order => the same bits), from the
// sample's noise row and the replicated plan -- what the "regen" sharding recomputes instead of
// communicating.  Halton-spline mode only (explicit noise table).
//
// The plan rows a time step's actions are assembled from (wave-uniform: loaded once per workgroup):
template <int NU>
struct RegenRows {
    float m1[NU], m2[NU], b1[NU], b2[NU];   // shifted mean of mode 1 (or the single mean) / mode 2, best rows
};
template <int NU>
__device__ __forceinline__ void regen_rows(const UpdateArgs& a, int t, RegenRows<NU>& R) {
    const int T = a.T, ts = (t + 1 < T) ? t + 1 : T - 1;   // _shift_action: mppi.py:266-273
    const bool multi = a.multi_modal != 0;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        R.m1[j] = multi ? a.mean1[ts * NU + j] : a.mean[ts * NU + j];
        R.m2[j] = multi ? a.mean2[ts * NU + j] : R.m1[j];
        R.b1[j] = a.best1[ts * NU + j];
        R.b2[j] = a.best2[ts * NU + j];
    }
}
template <int NU>
__device__ __forceinline__ void regen_action(const UpdateArgs& a, const RegenRows<NU>& R, int k, const float* drow,
                                             float (&e)[NU]) {
    const bool multi = a.multi_modal != 0;
    const bool is_last = k == a.Kg - 1;
    const bool first = k < a.half_g;
    const bool use_best = multi && (k == 0 || k == a.half_g);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const float d = is_last ? 0.0f : drow[j];
        float aj = fmaxf(fminf((first ? R.m1[j] : R.m2[j]) + d * a.scale_tril[j], a.u_max[j]), a.u_min[j]);
        if (use_best) aj = (k == 0) ? R.b1[j] : R.b2[j];
        if (NU == 9 && j >= 7) {
            if (a.gripper_cmd == 1) aj = 1.5f;
            else if (a.gripper_cmd == 2) aj = -1.5f;
        }
        float uj = a.u_scale * aj;
        if (a.sample_null_action && is_last) uj = 0.0f;
        e[j] = uj;   // mppi.py:313
    }
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

// SC1: the reduce buffer was written by OTHER workgroups of the same launch (fused into k_wsum):
// read it with write-through-coherent loads; from its own launch (k_finalize) plain loads do.
template <bool SC1>
__device__ __forceinline__ float rd_reduce(const float* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool SC1>
__device__ __forceinline__ void finalize_body(const UpdateArgs& a, float* sm /* LDS [T*nu] */) {
    const int T = a.T, nu = a.nu, n = T * nu, tid = threadIdx.x;
    // A device-side exchange that gave up on a rank (p2p.hip: sticky error word, that rank's record filled with NaN): the plan
    // of this command is NaN -- handed out as such, so that nothing acts on a plan built from a part of the samples -- but the
    // warm-start state (means, best trajectories) is NOT overwritten with it: once the peer is back and the word is cleared
    // (m3_p2p_clear_error, or m3_p2p_detach + another transport) the planner continues from its last good plan.  The host raises at its next poll (distributed.attach_p2p).
    // (ONE load of the word per workgroup, shared through LDS: the word may change while the kernel runs -- a host
    // clear, m3_p2p_clear_error -- and threads that read it on their own could part ways around the barriers below)
    if (a.p2p_err) {
        __shared__ int s_p2p_err;
        if (tid == 0) s_p2p_err = __hip_atomic_load(a.p2p_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (s_p2p_err != 0) {
            for (int o = tid; o < n; o += blockDim.x) a.action_out[o] = __builtin_nanf("");
            return;
        }
    }
    const bool multi = a.multi_modal && !a.mode_simple;
    const float* ps = a.reduce + reduce_off_psum(0, T, nu);
    const float wtot = a.info->wsum_push + a.info->wsum_pull;
    for (int o = tid; o < n; o += blockDim.x) {
        const int t = o / nu, j = o % nu;
        float nv;
        if (a.mode_simple) {
            const int ts = (t + 1 == T) ? 0 : t + 1;  // rolled U
            const float u = a.mean[ts * nu + j];
            nv = u + (rd_reduce<SC1>(ps + o) - u * wtot);               // U += sum_k w_k (a_k - U): mppi.py:231
        } else {
            const int ts = (t + 1 < T) ? t + 1 : T - 1;  // shifted mean
            nv = (1.0f - a.step_size_mean) * a.mean[ts * nu + j] + a.step_size_mean * rd_reduce<SC1>(ps + o);
        }
        sm[o] = nv;
    }
    __syncthreads();
    for (int o = tid; o < n; o += blockDim.x) {
        a.mean[o] = sm[o];
        if (multi) {
            a.mean1[o] = rd_reduce<SC1>(a.reduce + reduce_off_psum(1, T, nu) + o);  // m3p2i.py:82-83
            a.mean2[o] = rd_reduce<SC1>(a.reduce + reduce_off_psum(2, T, nu) + o);
            a.best1[o] = rd_reduce<SC1>(a.reduce + reduce_off_best(1, T, nu) + o);  // m3p2i.py:77-78
            a.best2[o] = rd_reduce<SC1>(a.reduce + reduce_off_best(2, T, nu) + o);
        } else if (!a.mode_simple) {
            a.best[o] = rd_reduce<SC1>(a.reduce + reduce_off_best(0, T, nu) + o);   // mppi.py:495
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
    if (a.Kl != a.Kg)  // sharded: the rows were summed over ranks in the reduce buffer
        for (int o = tid; o < M3_TOPK * T * 2; o += blockDim.x)
            a.top_trajs[o] = rd_reduce<SC1>(a.reduce + reduce_off_top(T, nu) + o);
}

}  // namespace m3
