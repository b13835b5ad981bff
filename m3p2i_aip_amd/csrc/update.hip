// update.hip -- importance-weight update of MPPI / M3P2I (gfx950).
//
//   k_update_small : the WHOLE update (weights, beta search, sums, top-k, mean update / filter) in one
//               launch for the sizes of the reference's configs: K <= 4096 per GPU (C2, C3, C4, a
//               shard_mix rank), K <= 16384 for single-mode point_env (the north-star size).  The
//               kernels below are the general path: larger K and the sharded phases.
//   k_mins    : (multi-modal only) per-workgroup minima of the trajectory costs
//   k_ladder  : (multi-modal only) eta(beta) for the whole ladder of betas the reference's
//               on-the-fly search can visit, in ONE chip-wide pass
//               update_infinite_beta           m3p2i.py:24-44
//   k_weights : walks the search on the ladder table (iterative passes only after a direction
//               reversal), softmin weights, argmax, half sums; top-k stage A as extra workgroups
//               _exp_util                      mppi.py:430-456
//               _multi_modal_exp_util          m3p2i.py:46-64
//               simple-mode weights            mppi.py:225-229
//               argmax / top-k                 mppi.py:248, 493; m3p2i.py:75-76
//   k_search + k_apply_weights : the same for the multi-modal search with K > 8192, split so that the
//               weights pass runs on many workgroups
//   k_wsum    : weighted action sums + best / top-trajectory row gathers (per time step); top-k
//               stage B as an extra workgroup; for the unsharded m3_command also k_finalize's work
//               mppi.py:497-498, 252-254; m3p2i.py:77-83
//   k_mix     : (sharded single-mode) per-rank softmin records -> the reduce buffer
//   k_finalize: mean update, per-mode means, simple-mode U update, Savitzky-Golay
//               mppi.py:502-503, 231, 245, 257-263; m3p2i.py:86-87
//
// The reference runs each beta-search pass as exp + sum kernels and a host sync (10-25 passes x
// 3 searches per command()).  A pass is a full reduction over the K costs, and the passes are
// sequential -- on one workgroup that is ~7 us per pass at K = 64000 (measured: 190-207 us per
// command).  But the betas a search can visit are known in advance: it starts at 1 and moves by
// x0.9 while eta > 10 or by x1.2 while eta < 3, so until it reverses direction it walks the
// ladder {0.9^j} or {1.2^j}.  k_ladder evaluates eta on both ladders for all three searches in
// one pass spread over the whole chip (K x 2 x 96 exps: ~3 us), and one thread per search then
// walks the table.  Only a search that overshoots the [3,10] window and has to turn around
// continues with iterative passes.  Values are identical to the iterative search (same betas by
// repeated multiplication, same per-sample exponent expression); only the summation order
// differs.  No MFMA: there is no dense contraction in this path.
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

// ---------------------------------------------------------------------------------------
// minima only (critical path of the multi-modal search: k_ladder needs them)
__global__ __launch_bounds__(PREP_T) void k_mins(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    const int Kg = a.Kg, half = Kg / 2, tid = threadIdx.x;
    const float INF = __builtin_inff();
    const int base = blockIdx.x * PREP_T * PREP_RPT;
    float mn[3] = {INF, INF, INF};
#pragma unroll
    for (int e = 0; e < PREP_RPT; ++e) {
        const int k = base + e * PREP_T + tid;
        const int kc = min(k, Kg - 1);
        float jv;
        if (a.regen) {   // (uniform) first reader of the gathered records: shard by shard -> one [K_global] array
            const int r = kc / a.Kls;
            jv = a.records_all[(size_t)r * a.rec_len + (kc - r * a.Kls)];
            if (k < Kg) a.Jout[k] = jv;
        } else {
            jv = a.Jall[kc];  // unconditional: loads stay in flight together
        }
        const float v = (k < Kg) ? jv : INF;
        mn[0] = fminf(mn[0], v);
        if (k < half) mn[1] = fminf(mn[1], v); else mn[2] = fminf(mn[2], v);
    }
    block_min<3>(mn, red);
    if (tid < 3) a.part_min[blockIdx.x * 3 + tid] = mn[tid];
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

int weights_threads(int Kg);
int mins_workgroups(int Kg) {
    const int per = PREP_T * PREP_RPT;
    return (Kg + per - 1) / per;
}
int topk_workgroups(int Kg) {
    const int per = PREP_T * PREP_RPT;
    return (Kg + per - 1) / per;
}
void launch_mins(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_mins, dim3(a.n_mins), dim3(PREP_T), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// k_ladder: partial eta sums of one workgroup's LAD_EL costs for every beta of both ladders and
// the three searches (all / first half / second half).  Thread = (ladder index j, element
// parity g): it loops over its half of the workgroup's costs (LDS broadcast reads) -- no
// reductions inside the loop, one LDS combine at the end.  lad[b][j][s].
// beta of ladder point j: 0.9^j (j < LAD_S), 1.2^(j - LAD_S + 1) -- formed by the same repeated f32 multiplication
// as the iterative search (=> identical bits), once, on the host (m3_create -> init_ladder_table): as a loop per
// use its back-edge was taken up to 63 times, ~1 us for the workgroups that need one value
__constant__ float c_ladder_beta[LAD_N];
__device__ __forceinline__ float ladder_beta(int j) { return c_ladder_beta[j]; }
int init_ladder_table() {
    float t[LAD_N];
    float b = 1.0f;
    for (int j = 0; j < LAD_S; ++j) { t[j] = b; b = b * 0.9f; }
    b = 1.0f;
    for (int j = 0; j < LAD_G; ++j) { b = b * 1.2f; t[LAD_S + j] = b; }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_ladder_beta), t, sizeof(t)) == hipSuccess ? 0 : 1;
}
__global__ __launch_bounds__(256) void k_ladder(const UpdateArgs a) {
    __shared__ float2 sd[LAD_EL];  // (J - min_all, J - min_of_its_half); +inf past the end => exp = 0
    __shared__ float sacc[2][LAD_N][3];
    __shared__ float smn[3];
    const int Kg = a.Kg, half = Kg / 2, tid = threadIdx.x, b = blockIdx.x;
    if (tid < 3) {
        float m = __builtin_inff();
        for (int i = 0; i < a.n_mins; ++i) m = fminf(m, a.part_min[i * 3 + tid]);
        smn[tid] = m;
    }
    __syncthreads();
    {
        const int k = b * LAD_EL + tid;
        float2 d = make_float2(__builtin_inff(), __builtin_inff());
        if (k < Kg) {
            const float v = a.Jall[k];
            d.x = v - smn[0];
            d.y = v - smn[(k < half) ? 1 : 2];
        }
        sd[tid] = d;
    }
    __syncthreads();
    if (tid < 2 * LAD_N) {
        const int j = tid % LAD_N, g = tid / LAD_N;
        const float nib = -1.0f / ladder_beta(j);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 8
        for (int i = 0; i < LAD_EL / 2; ++i) {
            const int e = 2 * i + g;
            const float2 d = sd[e];              // same address for every lane of a group: broadcast
            a0 += m3_exp(nib * d.x);
            const float xh = m3_exp(nib * d.y);
            if (b * LAD_EL + e < half) a1 += xh; else a2 += xh;
        }
        sacc[g][j][0] = a0; sacc[g][j][1] = a1; sacc[g][j][2] = a2;
    }
    __syncthreads();
    for (int o = tid; o < LAD_N * 3; o += 256) {
        const int j = o / 3, s = o % 3;
        a.lad[((size_t)b * LAD_N + j) * 3 + s] = sacc[0][j][s] + sacc[1][j][s];
    }
}
int ladder_workgroups(int Kg) { return (Kg + LAD_EL - 1) / LAD_EL; }
void launch_ladder(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_ladder, dim3(a.n_lad), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// k_weights: ONE workgroup.  JR = costs per thread held in registers.
// JR costs per thread in registers, NT threads: <32, 256> for K <= 8192 (512-VGPR budget),
// <24, 1024> beyond (16 waves share the SIMDs' register files: 128 VGPRs each, so the register
// tier is kept small enough not to spill and the LDS tier takes the next 32768 costs).
template <int JR, int NTHR>
__global__ __launch_bounds__(NTHR) void k_weights(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    __shared__ float s_beta[3], s_eta[3];
    __shared__ int s_done[3], s_it[3];
    __shared__ float s_tab[LAD_N * 3];
    __shared__ float s_part[3 * LAD_N * 3];  // WT_MAX / 288 = 3 segments
    const int Kg = a.Kg, half = a.half_g - a.kbase;  // k < half <=> global index in the first mode
    const int tid = threadIdx.x;
    const int WT = blockDim.x;
    const float* J = a.Jall;
    const float INF = __builtin_inff();
    const bool multi = a.multi_modal && !a.mode_simple;
    if (blockIdx.x > 0) {  // workgroups 1..n_cand: top-k stage A, concurrent with workgroup 0
        topk_stage_a(a, blockIdx.x - 1);
        return;
    }

    // The costs are read ONCE into registers (thread t holds J[t + e*blockDim], e < JR); a
    // second tier of a.lds_floats costs lives in LDS, anything beyond falls back to memory.
    // Every later pass is then pure VALU + wave reductions.
    float jr[JR];
#pragma unroll
    for (int e_ = 0; e_ < JR; ++e_) {
        const int k = e_ * WT + tid;
        const float jv_ = J[min(k, Kg - 1)];
        jr[e_] = (k < Kg) ? jv_ : INF;
    }
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
    if (!multi) {
        // single softmin: beta persists in info (panda adapts it), simple mode uses lambda
        const float b = a.mode_simple ? a.lambda_ : a.info->beta;
        float e[1] = {0.0f};
        const float nib = -1.0f / b;
        FOR_J({ e[0] += m3_exp(nib * (v - mn[0])); })
        block_sum<1>(e, red);
        beta[0] = b; eta[0] = e[0];
        beta[1] = beta[2] = 1.0f; eta[1] = eta[2] = 0.0f;
    } else {
        // (1) ladder table: sum k_ladder's partials over its workgroups.  Each of the 288 table
        // entries is owned by `nseg` threads that split the workgroups between them (8 loads in
        // flight each); fixed association order, so every rank / launch adds in the same order.
        {
            const int NT = LAD_N * 3;
            const int nseg = (WT / NT) > 0 ? (WT / NT) : 1;       // 3 with 1024 threads, 1 with 256
            for (int o0 = 0; o0 < NT; o0 += WT) {                 // one trip unless WT < 288
                const int idx = o0 + tid;
                const int o = (nseg > 1) ? (tid % NT) : idx, sg = (nseg > 1) ? (tid / NT) : 0;
                if (sg < nseg && o < NT) {
                    const int b0 = (int)(((long long)a.n_lad * sg) / nseg), b1 = (int)(((long long)a.n_lad * (sg + 1)) / nseg);
                    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    int b = b0;
                    for (; b + 7 < b1; b += 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[u] += a.lad[(size_t)(b + u) * NT + o];
                    }
                    for (; b < b1; ++b) acc[0] += a.lad[(size_t)b * NT + o];
                    s_part[sg * NT + o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
                }
                if (nseg > 1) break;
            }
            __syncthreads();
            for (int o = tid; o < NT; o += WT) {
                float t = s_part[o];
                for (int sg = 1; sg < nseg; ++sg) t += s_part[sg * NT + o];
                s_tab[o] = t;
            }
        }
        __syncthreads();
        // (2) one thread per search walks the reference's rule on the table; each search
        // restarts at beta = 1 (beta_1/beta_2/beta are never written back: m3p2i.py:58-60)
        if (tid < 3) {
            const int s = tid;
            float b = 1.0f, et = s_tab[0 * 3 + s];
            int it = 1, done = 0;
            if (et > 10.0f) {
                int j = 0;
                for (;;) {
                    b = b * 0.9f; ++j;
                    if (j >= LAD_S) break;                 // off the ladder: continue iteratively
                    et = s_tab[j * 3 + s]; ++it;
                    if (et > 10.0f) continue;
                    if (et < 3.0f) b = b * 1.2f;           // overshoot: reversal, continue iteratively
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
        // (3) iterative passes for searches that left their ladder (rare)
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
                s_eta[tid] = et;
                s_it[tid] = s_it[tid] + 1;
                if (et > 10.0f) s_beta[tid] = s_beta[tid] * 0.9f;
                else if (et < 3.0f) s_beta[tid] = s_beta[tid] * 1.2f;
                else s_done[tid] = 1;
            }
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) { beta[s] = s_beta[s]; eta[s] = s_eta[s]; iters[s] = s_it[s]; }
    }

    // ---- normalised weights, half sums, argmax ----
    // A search that ended with `found` did not change beta after the eta it accepted, so
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
        if (multi) {
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
    if (multi) {
        b1 = block_argmin(b1, redvi);
        b2 = block_argmin(b2, redvi);
    }
    if (tid == 0) {
        m3_info* f = a.info;
        f->eta = eta[0]; f->eta_1 = eta[1]; f->eta_2 = eta[2];
        f->iters = iters[0]; f->iters_1 = iters[1]; f->iters_2 = iters[2];
        f->best_idx = a.kbase + b0.i;
        f->best_idx_1 = multi ? b1.i : -1;
        f->best_idx_2 = multi ? b2.i : -1;
        f->wsum_push = hs[0]; f->wsum_pull = hs[1];
        f->pull_preference = hs[1] > hs[0];
        float nb = beta[0];
        if (!a.multi_modal && !a.mode_simple && a.env_type == M3_ENV_PANDA) {  // mppi.py:446-454
            if (eta[0] > 20.0f) nb = nb * 0.9f;
            else if (eta[0] < 10.0f) nb = nb * 1.2f;
        }
        if (a.record) {  // shard_mix: local softmin only; k_mix owns eta, beta and the best index
            a.record[0] = mn[0]; a.record[1] = eta[0];
            a.record[2] = hs[0]; a.record[3] = hs[1];
            a.record[4] = __int_as_float(a.kbase + b0.i);
        } else if (!a.multi_modal && !a.mode_simple) f->beta = nb;
        // multi-modal: the searched betas are locals in the reference (self.beta / beta_1 /
        // beta_2 are never written, m3p2i.py:58-60), so the persistent beta stays untouched;
        // the values found are reported for diagnostics only
        f->beta_1 = beta[1]; f->beta_2 = beta[2];
    }
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
int apply_workgroups(int Kg) { return (Kg + AP_T * AP_RPT - 1) / (AP_T * AP_RPT); }

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
// ---------------------------------------------------------------------------------------
// The unsharded multi-modal update with K > 8192 in THREE launches instead of five (round 4):
//   k_ladder_search -- n_lad ladder workgroups (as k_ladder; the three global minima come from the rows the rollout
//     workgroups left behind, wave_min.hpp, or from k_mins' rows when the costs were not produced by the rollout) +
//     ONE search workgroup that waits for their partial tables (a flag per ladder workgroup, written after its
//     write-through stores: no shared counter -- agent-scope atomics on one address retire at ~0.3 us each, 250 of
//     them would cost more than the launch they save), adds them in workgroup order and walks the table (k_search's
//     body) + the top-k stage-A workgroups.  The search workgroup is the LAST of the ladder's grid, so every ladder
//     workgroup has been dispatched before it; its wait is bounded, and a search whose wait gave up runs the
//     reference's iterative passes over the costs instead (same decisions: tests/test_hip_edge_cases.py).
//   k_regen_part<NU, false> -- weights formed on the fly from the costs + the weighted action sums of 2048-sample chunks
//     (the kernel of the shard_mix = 2 protocol, with the actions loaded instead of re-generated) + top-k stage B,
//   k_regen_done<NU, false> -- chunk combine in chunk order, best rows, finalize.
constexpr int LS_T = 512, LS_G = 4;   // threads of a k_ladder_search workgroup; element groups per ladder point (LS_G * LAD_N <= LS_T)
__device__ __forceinline__ void ladder_block(const UpdateArgs& a, float* red /* 48 */) {
    __shared__ float2 sd[LAD_EL];
    __shared__ float sacc[LS_G][LAD_N][3];
    __shared__ float smn[3];
    const int Kg = a.Kg, half = Kg / 2, tid = threadIdx.x, b = blockIdx.x;
    const float INF = __builtin_inff();
    {
        float mn[3] = {INF, INF, INF};
        for (int r = tid; r < a.n_mins; r += LS_T) {
            mn[0] = fminf(mn[0], a.part_min[r * 3 + 0]); mn[1] = fminf(mn[1], a.part_min[r * 3 + 1]); mn[2] = fminf(mn[2], a.part_min[r * 3 + 2]);
        }
        block_min<3>(mn, red);
        if (tid == 0) { smn[0] = mn[0]; smn[1] = mn[1]; smn[2] = mn[2]; }
    }
    __syncthreads();
    if (tid < LAD_EL) {
        const int k = b * LAD_EL + tid;
        float2 d = make_float2(INF, INF);
        if (k < Kg) {
            const float v = a.Jall[k];
            d.x = v - smn[0];
            d.y = v - smn[(k < half) ? 1 : 2];
        }
        sd[tid] = d;
    }
    __syncthreads();
    if (tid < LS_G * LAD_N) {   // thread = (ladder point j, element group g): 64 of the workgroup's 256 costs each
        const int j = tid % LAD_N, g = tid / LAD_N;
        const float nib = -1.0f / ladder_beta(j);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 8
        for (int i = 0; i < LAD_EL / LS_G; ++i) {
            const int e = LS_G * i + g;
            const float2 d = sd[e];
            a0 += m3_exp(nib * d.x);
            const float xh = m3_exp(nib * d.y);
            if (b * LAD_EL + e < half) a1 += xh; else a2 += xh;
        }
        sacc[g][j][0] = a0; sacc[g][j][1] = a1; sacc[g][j][2] = a2;
    }
    __syncthreads();
    for (int o = tid; o < LAD_N * 3; o += LS_T) {
        const int j = o / 3, sx = o % 3;
        float t = sacc[0][j][sx];
#pragma unroll
        for (int g = 1; g < LS_G; ++g) t += sacc[g][j][sx];
        __hip_atomic_store(&a.lad[((size_t)b * LAD_N + j) * 3 + sx], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.lflag[b], a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(LS_T) void k_ladder_search(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b > a.n_lad) {
        topk_stage_a(a, b - a.n_lad - 1);   // (its first PREP_T threads work)
        return;
    }
    if (b < a.n_lad) {
        ladder_block(a, red);
        return;
    }
    __shared__ int s_ok;
    __shared__ float s_pre[3];
    {   // (the minima while the ladder workgroups are still at work)
        const float INF = __builtin_inff();
        float mn[3] = {INF, INF, INF};
        for (int r = tid; r < a.n_mins; r += LS_T) {
            mn[0] = fminf(mn[0], a.part_min[r * 3 + 0]); mn[1] = fminf(mn[1], a.part_min[r * 3 + 1]); mn[2] = fminf(mn[2], a.part_min[r * 3 + 2]);
        }
        block_min<3>(mn, red);
        if (tid == 0) { s_pre[0] = mn[0]; s_pre[1] = mn[1]; s_pre[2] = mn[2]; s_ok = 1; }
    }
    __syncthreads();
    bool ok = a.ladder_spins > 0;      // (0: tests force the give-up branch)
    for (int q = tid; q < a.n_lad && ok; q += LS_T) {
        int spins = 0;
        while (__hip_atomic_load(&a.lflag[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > a.ladder_spins) { ok = false; break; }
        }
    }
    if (!ok) s_ok = 0;
    __syncthreads();
    SearchOut so;
    search_body<true>(a, so, true, s_ok != 0, s_pre);
}
void launch_ladder_search(const UpdateArgs& a_, hipStream_t s) {
    static const int spins = getenv("M3P2I_LADDER_SPINS") ? atoi(getenv("M3P2I_LADDER_SPINS")) : (1 << 18);
    UpdateArgs a = a_;
    a.ladder_spins = spins;
    hipLaunchKernelGGL(k_ladder_search, dim3(a.n_lad + 1 + a.n_cand), dim3(LS_T), 0, s, a);
}

__global__ __launch_bounds__(WT_MAX) void k_search(const UpdateArgs a) {
    if (blockIdx.x > 0) {  // workgroups 1..n_cand: top-k stage A, concurrent with the search
        if (a.fast) topk_merge_records(a);   // (shard_mix = 2: the global top-k from the shards' own lists)
        else topk_stage_a(a, blockIdx.x - 1);
        return;
    }
    SearchOut so;
    search_body(a, so, true);
}

// MULTI = false: single-mode MPPI with K > 16384 (beyond every reference config, but the rollout's
// saturation region): beta is given, so no search -- k_sumexp_single leaves every workgroup's local
// softmin (m_b, S_b = sum exp(-(J - m_b)/beta)) in part_min, and every workgroup here forms the global
// one from those <= 256 pairs itself: m = min m_b, eta = sum_b exp(-(m_b - m)/beta) S_b (workgroup order).
template <bool MULTI>
__global__ __launch_bounds__(AP_T) void k_apply_weights(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    const int Kg = a.Kg, half = a.half_g - a.kbase, tid = threadIdx.x, nb = gridDim.x;
    const float INF = __builtin_inff();
    SearchOut so;
    if constexpr (MULTI) {
        so = *a.srch;
    } else {
        __shared__ float s_mb[256], s_sb[256], s_so[2];
        const float bs = a.mode_simple ? a.lambda_ : a.info->beta;
        for (int b = tid; b < nb; b += AP_T) { s_mb[b] = a.part_min[b * 3 + 0]; s_sb[b] = a.part_min[b * 3 + 1]; }
        __syncthreads();
        if (tid == 0) {
            float m = INF;
            for (int b = 0; b < nb; ++b) m = fminf(m, s_mb[b]);
            const float nib = -1.0f / bs;
            float et = 0.0f;
            for (int b = 0; b < nb; ++b) et += m3_exp(nib * (s_mb[b] - m)) * s_sb[b];
            s_so[0] = m; s_so[1] = et;
        }
        __syncthreads();
        so.beta[0] = bs; so.eta[0] = s_so[1]; so.mn[0] = s_so[0];
        so.beta[1] = so.beta[2] = 1.0f; so.eta[1] = so.eta[2] = 1.0f; so.mn[1] = so.mn[2] = 0.0f;
    }
    const float i0 = 1.0f / so.eta[0], n0 = -1.0f / so.beta[0];
    const float i1 = 1.0f / so.eta[1], n1 = -1.0f / so.beta[1];
    const float i2 = 1.0f / so.eta[2], n2 = -1.0f / so.beta[2];
    float hs[2] = {0.0f, 0.0f};
    VI b0 = {INF, 0x7fffffff}, b1 = {INF, 0x7fffffff}, b2 = {INF, 0x7fffffff};
    const int base = blockIdx.x * AP_T * AP_RPT;
#pragma unroll
    for (int e = 0; e < AP_RPT; ++e) {
        const int k = base + e * AP_T + tid;
        const float v = a.Jall[min(k, Kg - 1)];
        if (k < Kg) {
            const float wk = i0 * m3_exp(n0 * (v - so.mn[0]));
            a.w[k] = wk;
            if (k < half) hs[0] += wk; else hs[1] += wk;
            if (vi_less(-wk, k, b0.v, b0.i)) { b0.v = -wk; b0.i = k; }
            if constexpr (MULTI) {
                if (k < half) {
                    const float w1k = i1 * m3_exp(n1 * (v - so.mn[1]));
                    a.w1[k] = w1k;
                    if (vi_less(-w1k, k, b1.v, b1.i)) { b1.v = -w1k; b1.i = k; }
                } else {
                    const float w2k = i2 * m3_exp(n2 * (v - so.mn[2]));
                    a.w2[k - half] = w2k;
                    if (vi_less(-w2k, k, b2.v, b2.i)) { b2.v = -w2k; b2.i = k; }
                }
            }
        }
    }
    block_sum<2>(hs, red);
    b0 = block_argmin(b0, redvi);
    if constexpr (MULTI) {
        b1 = block_argmin(b1, redvi);
        b2 = block_argmin(b2, redvi);
    }
    // partials: write-through, then a relaxed agent ticket (per-XCD L2s are not coherent)
    __shared__ float s_p[256 * 8];
    __shared__ int s_last;
    float* pf = a.apart + (size_t)blockIdx.x * 8;
    if (tid < 8) {
        const float val = tid == 0 ? hs[0] : tid == 1 ? hs[1] : tid == 2 ? b0.v : tid == 3 ? __int_as_float(b0.i)
                        : tid == 4 ? b1.v : tid == 5 ? __int_as_float(b1.i) : tid == 6 ? b2.v : __int_as_float(b2.i);
        __hip_atomic_store(pf + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&a.wcount[a.T + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == nb - 1;
        if (s_last) a.wcount[a.T + 1] = 0;
    }
    __syncthreads();
    if (!s_last) return;
    for (int q = tid; q < nb * 8; q += AP_T)   // all partials in flight at once
        s_p[q] = __hip_atomic_load(a.apart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (tid == 0) {
        float h0 = 0.0f, h1 = 0.0f;
        VI c0 = {INF, 0x7fffffff}, c1 = c0, c2 = c0;
        for (int b = 0; b < nb; ++b) {  // workgroup order
            const float* x = s_p + b * 8;
            h0 += x[0]; h1 += x[1];
            if (vi_less(x[2], __float_as_int(x[3]), c0.v, c0.i)) { c0.v = x[2]; c0.i = __float_as_int(x[3]); }
            if (vi_less(x[4], __float_as_int(x[5]), c1.v, c1.i)) { c1.v = x[4]; c1.i = __float_as_int(x[5]); }
            if (vi_less(x[6], __float_as_int(x[7]), c2.v, c2.i)) { c2.v = x[6]; c2.i = __float_as_int(x[7]); }
        }
        m3_info* f = a.info;
        // (global indices; a shard without a sample of a subset -- the ranks of the other mode -- has none: -1)
        const int g0 = c0.i == 0x7fffffff ? -1 : a.kbase + c0.i;
        const int g1 = (!MULTI || c1.i == 0x7fffffff) ? -1 : a.kbase + c1.i;
        const int g2 = (!MULTI || c2.i == 0x7fffffff) ? -1 : a.kbase + c2.i;
        f->best_idx = g0;
        f->best_idx_1 = g1;
        f->best_idx_2 = g2;
        f->wsum_push = h0; f->wsum_pull = h1;
        f->pull_preference = h1 > h0;
        if (a.rec_b) {   // shard_mix = 3: the header of this rank's second record (indices < 2^24: exact as floats)
            a.rec_b[0] = c0.v; a.rec_b[1] = (float)g0;
            a.rec_b[2] = c1.v; a.rec_b[3] = (float)g1;
            a.rec_b[4] = c2.v; a.rec_b[5] = (float)g2;
            a.rec_b[6] = h0; a.rec_b[7] = h1;
        }
        if constexpr (!MULTI) {   // what k_weights' single-softmin branch reports
            f->eta = so.eta[0]; f->eta_1 = 0.0f; f->eta_2 = 0.0f;
            f->iters = 1; f->iters_1 = 1; f->iters_2 = 1;
            f->beta_1 = 1.0f; f->beta_2 = 1.0f;
            float nb_ = so.beta[0];
            if (!a.mode_simple && a.env_type == M3_ENV_PANDA) {  // mppi.py:446-454
                if (so.eta[0] > 20.0f) nb_ = nb_ * 0.9f;
                else if (so.eta[0] < 10.0f) nb_ = nb_ * 1.2f;
            }
            if (!a.mode_simple) f->beta = nb_;
        }
    }
}

// local softmin of 4096 costs per workgroup (single mode, K > 16384); the top-k stage-A workgroups ride along
__global__ __launch_bounds__(AP_T) void k_sumexp_single(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    const int Kg = a.Kg, tid = threadIdx.x, nb = (Kg + AP_T * AP_RPT - 1) / (AP_T * AP_RPT);
    if ((int)blockIdx.x >= nb) {
        topk_stage_a(a, blockIdx.x - nb);
        return;
    }
    const float INF = __builtin_inff();
    const float bs = a.mode_simple ? a.lambda_ : a.info->beta;
    const int base = blockIdx.x * AP_T * AP_RPT;
    float v[AP_RPT];
    float mn[1] = {INF};
#pragma unroll
    for (int e = 0; e < AP_RPT; ++e) {
        const int k = base + e * AP_T + tid;
        const float jv = a.Jall[min(k, Kg - 1)];
        v[e] = (k < Kg) ? jv : INF;
        mn[0] = fminf(mn[0], v[e]);
    }
    block_min<1>(mn, red);
    const float nib = -1.0f / bs;
    float es[1] = {0.0f};
#pragma unroll
    for (int e = 0; e < AP_RPT; ++e) es[0] += m3_exp(nib * (v[e] - mn[0]));   // past-the-end rows: exp(-inf) = 0
    block_sum<1>(es, red);
    if (tid == 0) { a.part_min[blockIdx.x * 3 + 0] = mn[0]; a.part_min[blockIdx.x * 3 + 1] = es[0]; }
}

int weights_threads(int Kg) { return Kg <= 8192 ? 256 : WT_MAX; }

void launch_weights(const UpdateArgs& a, hipStream_t s) {
    // few waves for small K: the block-wide reductions dominate there, not the elements/thread
    const int threads = weights_threads(a.Kg);
    UpdateArgs b = a;
    if (threads != 256 && a.multi_modal && !a.mode_simple && apply_workgroups(a.Kg) <= 256) {  // split path
        // (k_mins / k_ladder already launched)
        hipLaunchKernelGGL(k_search, dim3(1 + a.n_cand), dim3(WT_MAX), 0, s, b);
        hipLaunchKernelGGL(k_apply_weights<true>, dim3(apply_workgroups(a.Kg)), dim3(AP_T), 0, s, b);
        return;
    }
    if (!(a.multi_modal && !a.mode_simple) && !a.record && a.Kg > 16384 && apply_workgroups(a.Kg) <= 256 &&
        mins_workgroups(a.Kg) == apply_workgroups(a.Kg)) {
        // single softmin over many costs: local softmins on many workgroups, then the weights pass
        hipLaunchKernelGGL(k_sumexp_single, dim3(apply_workgroups(a.Kg) + a.n_cand), dim3(AP_T), 0, s, b);
        hipLaunchKernelGGL(k_apply_weights<false>, dim3(apply_workgroups(a.Kg)), dim3(AP_T), 0, s, b);
        return;
    }
    if (threads == 256) {
        b.lds_floats = 0;
        hipLaunchKernelGGL((k_weights<32, 256>), dim3(1 + a.n_cand), dim3(256), 0, s, b);
    } else {
        const int rest = a.Kg - 24 * WT_MAX;
        b.lds_floats = rest <= 0 ? 0 : (rest < WEIGHTS_LDS_MAX ? rest : WEIGHTS_LDS_MAX);
        static bool lds_opt_in = false;  // > 64 KB of dynamic LDS needs the attribute once
        if (!lds_opt_in) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_weights<24, WT_MAX>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      WEIGHTS_LDS_MAX * (int)sizeof(float));
            lds_opt_in = true;
        }
        hipLaunchKernelGGL((k_weights<24, WT_MAX>), dim3(1 + a.n_cand), dim3(WT_MAX), b.lds_floats * sizeof(float), s, b);
    }
}

// ---------------------------------------------------------------------------------------
// weighted action sums: sum_k w_k * actions[t][k][:] over the local shard, for the global
// weights and (multi-modal) the two per-mode weight sets; plus row gathers.  Grid = T x n_chunk
// workgroups (+1 for top-k stage B): each reads its slice of the action rows once (all nu
// columns in one pass) and, when n_chunk > 1, the last workgroup to arrive for a time step adds
// the partials in chunk order -- the result does not depend on which one that is.
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
int wsum_chunks(int Kl) { const int L = wsum_chunk_len(Kl); return (Kl + L - 1) / L; }

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
// MULTI (compile time: the three weight sets of the multi-modal update): as a run-time flag the wave-uniform
// `if (multi)` around the two extra weight loads made every sample of the unrolled batch its own basic block ending in
// `s_waitcnt vmcnt(0)` -- EIGHT serialised memory round trips per batch instead of one (round 3, found in the ISA:
// k_wsum 19.8 us at K = 64000 multi-modal, 361 us at K = 1 M single-mode).
template <int NU, bool REGEN, bool MULTI>
__global__ __launch_bounds__(ST) void k_wsum(const UpdateArgs a) {
    __shared__ float red[3 * 16];
    const int tid = threadIdx.x, C = a.n_chunk;
    const int Kl = a.Kl, k0 = a.k0, T = a.T, half = a.Kg / 2;
    if ((int)blockIdx.x == T * C) {  // extra workgroup (n_cand > 1 only): top-k stage B
        topk_stage_b(a);
        return;
    }
    const int t = blockIdx.x / C, c = blockIdx.x % C;
    const float* act = a.actions + (size_t)t * Kl * NU;
    float acc[3][NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.0f;
    RegenRows<NU> rows;
    const float inv_Kls = 1.0f / (float)a.Kls;
    if constexpr (REGEN) regen_rows<NU>(a, t, rows);
    // fixed trip count, clamped unconditional loads: all the loads of a workgroup's slice are in
    // flight together (conditional loads each became a branch region ending in s_waitcnt vmcnt(0))
    const int clen = wsum_chunk_len(Kl);
    const int i1 = min(Kl, (c + 1) * clen);
    const int nh2 = a.Kg - half;
    for (int i0 = c * clen; i0 < i1; i0 += WS_BATCH * ST)
#pragma unroll
    for (int it = 0; it < WS_BATCH; ++it) {
        const int i = i0 + it * ST + tid;
        const bool ok = i < i1;
        const int ic = ok ? i : (i1 - 1);
        const int k = k0 + ic;
        float av[NU];
        if constexpr (REGEN) {
            // (Kl == Kg, k0 == 0 here) the sample's noise row lives in its shard's block
            const int r = shard_of(k, a.Kls, inv_Kls), kk = k - r * a.Kls;
            const float* drow = a.noise_all + (((size_t)r * T + t) * a.Kls + kk) * NU;
            float dv[NU];
            if constexpr (NU == 2) {
                const float2 v = *reinterpret_cast<const float2*>(drow);
                dv[0] = v.x; dv[1] = v.y;
            } else {
#pragma unroll
                for (int j = 0; j < NU; ++j) dv[j] = drow[j];
            }
            regen_action<NU>(a, rows, k, dv, av);
        } else if constexpr (NU == 2) {
            const float2 v = reinterpret_cast<const float2*>(act)[ic];
            av[0] = v.x; av[1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < NU; ++j) av[j] = act[(size_t)ic * NU + j];
        }
        float w = a.w[k];
        float wa = 0.0f, wb = 0.0f;
        if constexpr (MULTI) {
            const float x1 = a.w1[min(k, max(half - 1, 0))];
            const float x2 = a.w2[min(max(k - half, 0), nh2 - 1)];
            wa = (k < half) ? x1 : 0.0f;
            wb = (k < half) ? 0.0f : x2;
        }
        if (!ok) { w = 0.0f; wa = 0.0f; wb = 0.0f; }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            acc[0][j] += w * av[j];
            acc[1][j] += wa * av[j];
            acc[2][j] += wb * av[j];
        }
    }
    float* out = (C == 1) ? a.reduce : a.wpart + (size_t)c * 3 * T * NU;
    // all 3*NU sums through ONE LDS exchange (one pair of barriers) instead of one per column:
    // the nine-column Panda rows paid nine barrier rounds here (~1 us each)
    {
        __shared__ float sred[3 * 9 * (ST / 64)];
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const float ws = wave_sum(acc[s3][j]);
                if (lane == 0) sred[(s3 * NU + j) * (ST / 64) + wv] = ws;
            }
        __syncthreads();
        if (tid < 3 * NU) {
            float rv = 0.0f;
#pragma unroll
            for (int w = 0; w < ST / 64; ++w) rv += sred[tid * (ST / 64) + w];   // wave order, as block_sum
            const int s3 = tid / NU, j = tid % NU;
            float* dst = &out[s3 * T * NU + t * NU + j];  // == reduce_off_psum(s3) + t*NU + j
            if (C == 1 && !a.fuse_finalize) *dst = rv;
            else __hip_atomic_store(dst, rv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: write-through
        }
        __syncthreads();
    }
    // best rows (zero unless the owning rank); before the arrival ticket below, which covers them
    if (c == 0 && tid < 3 * NU) {
        const int which = tid / NU, j = tid % NU;
        const int gi = (which == 0) ? a.info->best_idx : (which == 1 ? a.info->best_idx_1 : a.info->best_idx_2);
        float v = 0.0f;
        const int li = gi - k0;
        if (gi >= 0 && li >= 0 && li < Kl) {
            if constexpr (REGEN) {
                const int r = gi / a.Kls, kk = gi - r * a.Kls;
                const float* drow = a.noise_all + (((size_t)r * T + t) * a.Kls + kk) * NU;
                float dv[NU], ev[NU];
#pragma unroll
                for (int q = 0; q < NU; ++q) dv[q] = drow[q];
                regen_action<NU>(a, rows, gi, dv, ev);
#pragma unroll
                for (int q = 0; q < NU; ++q) if (q == j) v = ev[q];
            } else {
                v = act[(size_t)li * NU + j];
            }
        }
        float* dst = &a.reduce[reduce_off_best(which, T, NU) + t * NU + j];
        if (a.fuse_finalize) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = v;
    }
    bool t_last = true;   // this workgroup completes its time step (always, with one chunk)
    if (C > 1) {
        // in-launch combine.  Per-XCD L2s are not coherent with each other: the partials are
        // written through (sc1 stores) and read back with sc1 loads, so no L2 write-back /
        // invalidate (agent-scope fences cost ~3.5 us per workgroup here) is needed; the ticket
        // is an agent-scope atomic issued after every partial store of the workgroup has retired.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(&a.wcount[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = ticket == C - 1;
            if (is_last) a.wcount[t] = 0;  // re-armed for the next launch (zeroed at m3_create)
            red[47] = __int_as_float(is_last);
        }
        __syncthreads();
        t_last = __float_as_int(red[47]) != 0;
        if (t_last && tid < 3 * NU) {
            const int which = tid / NU, j = tid % NU;
            float sum = 0.0f;
            for (int cc = 0; cc < C; ++cc)
                sum += __hip_atomic_load(&a.wpart[((size_t)cc * 3 + which) * T * NU + t * NU + j],
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float* dst = &a.reduce[reduce_off_psum(which, T, NU) + t * NU + j];
            if (a.fuse_finalize) __hip_atomic_store(dst, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = sum;
        }
    }
    if (a.fuse_finalize && t_last) {
        // Unsharded command(): the mean update / filter (k_finalize's work, T*nu values) is done by
        // the LAST workgroup to finish instead of by one more launch (a dependent launch costs ~3.5 us
        // of turnaround + ~5 us for the one-workgroup kernel).  Same hand-off as the chunk combine:
        // write-through stores, then a relaxed agent ticket -- taken only by the workgroup that
        // completed its time step (T arrivals on this address, not T x n_chunk: 3840 of them at
        // K = 1 M serialised to ~0.4 ms).
        extern __shared__ float sm_fin[];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(&a.wcount[T], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = ticket == T - 1;
            if (is_last) a.wcount[T] = 0;
            red[46] = __int_as_float(is_last);
        }
        __syncthreads();
        if (__float_as_int(red[46])) finalize_body<true>(a, sm_fin);
    }
}
void launch_wsum(const UpdateArgs& a, hipStream_t s) {
    const dim3 grid(a.T * a.n_chunk + (a.n_cand > 1 ? 1 : 0));
    const size_t lds = a.fuse_finalize ? (size_t)a.T * a.nu * sizeof(float) : 0;
    const bool multi = a.multi_modal && !a.mode_simple;
    if (a.regen) {
        if (a.nu == 2) hipLaunchKernelGGL((k_wsum<2, true, true>), grid, dim3(ST), lds, s, a);     // (regen is a multi-modal protocol)
        else hipLaunchKernelGGL((k_wsum<9, true, true>), grid, dim3(ST), lds, s, a);
        return;
    }
    if (a.nu == 2) {
        if (multi) hipLaunchKernelGGL((k_wsum<2, false, true>), grid, dim3(ST), lds, s, a);
        else hipLaunchKernelGGL((k_wsum<2, false, false>), grid, dim3(ST), lds, s, a);
    } else {
        if (multi) hipLaunchKernelGGL((k_wsum<9, false, true>), grid, dim3(ST), lds, s, a);
        else hipLaunchKernelGGL((k_wsum<9, false, false>), grid, dim3(ST), lds, s, a);
    }
}

// ---------------------------------------------------------------------------------------
// shard_mix = 2, after the all-gather and k_search (which mixed the shards' ladder tables into beta, eta and the
// minima): weights, weighted sums over ALL K samples with re-generated actions, the best rows, m3_info and
// the finalize in ONE launch.  Grid = T x n_chunk workgroups as k_wsum.  Every workgroup forms the weights
// of its samples itself from the costs in the gathered records (w = exp(-(J - m) / beta) / eta: the
// expression of k_apply_weights); the chunk workgroups of time step 0 also store them and keep the half sums
// / argmax keys of their chunk; the workgroup that finishes last combines those in chunk order, re-generates
// the three best rows and runs the finalize.
// Two launches: k_regen_part -- T x n_chunk workgroups of 2048 samples each, nothing but partial sums (no arrival
// tickets: with 32 chunks per time step their serialised atomics on one address cost more than the sums) -- and
// k_regen_done, one workgroup that adds the partials in chunk order, combines the half sums / argmax keys,
// re-generates the three best rows and runs the finalize.  (One launch with tickets, 8192-sample chunks: 29 us at
// K = 64000; this pair: see DESIGN.md section 7.)
int regen_chunk_len(int Kg) {          // 2048 = WS_BATCH * ST samples, more beyond 64 chunks per time step
    const int unit = WS_BATCH * ST;
    const int per64 = (((Kg + 63) / 64) + unit - 1) / unit * unit;
    return per64 > unit ? per64 : unit;
}
int regen_chunks(int Kg) { const int L = regen_chunk_len(Kg); return (Kg + L - 1) / L; }

// REGEN = false (round 4): the same kernel for the UNSHARDED multi-modal update with K > 8192 -- costs from the
// rollout's buffer, actions loaded from it, beta / eta / minima as k_ladder_search's search workgroup published them
// (a.srch), top-k stage B as the extra workgroup.
template <int NU, bool REGEN = true>
__global__ __launch_bounds__(ST) void k_regen_part(const UpdateArgs a, const int clen) {
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    __shared__ float sred[3 * 9 * (ST / 64)];
    const int tid = threadIdx.x, C = a.n_chunk, Kg = a.Kg, T = a.T, half = a.half_g;
    if ((int)blockIdx.x == T * C) {   // the extra workgroup: the global top-k from the shards' own lists
        if constexpr (REGEN) topk_merge_records(a);
        else topk_stage_b(a);
        return;
    }
    const int t = blockIdx.x / C, c = blockIdx.x % C;
    const float INF = __builtin_inff();
    const float inv_Kls = 1.0f / (float)a.Kls;
    const int iend = min(Kg, (c + 1) * clen);
    const float* act = a.actions + (size_t)t * Kg * NU;   // (REGEN = false: Kl == Kg, k0 == 0)
    // this workgroup's costs and noise rows (the first batch: all of them up to K = 131072) are requested BEFORE
    // the search, whose table loads and serial walk would otherwise sit in front of their latency
    float v8[WS_BATCH], d8[WS_BATCH][NU];
    auto load_batch = [&](int ib) {
#pragma unroll
        for (int it = 0; it < WS_BATCH; ++it) {
            const int k = min(ib + it * ST + tid, iend - 1);
            const float* drow;
            if constexpr (REGEN) {
                const int r = shard_of(k, a.Kls, inv_Kls), kk = k - r * a.Kls;
                v8[it] = a.records_all[(size_t)r * a.rec_len + kk];
                drow = a.noise_all + (((size_t)r * T + t) * a.Kls + kk) * NU;
            } else {
                v8[it] = a.Jall[k];
                drow = act + (size_t)k * NU;
            }
            if constexpr (NU == 2) {
                const float2 d2 = *reinterpret_cast<const float2*>(drow);
                d8[it][0] = d2.x; d8[it][1] = d2.y;
            } else {
#pragma unroll
                for (int j = 0; j < NU; ++j) d8[it][j] = drow[j];
            }
        }
    };
    load_batch(c * clen);
    // the beta searches on the MIXTURE of the shards' ladder tables, by every workgroup for itself (a few hundred
    // exps and a serial walk: cheaper than a launch of its own in front of this one; same code, same data => the
    // same result in every workgroup); workgroup 0 publishes it
    SearchOut so;
    if constexpr (REGEN) search_body(a, so, blockIdx.x == 0);
    else so = *a.srch;
    const float i0 = uniform_f(1.0f / so.eta[0]), n0 = uniform_f(-1.0f / so.beta[0]);
    const float i1 = uniform_f(1.0f / so.eta[1]), n1 = uniform_f(-1.0f / so.beta[1]);
    const float i2 = uniform_f(1.0f / so.eta[2]), n2 = uniform_f(-1.0f / so.beta[2]);
    RegenRows<NU> rows;
    if constexpr (REGEN) regen_rows<NU>(a, t, rows);
    float acc[3][NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) acc[0][j] = acc[1][j] = acc[2][j] = 0.0f;
    float hs[2] = {0.0f, 0.0f};
    VI b0 = {INF, 0x7fffffff}, b1 = {INF, 0x7fffffff}, b2 = {INF, 0x7fffffff};
    for (int ib = c * clen; ib < iend; ib += WS_BATCH * ST) {
    if (ib != c * clen) load_batch(ib);
#pragma unroll
    for (int it = 0; it < WS_BATCH; ++it) {
        const int i = ib + it * ST + tid;
        const bool ok = i < iend;
        const int k = ok ? i : (iend - 1);
        const float v = v8[it];
        float dv[NU], av[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) dv[j] = d8[it][j];
        if constexpr (REGEN) regen_action<NU>(a, rows, k, dv, av);
        else {
#pragma unroll
            for (int j = 0; j < NU; ++j) av[j] = dv[j];
        }
        const bool first = k < half;
        float w = i0 * m3_exp(n0 * (v - so.mn[0]));
        float wh = (first ? i1 : i2) * m3_exp((first ? n1 : n2) * (v - (first ? so.mn[1] : so.mn[2])));
        if (!ok) { w = 0.0f; wh = 0.0f; }
        const float wa = first ? wh : 0.0f, wb = first ? 0.0f : wh;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            acc[0][j] += w * av[j];
            acc[1][j] += wa * av[j];
            acc[2][j] += wb * av[j];
        }
        if (t == 0 && ok) {   // (workgroup-uniform on t) the weights themselves, half sums, argmax keys
            a.w[k] = w;
            if (first) a.w1[k] = wh; else a.w2[k - half] = wh;
            hs[0] += first ? w : 0.0f;
            hs[1] += first ? 0.0f : w;
            if (vi_less(-w, k, b0.v, b0.i)) { b0.v = -w; b0.i = k; }
            if (first) { if (vi_less(-wh, k, b1.v, b1.i)) { b1.v = -wh; b1.i = k; } }
            else { if (vi_less(-wh, k, b2.v, b2.i)) { b2.v = -wh; b2.i = k; } }
        }
    }
    }
    if (t == 0) {
        block_sum<2>(hs, red);
        b0 = block_argmin(b0, redvi);
        b1 = block_argmin(b1, redvi);
        b2 = block_argmin(b2, redvi);
        if (tid < 8) {
            const float val = tid == 0 ? hs[0] : tid == 1 ? hs[1] : tid == 2 ? b0.v : tid == 3 ? __int_as_float(b0.i)
                            : tid == 4 ? b1.v : tid == 5 ? __int_as_float(b1.i) : tid == 6 ? b2.v : __int_as_float(b2.i);
            a.apart[(size_t)c * 8 + tid] = val;
        }
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const float ws = wave_sum(acc[s3][j]);
            if (lane == 0) sred[(s3 * NU + j) * (ST / 64) + wv] = ws;
        }
    __syncthreads();
    if (tid < 3 * NU) {
        float rv = 0.0f;
#pragma unroll
        for (int w = 0; w < ST / 64; ++w) rv += sred[tid * (ST / 64) + w];
        const int s3 = tid / NU, j = tid % NU;
        a.wpart[((size_t)c * 3 + s3) * T * NU + t * NU + j] = rv;
    }
}

template <int NU, bool REGEN = true>
__global__ __launch_bounds__(ST) void k_regen_done(const UpdateArgs a) {
    extern __shared__ float sm_fin[];
    __shared__ int s_best[3];
    __shared__ float s_part[64 * 8];
    const int tid = threadIdx.x, C = a.n_chunk, Kg = a.Kg, T = a.T;
    const float INF = __builtin_inff();
    // (every load of this workgroup is a first touch of a line another workgroup wrote: they are issued eight at
    // a time and added afterwards, in chunk order -- a dependent load per chunk was 20 us of latency here)
    for (int o = tid; o < C * 8; o += ST) s_part[o] = a.apart[o];
    // (a) the partial sums in chunk order
    for (int o = tid; o < 3 * T * NU; o += ST) {
        const int which = o / (T * NU), rem = o - which * T * NU;
        float sum = 0.0f;
        for (int c0 = 0; c0 < C; c0 += 8) {
            float p[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cc = min(c0 + q, C - 1);
                p[q] = a.wpart[((size_t)cc * 3 + which) * T * NU + rem];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += (c0 + q < C) ? p[q] : 0.0f;
        }
        a.reduce[reduce_off_psum(which, T, NU) + rem] = sum;
    }
    __syncthreads();
    // (b) half sums and argmax keys of time step 0's chunks: lane cc of the first wavefront holds chunk cc
    // (C <= 64), fixed reduction tree
    if (tid < 64) {
        const bool on = tid < C;
        const float* x = s_part + (size_t)(on ? tid : 0) * 8;
        const float h0 = wave_sum(on ? x[0] : 0.0f), h1 = wave_sum(on ? x[1] : 0.0f);
        const VI none = {INF, 0x7fffffff};
        const VI c0 = wave_argmin(on ? VI{x[2], __float_as_int(x[3])} : none);
        const VI c1 = wave_argmin(on ? VI{x[4], __float_as_int(x[5])} : none);
        const VI c2 = wave_argmin(on ? VI{x[6], __float_as_int(x[7])} : none);
        if (tid == 0) {
            m3_info* f = a.info;
            f->best_idx = c0.i; f->best_idx_1 = c1.i; f->best_idx_2 = c2.i;
            f->wsum_push = h0; f->wsum_pull = h1;
            f->pull_preference = h1 > h0;
            s_best[0] = c0.i; s_best[1] = c1.i; s_best[2] = c2.i;
        }
    }
    __syncthreads();
    // (c) the best rows: actions of the three argmax samples, re-generated for every time step
    for (int o = tid; o < 3 * T; o += ST) {
        const int which = o / T, tt = o - which * T, gi = s_best[which];
        float dv[NU], ev[NU];
        const bool valid = gi >= 0 && gi < Kg;     // (no argmax at all when every weight is NaN: zero rows then)
        const int gc = valid ? gi : 0;
        if constexpr (REGEN) {
            RegenRows<NU> rr;
            regen_rows<NU>(a, tt, rr);
            const int r = gc / a.Kls, kk = gc - r * a.Kls;
            const float* drow = a.noise_all + (((size_t)r * T + tt) * a.Kls + kk) * NU;
#pragma unroll
            for (int j = 0; j < NU; ++j) dv[j] = drow[j];
            regen_action<NU>(a, rr, gc, dv, ev);
        } else {
            const float* arow = a.actions + ((size_t)tt * Kg + gc) * NU;
#pragma unroll
            for (int j = 0; j < NU; ++j) ev[j] = arow[j];
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) a.reduce[reduce_off_best(which, T, NU) + tt * NU + j] = valid ? ev[j] : 0.0f;
    }
    __threadfence_block();
    __syncthreads();
    finalize_body<false>(a, sm_fin);
}
void launch_fused_large(const UpdateArgs& a_, hipStream_t s) {   // (after launch_ladder_search)
    UpdateArgs a = a_;
    const int clen = regen_chunk_len(a.Kg);
    a.n_chunk = regen_chunks(a.Kg);
    const dim3 grid(a.T * a.n_chunk + (a.n_cand > 1 ? 1 : 0));
    const size_t lds = (size_t)a.T * a.nu * sizeof(float);
    if (a.nu == 2) {
        hipLaunchKernelGGL((k_regen_part<2, false>), grid, dim3(ST), 0, s, a, clen);
        hipLaunchKernelGGL((k_regen_done<2, false>), dim3(1), dim3(ST), lds, s, a);
    } else {
        hipLaunchKernelGGL((k_regen_part<9, false>), grid, dim3(ST), 0, s, a, clen);
        hipLaunchKernelGGL((k_regen_done<9, false>), dim3(1), dim3(ST), lds, s, a);
    }
}
void launch_regen_fast(const UpdateArgs& a_, hipStream_t s) {
    UpdateArgs a = a_;
    const int clen = regen_chunk_len(a.Kg);
    a.n_chunk = regen_chunks(a.Kg);
    const dim3 grid(a.T * a.n_chunk + 1);   // + the top-k merge
    const size_t lds = (size_t)a.T * a.nu * sizeof(float);
    if (a.nu == 2) {
        hipLaunchKernelGGL(k_regen_part<2>, grid, dim3(ST), 0, s, a, clen);
        hipLaunchKernelGGL(k_regen_done<2>, dim3(1), dim3(ST), lds, s, a);
    } else {
        hipLaunchKernelGGL(k_regen_part<9>, grid, dim3(ST), 0, s, a, clen);
        hipLaunchKernelGGL(k_regen_done<9>, dim3(1), dim3(ST), lds, s, a);
    }
}

// ---- shard_mix = 3: two small exchanges, O(K_local) work per rank after the first --------------------------------
//   Before exchange A: the shard_mix = 2 record (k_local_topk: the shard's costs, top-k,
// minima and ladder table).  After it: the searches on the MIXTURE of the tables (k_search with a.fast; passes over
// the gathered costs only if a search leaves its ladder), then the weights of the rank's OWN samples with the global
// minima / eta (k_apply_weights<true> over the local costs) and their weighted action sums from the rank's own action
// buffer (k_wsum) -- nothing is re-generated, nothing of size K_global is touched.  Exchange B gathers the ranks'
// sums, best rows and (-w, index) pairs; k_p3_done adds the sums in rank order, takes the best rows from the rank
// whose best sample wins (the unsharded argmax: largest weight, lowest index), and writes the plan.
template <int NU>
__global__ __launch_bounds__(ST) void k_p3_done(const UpdateArgs a) {
    extern __shared__ float sm_fin[];
    __shared__ int s_win[3];
    const int tid = threadIdx.x, N = a.n_ranks, T = a.T, L = a.recb_len;
    if (tid == 0) {
        const float INF = __builtin_inff();
        float h0 = 0.0f, h1 = 0.0f;
        VI best[3] = {{INF, 0x7fffffff}, {INF, 0x7fffffff}, {INF, 0x7fffffff}};
        int win[3] = {0, 0, 0};
        for (int r = 0; r < N; ++r) {   // rank order
            const float* x = a.recb_all + (size_t)r * L;
            h0 += x[6]; h1 += x[7];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int gi = (int)x[2 * sx + 1];
                if (gi >= 0 && vi_less(x[2 * sx], gi, best[sx].v, best[sx].i)) { best[sx].v = x[2 * sx]; best[sx].i = gi; win[sx] = r; }
            }
        }
        m3_info* f = a.info;
        f->best_idx = best[0].i == 0x7fffffff ? -1 : best[0].i;
        f->best_idx_1 = best[1].i == 0x7fffffff ? -1 : best[1].i;
        f->best_idx_2 = best[2].i == 0x7fffffff ? -1 : best[2].i;
        f->wsum_push = h0; f->wsum_pull = h1;
        f->pull_preference = h1 > h0;
        s_win[0] = win[0]; s_win[1] = win[1]; s_win[2] = win[2];
    }
    __syncthreads();
    const int n = T * NU;
    for (int o = tid; o < 3 * n; o += ST) {
        const int which = o / n, rem = o - which * n;
        float sum = 0.0f;
        for (int r = 0; r < N; ++r) sum += a.recb_all[(size_t)r * L + RECB_HDR + which * n + rem];
        a.reduce[reduce_off_psum(which, T, NU) + rem] = sum;
        a.reduce[reduce_off_best(which, T, NU) + rem] = a.recb_all[(size_t)s_win[which] * L + RECB_HDR + (3 + which) * n + rem];
    }
    __threadfence_block();
    __syncthreads();
    finalize_body<false>(a, sm_fin);
}
void launch_p3_search(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_search, dim3(2), dim3(WT_MAX), 0, s, a);     // workgroup 1: the global top-k from the shards' lists
}
void launch_p3_local_weights(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_weights<true>, dim3(apply_workgroups(a.Kg)), dim3(AP_T), 0, s, a);
}
void launch_p3_done(const UpdateArgs& a, hipStream_t s) {
    const size_t lds = (size_t)a.T * a.nu * sizeof(float);
    if (a.nu == 2) hipLaunchKernelGGL(k_p3_done<2>, dim3(1), dim3(ST), lds, s, a);
    else hipLaunchKernelGGL(k_p3_done<9>, dim3(1), dim3(ST), lds, s, a);
}

// the shard's own top-k before the collective ("regen" sharding): stage A per 4096 costs, the last
// workgroup to finish merges (as the top-k workgroups of k_update_small)
constexpr int LREC_WG = 96;   // extra workgroups of the pre-gather launch that evaluate the shard's ladder table: one per ladder point (32 of them, three points each: 11.4 us at 8000 costs)
template <int RPT>
__global__ __launch_bounds__(PREP_T) void k_local_topk(const UpdateArgs a) {
    __shared__ int s_lastb;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.n_cand) {
        // shard_mix = 2: this shard's minima (all / mode 1 / mode 2) and its eta(beta) sums relative to them for
        // the ladder points w, w + LREC_WG, ...
        __shared__ float red[3 * 16];
        const int w = blockIdx.x - a.n_cand, Kn = a.Kg, half = a.half_g - a.kbase;   // k < half <=> mode 1
        const float INF = __builtin_inff();
        // up to 8192 costs live in registers (32 rows of 256, all loads in flight at once); beyond that
        // they are re-read from memory (L2) per ladder point, eight loads in flight
        constexpr int LR = 32;
        const bool in_regs = Kn <= LR * PREP_T;
        float rv[LR];
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < LR; ++e) {
                const int k = e * PREP_T + tid;
                const float jv = a.Jall[min(k, Kn - 1)];
                rv[e] = (k < Kn) ? jv : INF;
            }
        }
        float mn[3] = {INF, INF, INF};
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < LR; ++e) {
                const bool first = e * PREP_T + tid < half;
                mn[0] = fminf(mn[0], rv[e]);
                mn[1] = fminf(mn[1], first ? rv[e] : INF);
                mn[2] = fminf(mn[2], first ? INF : rv[e]);
            }
        } else {
            for (int k0 = 0; k0 < Kn; k0 += 8 * PREP_T) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = a.Jall[min(k0 + u * PREP_T + tid, Kn - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * PREP_T + tid;
                    const float x = (k < Kn) ? v[u] : INF;
                    mn[0] = fminf(mn[0], x);
                    if (k < half) mn[1] = fminf(mn[1], x); else mn[2] = fminf(mn[2], x);
                }
            }
        }
        block_min<3>(mn, red);
        if (w == 0 && tid < 4) a.rec_mins[tid] = tid < 3 ? mn[tid] : 0.0f;
        for (int p = w; p < LAD_N; p += LREC_WG) {
            const float nib = uniform_f(-1.0f / ladder_beta(p));
            float e3[3] = {0.0f, 0.0f, 0.0f};
            if (in_regs) {
#pragma unroll
                for (int e = 0; e < LR; ++e) {   // rows past the end hold +inf: exp(-inf) = 0 ...
                    const int k = e * PREP_T + tid;
                    const bool ok = k < Kn, first = k < half;
                    e3[0] += m3_exp(nib * (rv[e] - mn[0]));
                    // ... but not against the +inf minimum of a mode this shard has no sample of (inf - inf)
                    const float xh = m3_exp(nib * (rv[e] - (first ? mn[1] : mn[2])));
                    e3[1] += (ok && first) ? xh : 0.0f;
                    e3[2] += (ok && !first) ? xh : 0.0f;
                }
            } else {
                for (int k0 = 0; k0 < Kn; k0 += 8 * PREP_T) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = a.Jall[min(k0 + u * PREP_T + tid, Kn - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = k0 + u * PREP_T + tid;
                        const bool ok = k < Kn, first = k < half;
                        const float x0 = m3_exp(nib * (v[u] - mn[0]));
                        const float xh = m3_exp(nib * (v[u] - (first ? mn[1] : mn[2])));
                        e3[0] += ok ? x0 : 0.0f;
                        e3[1] += (ok && first) ? xh : 0.0f;
                        e3[2] += (ok && !first) ? xh : 0.0f;
                    }
                }
            }
            block_sum<3>(e3, red);
            if (tid < 3) a.rec_table[p * 3 + tid] = e3[tid];
            __syncthreads();
        }
        return;
    }
    topk_stage_a<RPT>(a, blockIdx.x);
    if (a.n_cand > 1) {
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(&a.wcount[a.T + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_lastb = ticket == a.n_cand - 1;
            if (s_lastb) a.wcount[a.T + 1] = 0;
        }
        __syncthreads();
        if (s_lastb) {
            __threadfence();
            topk_stage_b(a);
        }
    }
}
void launch_local_topk(const UpdateArgs& a, hipStream_t s) {
    // up to 8192 costs: ONE workgroup with 32 rows per thread (no second stage, no ticket)
    const int extra = a.fast ? LREC_WG : 0;   // + the ladder-table workgroups
    if (a.n_cand == 1 && a.Kg > PREP_T * 16) hipLaunchKernelGGL(k_local_topk<32>, dim3(1 + extra), dim3(PREP_T), 0, s, a);
    else hipLaunchKernelGGL(k_local_topk<16>, dim3(a.n_cand + extra), dim3(PREP_T), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// Unsharded command() with K <= 4096 (C2, C3, C4): the whole update in ONE launch.
// The softmin over <= 4096 costs is a few microseconds of work for one workgroup but ~7 us as its
// own launch (dispatch + first-load latency + its reductions, all exposed between the rollout and
// the next command) -- and the multi-modal search was three launches (k_mins, k_ladder, k_weights:
// 29 us at K = 4000, most of it dispatch).  Here every one of the T column workgroups of the
// weighted sums does the softmin itself: costs AND the workgroup's action rows are loaded together
// into registers; min / sum-of-exps / argmax go through the same block reductions with the same
// element -> thread mapping as k_weights (single mode: identical eta and weights); the multi-modal
// beta searches run the reference's rule directly (m3p2i.py:24-64), all three side by side, one
// register pass + one block reduction per iteration (~0.7 us; the ladder of k_ladder only pays
// when the costs do not fit one workgroup's registers); the sums accumulate in k_wsum's order.
// Workgroup 0 also stores the weights and m3_info, workgroup T is the top-k stage, the last
// workgroup to finish does the mean update / filter (same hand-off as in k_wsum) and writes the
// adapted beta -- after every workgroup has read the old one.
template <int NU, bool MULTI, int JR, int WT = 256>
__global__ __launch_bounds__(WT) void k_update_small(const UpdateArgs a) {
    constexpr int NS = MULTI ? 3 : 1, NW = WT / 64;   // JR rows of WT costs per thread: K <= JR * WT
    __shared__ float red[3 * 16];
    __shared__ VI redvi[16];
    __shared__ float sred[3 * 9 * (WT / 64)];
    __shared__ float s_part[2][3 * NW];
    const int T = a.T, tid = threadIdx.x, Kg = a.Kg;
    if ((int)blockIdx.x >= T) {  // top-k workgroups, concurrent with the column workgroups
        // one per 4096 costs; with more than one, the last of them to finish merges the lists (stage
        // B): candidates out through agent-scope fences (off the command's critical path), a ticket
        __shared__ int s_lastb;
        if (tid >= PREP_T) return;   // (512-thread instances: the top-k stage is written for PREP_T threads ...
        if constexpr (WT > PREP_T) topk_stage_a<32>(a, blockIdx.x - T);   // ... and ONE workgroup selects from all K <= 8192 costs)
        else topk_stage_a(a, blockIdx.x - T);
        if (a.n_cand > 1) {
            __threadfence();
            __syncthreads();
            if (tid == 0) {
                const int ticket = __hip_atomic_fetch_add(&a.wcount[T + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_lastb = ticket == a.n_cand - 1;
                if (s_lastb) a.wcount[T + 1] = 0;
            }
            __syncthreads();
            if (s_lastb) {
                __threadfence();
                topk_stage_b(a);
            }
        }
        return;
    }
    const int t = blockIdx.x;
    const float INF = __builtin_inff();
    const float* J = a.Jall;
    const float* act = a.actions + (size_t)t * Kg * NU;
    const int half = a.half_g - a.kbase;
    const float b_in = a.mode_simple ? a.lambda_ : a.info->beta;
    // Every loop over the register rows below is fully unrolled and branch-free (invalid rows
    // contribute through selects): one basic block, so the scheduler can overlap the rows' exp
    // sequences -- with a wave-uniform early exit per row each row was its own block and its
    // ~10-deep dependent chain ran alone at ~8 cycles per instruction (2.5 us per search pass).
    float jr[JR], av[JR][NU];
    bool valid[JR];
#pragma unroll
    for (int e = 0; e < JR; ++e) {   // unconditional clamped loads: all in flight together
        const int k = e * WT + tid;
        const int kc = min(k, Kg - 1);
        const float jv = J[kc];
        valid[e] = k < Kg;
        jr[e] = valid[e] ? jv : INF;
        if constexpr (NU == 2) {
            const float2 v = reinterpret_cast<const float2*>(act)[kc];
            av[e][0] = v.x; av[e][1] = v.y;
        } else {
#pragma unroll
            for (int j = 0; j < NU; ++j) av[e][j] = act[(size_t)kc * NU + j];
        }
    }
    // ---- minima ----
    float mn[3] = {INF, INF, INF};
#pragma unroll
    for (int e = 0; e < JR; ++e) {
        const float v = jr[e];
        mn[0] = fminf(mn[0], v);
        if constexpr (MULTI) {
            const bool first = e * WT + tid < half;
            mn[1] = fminf(mn[1], first ? v : INF);
            mn[2] = fminf(mn[2], first ? INF : v);
        }
    }
    block_min<3>(mn, red);
    // ---- beta / eta ----
    float beta[3] = {MULTI ? 1.0f : b_in, 1.0f, 1.0f}, eta[3] = {0.0f, 0.0f, 0.0f};
    int iters[3] = {1, 1, 1};
    if constexpr (!MULTI) {
        float es[1] = {0.0f};
        const float nib = -1.0f / b_in;
#pragma unroll
        for (int e = 0; e < JR; ++e) {
            const float x = m3_exp(nib * (jr[e] - mn[0]));
            es[0] += valid[e] ? x : 0.0f;
        }
        block_sum<1>(es, red);
        eta[0] = es[0];
    } else {
        // every search starts at beta = 1 (beta / beta_1 / beta_2 are never written back: m3p2i.py:58-60).
        // One pass = 2 exps per cost (the half's beta / minimum by select), three wave sums, ONE
        // barrier (double-buffered partials); every thread then applies the rule to its own copy of
        // (beta, eta, done) -- identical in all threads, so no second exchange.
        int done[3] = {0, 0, 0};
        iters[0] = iters[1] = iters[2] = 0;
        const int lane = tid & 63, wv = tid >> 6;
        // (a) The betas a search can visit before it reverses are the ladders {0.9^j}, {1.2^j}: the T
        // column workgroups would all walk them one pass at a time, each computing the same sums.
        // Instead workgroup t evaluates ladder point(s) t, t + T, ... for all three searches, the
        // workgroups exchange the table through memory (write-through stores, one arrive counter,
        // L2-coherent loads: the T + 1 workgroups of this launch are co-resident, 256 CUs), and every
        // workgroup walks the table -- eta(beta) is formed by the same code in the same order as in a
        // pass, so the walk makes the same decisions.  A search that leaves the ladder or reverses
        // continues with the passes below.  (C3: ~16 passes of 1.4 us -> one + ~2 us of exchange.)
        if (T <= 256) {   // (co-residency of the T + 1 workgroups is what the wait relies on)
            constexpr int LS = 16, LG = 24, NPT = LS + LG;   // 0.9^0 .. 0.9^15, 1.2^1 .. 1.2^24
            __shared__ float s_tab[NPT * 3];
            __shared__ float s_walk[3][4];
            int nbuf = 0;
            for (int p = t; p < NPT; p += T, ++nbuf) {
                const float bp = ladder_beta(p < LS ? p : LAD_S + (p - LS));   // 0.9^p / 1.2^(p - LS + 1)
                const float np_ = uniform_f(-1.0f / bp);
                float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
#pragma unroll
                for (int e = 0; e < JR; ++e) e0 += m3_exp(np_ * (jr[e] - mn[0]));
#pragma unroll
                for (int e = 0; e < JR; ++e) {
                    const bool first = e * WT + tid < half;
                    const float xh = m3_exp(np_ * (jr[e] - (first ? mn[1] : mn[2])));
                    e1 += first ? xh : 0.0f;
                    e2 += first ? 0.0f : xh;
                }
                e0 = wave_sum(e0); e1 = wave_sum(e1); e2 = wave_sum(e2);
                float* buf = s_part[nbuf & 1];
                if (lane == 0) { buf[0 * NW + wv] = e0; buf[1 * NW + wv] = e1; buf[2 * NW + wv] = e2; }
                __syncthreads();
                if (tid < 3) {
                    float et = 0.0f;
#pragma unroll
                    for (int w = 0; w < WT / 64; ++w) et += buf[tid * NW + w];   // wave order, as in a pass
                    __hip_atomic_store(&a.lad[p * 3 + tid], et, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // arrive + wait (the counter is re-armed by the last workgroup of the launch, below)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                // bounded wait (~20 ms): a workgroup that gives up simply runs all its passes itself,
                // which makes the same decisions -- the exchange can cost time, never a hang
                __hip_atomic_fetch_add(&a.wcount[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0, ok = 1;
                while (__hip_atomic_load(&a.wcount[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > a.ladder_spins) { ok = 0; break; }
                }
                if (a.ladder_spins == 0) ok = 0;   // (tests: force the give-up branch even when everyone has arrived)
                s_walk[0][0] = __int_as_float(ok);
            }
            __syncthreads();
            const bool have_table = __float_as_int(s_walk[0][0]) != 0;
            __syncthreads();
            for (int o = tid; o < NPT * 3; o += WT)
                s_tab[o] = __hip_atomic_load(&a.lad[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid < 3 && !have_table) {
                s_walk[tid][0] = 1.0f; s_walk[tid][1] = 0.0f; s_walk[tid][2] = __int_as_float(0); s_walk[tid][3] = __int_as_float(0);
            }
            if (tid < 3 && have_table) {   // the reference's rule on the table (m3p2i.py:24-64)
                const int sx = tid;
                float b = 1.0f, et = s_tab[0 * 3 + sx];
                int it = 1, dn = 0;
                if (et > 10.0f) {
                    int j = 0;
                    for (;;) {
                        b = b * 0.9f; ++j;
                        if (j >= LS) break;                    // off the ladder: passes below
                        et = s_tab[j * 3 + sx]; ++it;
                        if (et > 10.0f) continue;
                        if (et < 3.0f) b = b * 1.2f;           // overshoot: reversal, passes below
                        else dn = 1;
                        break;
                    }
                } else if (et < 3.0f) {
                    int j = 0;
                    for (;;) {
                        b = b * 1.2f; ++j;
                        if (j > LG) break;
                        et = s_tab[(LS + j - 1) * 3 + sx]; ++it;
                        if (et < 3.0f) continue;
                        if (et > 10.0f) b = b * 0.9f;
                        else dn = 1;
                        break;
                    }
                } else {
                    dn = 1;
                }
                s_walk[sx][0] = b; s_walk[sx][1] = et; s_walk[sx][2] = __int_as_float(dn); s_walk[sx][3] = __int_as_float(it);
            }
            __syncthreads();
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                beta[s3] = s_walk[s3][0]; eta[s3] = s_walk[s3][1];
                done[s3] = __float_as_int(s_walk[s3][2]); iters[s3] = __float_as_int(s_walk[s3][3]);
            }
            __syncthreads();
        }
        // (b) passes for what the ladder did not settle
        for (int pass = 0; pass < 1000; ++pass) {
            if (done[0] && done[1] && done[2]) break;
            // (quotients behind an optimisation barrier: otherwise the compiler rewrites the per-row
            // select between two quotients as a division by a selected beta -- 16 IEEE divisions per pass)
            const float n0 = uniform_f(-1.0f / beta[0]), n1 = uniform_f(-1.0f / beta[1]), n2 = uniform_f(-1.0f / beta[2]);
            float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f;
            if (!done[0]) {   // (uniform) a finished search costs nothing more
#pragma unroll
                for (int e = 0; e < JR; ++e)   // rows past the end hold +inf: exp(-inf) = 0, no select needed
                    e0 += m3_exp(n0 * (jr[e] - mn[0]));
            }
            if (!(done[1] && done[2])) {
#pragma unroll
                for (int e = 0; e < JR; ++e) {
                    const bool first = e * WT + tid < half;
                    const float xh = m3_exp((first ? n1 : n2) * (jr[e] - (first ? mn[1] : mn[2])));
                    e1 += first ? xh : 0.0f;
                    e2 += first ? 0.0f : xh;
                }
            }
            e0 = wave_sum(e0); e1 = wave_sum(e1); e2 = wave_sum(e2);
            float* buf = s_part[pass & 1];
            if (lane == 0) { buf[0 * NW + wv] = e0; buf[1 * NW + wv] = e1; buf[2 * NW + wv] = e2; }
            __syncthreads();
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                float et = 0.0f;
#pragma unroll
                for (int w = 0; w < WT / 64; ++w) et += buf[s3 * NW + w];   // wave order, as block_sum
                if (!done[s3]) {
                    eta[s3] = et;
                    iters[s3] += 1;
                    if (et > 10.0f) beta[s3] = beta[s3] * 0.9f;
                    else if (et < 3.0f) beta[s3] = beta[s3] * 1.2f;
                    else done[s3] = 1;
                }
            }
        }
        __syncthreads();
    }
    // ---- weights, half sums, argmax, weighted sums of this workgroup's time step ----
    const float i0 = uniform_f(1.0f / eta[0]), n0 = uniform_f(-1.0f / beta[0]);
    const float i1 = uniform_f(1.0f / eta[1]), n1 = uniform_f(-1.0f / beta[1]);
    const float i2 = uniform_f(1.0f / eta[2]), n2 = uniform_f(-1.0f / beta[2]);
    float hs[2] = {0.0f, 0.0f};
    VI bi[3] = {{INF, 0x7fffffff}, {INF, 0x7fffffff}, {INF, 0x7fffffff}};
    float acc[NS][NU], wk[JR], wh[MULTI ? JR : 1];
#pragma unroll
    for (int s3 = 0; s3 < NS; ++s3)
#pragma unroll
        for (int j = 0; j < NU; ++j) acc[s3][j] = 0.0f;
#pragma unroll
    for (int e = 0; e < JR; ++e) {
        const int k = e * WT + tid;
        const bool ok = valid[e], first = k < half;
        const float v = jr[e];
        const float x = i0 * m3_exp(n0 * (v - mn[0]));
        wk[e] = ok ? x : 0.0f;
        hs[0] += (ok && first) ? x : 0.0f;
        hs[1] += (ok && !first) ? x : 0.0f;
        {   // argmax of the weights, first index on ties: key = -w
            const bool take = ok && vi_less(-x, k, bi[0].v, bi[0].i);
            bi[0].v = take ? -x : bi[0].v; bi[0].i = take ? k : bi[0].i;
        }
        float wa = 0.0f, wb = 0.0f;
        if constexpr (MULTI) {
            const float xh = (first ? i1 : i2) * m3_exp((first ? n1 : n2) * (v - (first ? mn[1] : mn[2])));
            wh[e] = xh;
            wa = (ok && first) ? xh : 0.0f;
            wb = (ok && !first) ? xh : 0.0f;
            const bool t1 = ok && first && vi_less(-xh, k, bi[1].v, bi[1].i);
            bi[1].v = t1 ? -xh : bi[1].v; bi[1].i = t1 ? k : bi[1].i;
            const bool t2 = ok && !first && vi_less(-xh, k, bi[2].v, bi[2].i);
            bi[2].v = t2 ? -xh : bi[2].v; bi[2].i = t2 ? k : bi[2].i;
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            acc[0][j] += wk[e] * av[e][j];
            if constexpr (MULTI) { acc[1][j] += wa * av[e][j]; acc[2][j] += wb * av[e][j]; }
        }
    }
    if (t == 0) {   // workgroup-uniform: this workgroup also stores the weights and the half sums
#pragma unroll
        for (int e = 0; e < JR; ++e) {
            const int k = e * WT + tid;
            if (valid[e]) {
                a.w[k] = wk[e];
                if constexpr (MULTI) {
                    if (k < half) a.w1[k] = wh[e];
                    else a.w2[k - half] = wh[e];
                }
            }
        }
        block_sum<2>(hs, red);
    }
    bi[0] = block_argmin(bi[0], redvi);
    if constexpr (MULTI) {
        bi[1] = block_argmin(bi[1], redvi);
        bi[2] = block_argmin(bi[2], redvi);
    }
    float nb = beta[0];
    if (!MULTI && !a.mode_simple && a.env_type == M3_ENV_PANDA) {  // mppi.py:446-454
        if (eta[0] > 20.0f) nb = nb * 0.9f;
        else if (eta[0] < 10.0f) nb = nb * 1.2f;
    }
    if (t == 0 && tid == 0) {
        m3_info* f = a.info;
        f->eta = eta[0]; f->eta_1 = eta[1]; f->eta_2 = eta[2];
        f->iters = iters[0]; f->iters_1 = iters[1]; f->iters_2 = iters[2];
        f->best_idx = a.kbase + bi[0].i;
        f->best_idx_1 = MULTI ? bi[1].i : -1;
        f->best_idx_2 = MULTI ? bi[2].i : -1;
        f->wsum_push = hs[0]; f->wsum_pull = hs[1];
        f->pull_preference = hs[1] > hs[0];
        f->beta_1 = beta[1]; f->beta_2 = beta[2];
        if (a.record) {  // shard_mix: local softmin only; k_mix owns eta, beta and the best index
            a.record[0] = mn[0]; a.record[1] = eta[0];
            a.record[2] = hs[0]; a.record[3] = hs[1];
            a.record[4] = __int_as_float(a.kbase + bi[0].i);
        }
    }
    // ---- column sums through one LDS exchange (k_wsum) ----
    {
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3)
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const float ws = wave_sum(acc[s3][j]);
                if (lane == 0) sred[(s3 * NU + j) * (WT / 64) + wv] = ws;
            }
        __syncthreads();
        if (tid < 3 * NU) {
            const int s3 = tid / NU, j = tid % NU;
            float rv = 0.0f;
            if (s3 < NS) {
#pragma unroll
                for (int w = 0; w < WT / 64; ++w) rv += sred[tid * (WT / 64) + w];
            }
            __hip_atomic_store(&a.reduce[reduce_off_psum(s3, T, NU) + t * NU + j], rv, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            if (s3 >= NS)  // no per-mode best rows in single mode
                __hip_atomic_store(&a.reduce[reduce_off_best(s3, T, NU) + t * NU + j], 0.0f, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        // best rows: the thread that holds a best sample's action writes it
#pragma unroll
        for (int e = 0; e < JR; ++e) {
            const int k = e * WT + tid;
#pragma unroll
            for (int s3 = 0; s3 < NS; ++s3)
                if (k == bi[s3].i) {
#pragma unroll
                    for (int j = 0; j < NU; ++j)
                        __hip_atomic_store(&a.reduce[reduce_off_best(s3, T, NU) + t * NU + j], av[e][j],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
    }
    if (!a.fuse_finalize) return;   // sharded (shard_mix): the record goes to the collective, k_mix + k_finalize follow
    // ---- last workgroup: mean update / filter, adapted beta ----
    extern __shared__ float sm_fin[];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&a.wcount[T], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int is_last = ticket == T - 1;
        if (is_last) a.wcount[T] = 0;
        red[46] = __int_as_float(is_last);
    }
    __syncthreads();
    if (__float_as_int(red[46])) {
        if (tid == 0 && !MULTI && !a.mode_simple) a.info->beta = nb;
        if (tid == 0 && MULTI) a.wcount[0] = 0;   // the ladder exchange's arrive counter, for the next launch
        finalize_body<true>(a, sm_fin);
    }
}
void launch_update_small(const UpdateArgs& a_, hipStream_t s) {
    // bounded wait of the in-launch ladder exchange (~20 ms); M3P2I_LADDER_SPINS=0 makes every workgroup
    // give up at once and run all its passes itself (tests/test_hip_edge_cases.py: same decisions)
    static const int spins = getenv("M3P2I_LADDER_SPINS") ? atoi(getenv("M3P2I_LADDER_SPINS")) : (1 << 18);
    UpdateArgs a = a_;
    a.ladder_spins = spins;
    const dim3 grid(a.T + a.n_cand);
    const size_t lds = (size_t)a.T * a.nu * sizeof(float);
    const bool multi = a.multi_modal && !a.mode_simple;
    const int rows = (a.Kg + 255) / 256;
#define M3_LAUNCH_SMALL(NU_, MULTI_)                                                                         \
    do {                                                                                                     \
        if (rows <= 8) hipLaunchKernelGGL((k_update_small<NU_, MULTI_, 8>), grid, dim3(256), lds, s, a);     \
        else hipLaunchKernelGGL((k_update_small<NU_, MULTI_, 16>), grid, dim3(256), lds, s, a);              \
    } while (0)
    if (a.nu == 2) {
        // multi-modal with more than 2048 costs: 512-thread workgroups (half the register rows per thread: every
        // per-row loop of the kernel -- loads, ladder points, weights, sums -- halves; C3 24.2 -> 22.4 us, K = 8000
        // 33 -> 27.7 us).  Single mode measured no gain (panda -1 %) or a loss (C2: +10 us on the command although
        // the kernel itself is not slower -- the wider workgroups delay the next rollout's dispatch).
        static const bool wide = getenv("M3P2I_UPDATE_WT256") == nullptr;   // (experiments: the 256-thread instances)
        if (multi && wide && rows > 8) {   // 512 threads per workgroup, ONE top-k workgroup (32 rows of 256 costs)
            a.n_cand = 1;
            const dim3 grid1(a.T + 1);
            if (rows > 16) hipLaunchKernelGGL((k_update_small<2, true, 16, 512>), grid1, dim3(512), lds, s, a);
            else hipLaunchKernelGGL((k_update_small<2, true, 8, 512>), grid1, dim3(512), lds, s, a);
        }
        else if (multi && rows > 16) hipLaunchKernelGGL((k_update_small<2, true, 32>), grid, dim3(256), lds, s, a);
        else if (multi) M3_LAUNCH_SMALL(2, true);
        else if (rows <= 16) M3_LAUNCH_SMALL(2, false);
        else if (rows <= 32) hipLaunchKernelGGL((k_update_small<2, false, 32>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_update_small<2, false, 64>), grid, dim3(256), lds, s, a);
    } else {
        if (multi) M3_LAUNCH_SMALL(9, true); else M3_LAUNCH_SMALL(9, false);
    }
#undef M3_LAUNCH_SMALL
}
bool update_small_applies(const UpdateArgs& a) {
    static const bool off = getenv("M3P2I_SPLIT_UPDATE") != nullptr;   // experiments: the multi-launch path
    if (off) return false;
    // unsharded (finalize fused in), or a shard_mix rank's local softmin (its costs ARE a.Jall)
    if (!(a.fuse_finalize || a.record) || (a.record && (a.multi_modal || a.fuse_finalize))) return false;
    // (with two controls: up to 64 register rows in single mode, K <= 16384, the north-star size; 32 in
    // multi-modal mode, K <= 8192, a C5 shard's size)
    const int kmax = (a.nu != 2) ? 4096 : a.multi_modal ? 8192 : 16384;
    // (mppi_mode 'simple' takes the single-mode path with beta = lambda_: mppi.py:226, skill_utils.py:3)
    return a.Kl == a.Kg && a.Kg <= kmax && a.n_cand == topk_workgroups(a.Kg) && (a.nu == 2 || a.nu == 9);
}

// ---------------------------------------------------------------------------------------
// k_mix (shard_mix): turns the ranks' records into the REDUCE buffer the all-reduce would have
// produced, so that k_finalize runs unchanged.  With beta fixed during the command, the global
// softmin is a mixture of the ranks' local softmins:
//   w_k = exp(-(J_k - m)/beta) / eta,  m = min_r m_r,
//   rho_r = exp(-(m_r - m)/beta) eta_r / sum_r' exp(-(m_r' - m)/beta) eta_r'
//   sum_k w_k a_k = sum_r rho_r S_r   (S_r = the rank's normalised local weighted sum)
// One workgroup; every rank computes the same thing from the same gathered records (fixed
// order over ranks), so the plans stay identical across ranks.
__global__ __launch_bounds__(256) void k_mix(const UpdateArgs a) {
    __shared__ float s_rho[MIX_MAX_RANKS];
    __shared__ float s_soft[3];   // global minimum, -1/beta, 1/Z
    __shared__ int s_best_rank;
    __shared__ tkey s_key[MIX_MAX_RANKS * M3_TOPK];
    __shared__ int s_src[M3_TOPK];
    const int tid = threadIdx.x, T = a.T, nu = a.nu, N = a.n_ranks;
    const int L = record_length(T, nu);
    const float* R = a.records_all;
    if (tid == 0) {
        float m = __builtin_inff();
        int br = 0;
        for (int r = 0; r < N; ++r) {
            const float mr = R[(size_t)r * L + 0];
            if (mr < m) { m = mr; br = r; }  // first rank on ties = lowest sample index
        }
        const float beta = a.mode_simple ? a.lambda_ : a.info->beta;
        const float nib = -1.0f / beta;
        float Z = 0.0f;
        for (int r = 0; r < N; ++r) {
            const float sr = m3_exp(nib * (R[(size_t)r * L + 0] - m)) * R[(size_t)r * L + 1];
            s_rho[r] = sr;
            Z += sr;
        }
        const float iz = 1.0f / Z;
        s_soft[0] = m; s_soft[1] = nib; s_soft[2] = iz;
        float h0 = 0.0f, h1 = 0.0f;
        for (int r = 0; r < N; ++r) {
            s_rho[r] = s_rho[r] * iz;
            h0 += s_rho[r] * R[(size_t)r * L + 2];
            h1 += s_rho[r] * R[(size_t)r * L + 3];
        }
        s_best_rank = br;
        m3_info* f = a.info;
        f->eta = Z; f->eta_1 = 0.0f; f->eta_2 = 0.0f;
        f->iters = 1; f->iters_1 = 1; f->iters_2 = 1;
        f->best_idx = __float_as_int(R[(size_t)br * L + 4]);
        f->best_idx_1 = -1; f->best_idx_2 = -1;
        f->wsum_push = h0; f->wsum_pull = h1;
        f->pull_preference = h1 > h0;
        if (!a.mode_simple) {
            float nb = beta;
            if (a.env_type == M3_ENV_PANDA) {  // mppi.py:446-454, on the GLOBAL eta
                if (Z > 20.0f) nb = nb * 0.9f;
                else if (Z < 10.0f) nb = nb * 1.2f;
            }
            f->beta = nb;
        }
    }
    // candidates of the global top-k: the ranks' sorted lists
    const int nc = N * M3_TOPK;
    for (int c = tid; c < nc; c += blockDim.x) {
        const float* rec = R + (size_t)(c / M3_TOPK) * L;
        s_key[c] = vi_key(rec[REC_TOPJ + c % M3_TOPK], __float_as_int(rec[REC_TOPI + c % M3_TOPK]));
    }
    if (tid < M3_TOPK) s_src[tid] = 0;   // (duplicated keys -- never from real shards -- must not leave a slot unset)
    __syncthreads();
    // weighted sums and the best rows (mode sets 1, 2 are unused in single-mode MPPI)
    const int n = T * nu, br = s_best_rank;
    for (int o = tid; o < n; o += blockDim.x) {
        float acc = 0.0f;
        for (int r = 0; r < N; ++r) acc += s_rho[r] * R[(size_t)r * L + REC_HDR + reduce_off_psum(0, T, nu) + o];
        a.reduce[reduce_off_psum(0, T, nu) + o] = acc;
        a.reduce[reduce_off_best(0, T, nu) + o] = R[(size_t)br * L + REC_HDR + reduce_off_best(0, T, nu) + o];
    }
    // rank counting over the N*20 candidates (keys are unique: the index is part of the key)
    for (int c = tid; c < nc; c += blockDim.x) {
        const tkey my = s_key[c];
        int rank = 0;
#pragma unroll 4
        for (int q = 0; q < nc; ++q) rank += (s_key[q] < my) ? 1 : 0;
        if (rank < M3_TOPK) {
            s_src[rank] = c;
            const VI win = key_vi(my);
            a.top_idx[rank] = win.i;
            // the weights buffer holds this rank's own shard; the top-k samples of OTHER ranks get their
            // global weight too, so that weights[top_idx] (the reference's top_values, mppi.py:248) is
            // complete on every rank
            if (win.i < a.k0 || win.i >= a.k0 + a.Kl) a.w[win.i] = m3_exp(s_soft[1] * (win.v - s_soft[0])) * s_soft[2];
        }
    }
    __syncthreads();
    for (int o = tid; o < M3_TOPK * T * 2; o += blockDim.x) {
        const int slot = o / (T * 2), c = s_src[slot];
        a.reduce[reduce_off_top(T, nu) + o] =
            R[(size_t)(c / M3_TOPK) * L + REC_HDR + reduce_off_top(T, nu) + (c % M3_TOPK) * T * 2 + o % (T * 2)];
    }
    // this rank's weights were normalised by its own eta_r: rescale to the global normalisation
    const float rho = s_rho[a.rank];
    for (int i = tid; i < a.Kl; i += blockDim.x) a.w[a.k0 + i] *= rho;
    // ... and the usual finalize on the buffer just formed, in the same launch (one dependent
    // launch less on the critical path after the collective): stores out to L2, then
    // finalize_body reads them back with L2-coherent loads
    extern __shared__ float sm_mix[];
    __threadfence();
    __syncthreads();
    finalize_body<true>(a, sm_mix);
}
void launch_mix(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_mix, dim3(1), dim3(256), (size_t)a.T * a.nu * sizeof(float), s, a);
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
__global__ __launch_bounds__(256) void k_finalize(const UpdateArgs a) {
    extern __shared__ float sm[];  // [T*nu] new plan
    finalize_body<false>(a, sm);
}
void launch_finalize(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), a.T * a.nu * sizeof(float), s, a);
}

// ---- MPPIConfig.update_cov (mppi.py:201-203, :508-516; off in every shipped config) ----
// delta = actions - mean_action (the NEW mean); cov_update = mean_t sum_k w_k delta^2 per control dimension;
// cov_action <- (1 - 0.7) cov_action + 0.7 cov_update; += 0.005; scale_tril = sqrt(cov_action).
// Two small launches after the finalize (one workgroup per time step, then one thread per dimension): the
// kernel boundary orders the partial sums, nothing on the default path changes.
__global__ __launch_bounds__(256) void k_cov_partial(const float* __restrict__ actions, const float* __restrict__ w,
                                                     const float* __restrict__ mean, float* __restrict__ part,
                                                     int K, int nu) {
    __shared__ float lds[M3_MAX_NU * 16];
    const int t = blockIdx.x, tid = threadIdx.x;
    float acc[M3_MAX_NU];
    float m[M3_MAX_NU];
#pragma unroll
    for (int j = 0; j < M3_MAX_NU; ++j) { acc[j] = 0.0f; m[j] = j < nu ? mean[t * nu + j] : 0.0f; }
    for (int k = tid; k < K; k += 256) {
        const float wk = w[k];
        const float* row = actions + ((size_t)t * K + k) * nu;
#pragma unroll
        for (int j = 0; j < M3_MAX_NU; ++j)
            if (j < nu) { const float d = row[j] - m[j]; acc[j] = acc[j] + wk * (d * d); }
    }
    block_sum<M3_MAX_NU>(acc, lds);
    if (tid < nu) part[t * nu + tid] = acc[tid];
}
__global__ void k_cov_apply(const float* __restrict__ part, float* __restrict__ cov /* [2][nu] */, int T, int nu) {
    const int j = threadIdx.x;
    if (j >= nu) return;
    float s = 0.0f;
    for (int t = 0; t < T; ++t) s = s + part[t * nu + j];
    const float upd = s / (float)T;                                       // torch.mean(..., dim=0)
    float c = (float)(1.0 - 0.7) * cov[j] + 0.7f * upd;                   // mppi.py:514 (step_size_cov = 0.7)
    c = c + 0.005f;                                                       // :515 (kappa)
    cov[j] = c;
    cov[nu + j] = sqrtf(c);                                               // :516
}
void launch_cov_update(const float* actions, const float* w, const float* mean, float* part, float* cov, int K, int T,
                       int nu, hipStream_t s) {
    hipLaunchKernelGGL(k_cov_partial, dim3(T), dim3(256), 0, s, actions, w, mean, part, K, nu);
    hipLaunchKernelGGL(k_cov_apply, dim3(1), dim3(64), 0, s, (const float*)part, cov, T, nu);
}

}  // namespace m3
