// update_sharded.hip -- the update kernels of the sharding protocols: the pre-gather record
// (k_local_topk), shard_mix = 2 (k_regen_part / k_regen_done -- which, with loaded instead of re-generated actions, also
// are the second and third launch of the unsharded three-launch update), shard_mix = 3 (k_p3_done), shard_mix = 1 single
// mode (k_mix).  Shared device code: update_common.hpp.
#include "update_common.hpp"

namespace m3 {

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
            f->best_idx = c0.i == 0x7fffffff ? -1 : c0.i;      // (no argmax: -1, as every other path reports it)
            f->best_idx_1 = c1.i == 0x7fffffff ? -1 : c1.i;
            f->best_idx_2 = c2.i == 0x7fffffff ? -1 : c2.i;
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

int init_ladder_table_sharded() { return init_ladder_table_tu(); }

}  // namespace m3
