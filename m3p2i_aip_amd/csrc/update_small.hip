// update_small.hip -- k_update_small: the WHOLE importance-weight update in ONE launch for the sizes of the reference's
// configs (update.hip's header comment; shared device code: update_common.hpp).
#include "update_common.hpp"

namespace m3 {

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
    // (a.ladder_spins: bounded wait of the in-launch ladder exchange, ~20 ms; m3_set_ladder_spins(h, 0) makes every
    // workgroup give up at once and run all its passes itself -- tests/test_hip_edge_cases.py: same decisions)
    UpdateArgs a = a_;
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
#ifdef M3_EXP_UPDATE_WT256   // (experiment build: the 256-thread instances)
        constexpr bool wide = false;
#else
        constexpr bool wide = true;
#endif
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
#ifdef M3_EXP_SPLIT_UPDATE   // (experiment build: the multi-launch path at every size)
    return false;
#endif
    // unsharded (finalize fused in), or a shard_mix rank's local softmin (its costs ARE a.Jall)
    if (!(a.fuse_finalize || a.record) || (a.record && (a.multi_modal || a.fuse_finalize))) return false;
    // (with two controls: up to 64 register rows in single mode, K <= 16384, the north-star size; 32 in
    // multi-modal mode, K <= 8192, a C5 shard's size)
    const int kmax = (a.nu != 2) ? 4096 : a.multi_modal ? 8192 : 16384;
    // (mppi_mode 'simple' takes the single-mode path with beta = lambda_: mppi.py:226, skill_utils.py:3)
    return a.Kl == a.Kg && a.Kg <= kmax && a.n_cand == topk_workgroups(a.Kg) && (a.nu == 2 || a.nu == 9);
}

int init_ladder_table_small() { return init_ladder_table_tu(); }

}  // namespace m3
