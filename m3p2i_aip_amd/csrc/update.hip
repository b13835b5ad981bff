// update.hip -- importance-weight update of MPPI / M3P2I (gfx950): the general multi-launch path.  (update_small.hip:
// k_update_small; update_sharded.hip: k_local_topk, k_regen_part / k_regen_done, k_p3_done, k_mix; update_common.hpp: the device
// code they share.)
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
#include "update_common.hpp"

namespace m3 {

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

int init_ladder_table_small();
int init_ladder_table_sharded();
int init_ladder_table() {     // every translation unit's copy of the beta ladder (update_common.hpp)
    return init_ladder_table_tu() | init_ladder_table_small() | init_ladder_table_sharded();
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

int apply_workgroups(int Kg) { return (Kg + AP_T * AP_RPT - 1) / (AP_T * AP_RPT); }
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
    const UpdateArgs& a = a_;   // (a.ladder_spins: m3_set_ladder_spins)
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
int wsum_chunks(int Kl) { const int L = wsum_chunk_len(Kl); return (Kl + L - 1) / L; }

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

void launch_p3_search(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_search, dim3(2), dim3(WT_MAX), 0, s, a);     // workgroup 1: the global top-k from the shards' lists
}
void launch_p3_local_weights(const UpdateArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_weights<true>, dim3(apply_workgroups(a.Kg)), dim3(AP_T), 0, s, a);
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
