// rollout_panda.hip -- fused MPPI rollout kernel for the panda_env + step-mode kernels (gfx950).
//
// Same structure as rollout_point.hip: one launch = MPPI._compute_rollout_costs
// (mppi.py:296-315) for all K samples with the 9-dof action assembly (mppi.py:381-416, gripper
// override :412-416), T x { chain step (replaces reactive_tamp.py:63-70 -> Isaac Gym), task
// cost (cost_functions.py:91-169) }.  `dynamics` still reports dofs 0 and 1 as the 4-vector
// state (reactive_tamp.py:66-70), so states stay [T][K][4]; actions are [T][K][9].
// Algorithmic traffic per state-step: delta 36 B read; state 16 + action 36 + cost 4 B written.
#include "m3_internal.hpp"
#include "noise_stream.hpp"
#include "panda_dyn.hpp"
#include "wave_min.hpp"

#include <cstdlib>

namespace m3 {

// raw world, 57 floats: q9 qd9 | cubeA13 | cubeB13 | dyn-obs13 (each: pos3 quat4(xyzw) linvel3 angvel3)
__device__ __forceinline__ void body_from13(const float* b, Body& o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.p[i] = b[i]; o.v[i] = b[7 + i]; o.w[i] = b[10 + i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) o.q[i] = b[3 + i];
}
__device__ __forceinline__ void panda_world_clear_derived(PandaWorld& w) {
    w.held = 0.0f;
    w.rel_p[0] = w.rel_p[1] = w.rel_p[2] = 0.0f;
    w.rel_q[0] = w.rel_q[1] = w.rel_q[2] = 0.0f; w.rel_q[3] = 1.0f;
    w.awake[0] = w.awake[1] = 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { w.f_table[i] = 0.0f; w.f_shelf[i] = 0.0f; w.f_cubeB[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { w.warm_t[i] = 0.0f; w.warm_l[i] = 0.0f; }
}
__device__ __forceinline__ void panda_world_from_raw(const float* p, PandaWorld& w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { w.q[i] = p[i]; w.qd[i] = p[9 + i]; }
    body_from13(p + 18, w.A);
    body_from13(p + 31, w.B);
#pragma unroll
    for (int i = 0; i < 3; ++i) { w.obs_p[i] = p[44 + i]; w.obs_v[i] = p[51 + i]; }
    panda_world_clear_derived(w);
}

// env 0 of the wrapper's tensors: dof_state row = 9 x (pos, vel) interleaved
// (isaacgym_wrapper.py:98-100), root_state row = pos3 quat4 vel3 ang3 (:102-104)
__device__ __forceinline__ void panda_world_from_sim(const float* dof, const float* root, int ia, int ib, int io,
                                                     PandaWorld& w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { w.q[i] = dof[2 * i]; w.qd[i] = dof[2 * i + 1]; }
    body_from13(root + (size_t)ia * 13, w.A);
    body_from13(root + (size_t)ib * 13, w.B);
    const float* o = root + (size_t)io * 13;
#pragma unroll
    for (int i = 0; i < 3; ++i) { w.obs_p[i] = o[i]; w.obs_v[i] = o[7 + i]; }
    panda_world_clear_derived(w);
}

// the per-lane store of the contact solver in LDS (manifold contact points: 120 floats per lane = 30 KB per wavefront; with one
// lane per sample also the gripper contacts' generalized rows: 312 floats, 78 KB), lane-strided: conflict-free, one wavefront
// per workgroup
#define PANDA_CORNER_LDS(LPS) __shared__ float corner_lds[panda_store_floats(LPS) * 64]; const CornerStore cs{corner_lds + threadIdx.x, 64}

__device__ __forceinline__ float in_vgpr(float v) {   // keep a uniform value in a vector register
    asm volatile("" : "+v"(v));
    return v;
}

// GENERAL = false: the reference's default sampler (halton-spline noise table), the path of every BASELINE
// config.  GENERAL = true adds what no shipped config turns on: the in-kernel random stream with a noise mean and a
// full covariance (sampling_method = 'random', mppi.py:129-131, :481; quirk Q4: that sample is scaled by
// sqrt(diag Sigma) once more) and mppi_mode = 'simple' (mppi.py:220-233, :335-372).
// LPS = lanes per sample (panda_dyn.hpp, world spec v3): 1 -- a lane simulates a sample on its own, 64 samples per wavefront;
// 16 -- the sixteen lanes of a DPP row simulate ONE sample together: everything but the contact solver's generalized vectors
// is replicated in them, the joint-space rows of the gripper contacts run across them.  Four samples per wavefront, so K = 4000
// is 1000 wavefronts, one per SIMD, instead of 63: the launch is as long as its slowest wavefront either way, and a wavefront's
// velocity passes are ~2.3x shorter (tools/ubench/coop_panda_rows.hip).  Chosen by launch_rollout_panda.
template <bool FORCES, bool GENERAL, int LPS>
__global__ __launch_bounds__(64) void k_rollout_panda(const RolloutArgs a_, const PandaArgs pa,
                                                      const PandaScene sc_) {
    PANDA_CORNER_LDS(LPS);
    // (a_.lanes samples per 64-wide wavefront: m3_set_rollout_lanes; the idle lanes leave at once)
    // Shadow lanes (quirk Q8, pa.shadows = 1 or 2; reach on an unsharded handle): the reference's reach cost measures every
    // rollout against the cube of ENVIRONMENT 0 (and, for the tilted mode, the orientation of the first environment of the
    // second half), which under world spec v2 is a quantity of THAT rollout's simulation.  The last sample slot of every
    // wavefront (lane 63; with LPS = 16 the last group of sixteen) re-simulates sample 0 and the one before it sample K / 2 in
    // lockstep with the wavefront's own samples (same noise rows, same operations, so the same bits in every wavefront);
    // their cubes are read with v_readlane after each step.  They store nothing.  Cost: 64 / 63 (62) more wavefronts
    // (LPS = 16: 4 / 3, 4 / 2), no cross-wavefront synchronisation.
    constexpr int SPW = 64 / LPS;                       // sample slots per wavefront
    const bool deferred = LPS != 1 && pa.reach_rec != nullptr;   // (the one-lane form keeps its shadow slots: launch_rollout_panda)
    const int slot = (int)threadIdx.x / LPS, gl = (int)threadIdx.x % LPS;
    const bool shadow = slot >= SPW - pa.shadows;
    const bool writer = !shadow && gl == 0;             // the lane that stores the sample's scalars
    int i = blockIdx.x * a_.lanes + slot;
    if (shadow) i = (slot == SPW - 1) ? 0 : pa.cp.half_K;
    else if (slot >= a_.lanes || i >= a_.Kl) return;
    // The per-joint constants (bounds, noise scale, servo coefficients: 54 floats) are uniform, but
    // there are not enough scalar registers to keep them across the step loop, and the compiler
    // re-read them from the kernel arguments every step (~12 scalar loads per step, each followed by
    // a wait with nothing else resident on the SIMD to cover it).  Vector registers are plentiful at
    // one wave per SIMD, so they are parked there once.
    RolloutArgs a = a_;
    PandaScene sc = sc_;
    if constexpr (!GENERAL) { a.sampling_random = 0; a.mode_simple = 0; a.full_sigma = 0; a.noise_abs_cost = 0; }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        a.u_min[j] = in_vgpr(a_.u_min[j]); a.u_max[j] = in_vgpr(a_.u_max[j]);
        // (update_cov rewrites the scale on the device after every command: mppi.py:516)
        a.scale_tril[j] = in_vgpr(a_.scale_dev ? a_.scale_dev[j] : a_.scale_tril[j]);
        sc.a[j] = in_vgpr(sc_.a[j]); sc.rden[j] = in_vgpr(sc_.rden[j]); sc.dv[j] = in_vgpr(sc_.dv[j]);
    }
    const int Kl = a.Kl, T = a.T;
    const int k = a.k0 + i;
    PandaWorld w;
    if (a.sim_dof) panda_world_from_sim(a.sim_dof, a.sim_root, pa.cubeA_actor, pa.cubeB_actor, pa.obs_actor, w);
    else panda_world_from_raw(pa.world0, w);
    float hp[3], trav = 0.0f;   // hand origin at the last evaluated kinematics, joint travel since (panda_step)
    panda_infer_held(sc, w, hp);

    const bool is_last = (k == a.Kg - 1);
    const bool first_half = k < pa.cp.half_K;
    const bool halton = !a.mode_simple;
    const float* mptr = a.mean;
    if (a.multi_modal && halton) mptr = first_half ? a.mean1 : a.mean2;

    // inputs of step t+1 are fetched before step t is simulated (one wavefront per SIMD: nothing
    // else hides the latency of a load that is consumed at once)
    const bool use_best = halton && a.multi_modal && (k == 0 || k == pa.cp.half_K);
    const float* bptr = (k == 0) ? a.best1 : a.best2;
    float nd[9], nm[9];
    auto fetch = [&](int t) {
        // _shift_action: mppi.py:266-273; simple mode: torch.roll(U, -1), mppi.py:221
        const int ts = a.mode_simple ? ((t + 1 == T) ? 0 : t + 1) : ((t + 1 < T) ? t + 1 : T - 1);
        const float* dptr = a.delta + ((size_t)t * Kl + i) * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            nd[j] = a.sampling_random ? 0.0f : dptr[j];
            nm[j] = use_best ? bptr[ts * 9 + j] : mptr[ts * 9 + j];
        }
    };
    fetch(0);
    FkCarry<LPS> fkc;
    fkc.valid = false;
    fkc.near_lane_substeps = 0;
    float J = 0.0f, g = 1.0f, S = 0.0f, pc = 0.0f;
#ifdef M3_PABL_PROF
    PandaProf prof = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_step = 0;
    const long long prof_start = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < T; ++t) {
        float cd[9], cm[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) { cd[j] = nd[j]; cm[j] = nm[j]; }
        if (t + 1 < T) fetch(t + 1);
        if constexpr (GENERAL) {
            if (a.sampling_random) {   // N(noise_mu, noise_sigma) = mu + L z (noise_stream.hpp; order as the oracle)
                float z[10];
#pragma unroll
                for (int p = 0; p < 5; ++p) gauss_pair(a.seed, a.call, (unsigned)k, (unsigned)t, (unsigned)p, z[2 * p], z[2 * p + 1]);
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float acc;
                    if (a.full_sigma) {
                        acc = a.noise_mats[j * 9 + 0] * z[0];
#pragma unroll
                        for (int q = 1; q <= j; ++q) acc = acc + a.noise_mats[j * 9 + q] * z[q];
                    } else acc = z[j] * a_.scale_tril[j];   // (the configured scale, not update_cov's)
                    cd[j] = a.noise_mu[j] + acc;
                }
            }
        }
        float u[9], e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float aj;
            if (GENERAL && a.mode_simple) {
                aj = fmaxf(fminf(cm[j] + cd[j], a.u_max[j]), a.u_min[j]);             // mppi.py:341-345
            } else {
                const float d = is_last ? 0.0f : cd[j];                                // mppi.py:392
                aj = fmaxf(fminf(cm[j] + d * a.scale_tril[j], a.u_max[j]), a.u_min[j]);
                if (use_best) aj = cm[j];                                              // :407-409
            }
            if (j >= 7) {                                                              // :412-416 / :346-350
                if (a.gripper_cmd == 1) aj = 1.5f;
                else if (a.gripper_cmd == 2) aj = -1.5f;
            }
            float uj = a.u_scale * aj;                                                 // :297
            if (a.sample_null_action && is_last) uj = 0.0f;                            // :300-302
            u[j] = uj;
            e[j] = uj;                                     // :313 (the update consumes the scaled stack)
        }
        PandaObs obs;
#ifdef M3_PABL_PROF
        const long long prof_s0 = __builtin_readcyclecounter();
        panda_step<FORCES, true, LPS>(sc, w, u, obs, cs, hp, &trav, &fkc, &prof);
        prof_step += __builtin_readcyclecounter() - prof_s0;
#else
        panda_step<FORCES, true, LPS>(sc, w, u, obs, cs, hp, &trav, &fkc);
#endif
        float cube0[3], qh0[4];
        if (deferred) {
            // the reach cost of this step is formed by k_panda_reach_cost (below) from what it reads of the sample -- and of
            // samples 0 and K / 2, whose cube it is measured against (quirk Q8): no shadow slots in this launch
            if (writer) {
                float* r = pa.reach_rec + (size_t)t * REACH_REC * Kl + i;
#pragma unroll
                for (int j = 0; j < 3; ++j) { r[(0 + j) * Kl] = obs.left[j]; r[(3 + j) * Kl] = obs.right[j]; r[(14 + j) * Kl] = w.A.p[j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { r[(6 + j) * Kl] = obs.left_q[j]; r[(10 + j) * Kl] = w.A.q[j]; }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) cube0[j] = w.A.p[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) qh0[j] = w.A.q[j];
        } else if (pa.shadows) {
#pragma unroll
            for (int j = 0; j < 3; ++j) cube0[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.A.p[j]), 63));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.A.q[j]), 63));
                const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w.A.q[j]), 63 - LPS));
                qh0[j] = first_half ? q0 : q1;     // (one shadow: single mode, the tilt term does not read it)
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) cube0[j] = w.A.p[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) qh0[j] = w.A.q[j];
        }
        const float c = deferred ? 0.0f : panda_cost(pa.cp, w, obs, k, cube0, qh0);
        if (writer) {
            *reinterpret_cast<float4*>(a.states + ((size_t)t * Kl + i) * 4) =
                make_float4(w.q[0], w.qd[0], w.q[1], w.qd[1]);                   // reactive_tamp.py:66-69
            if (!deferred) a.cost_h[(size_t)t * Kl + i] = c;
        }
        if constexpr (LPS == 1) {
            if (!shadow) {
                float* ap = a.actions + ((size_t)t * Kl + i) * 9;
#pragma unroll
                for (int j = 0; j < 9; ++j) ap[j] = e[j];
            }
        } else {          // the sample's lanes store one control each
            const Gen<LPS> eo = gen_from9<LPS>(e);
#pragma unroll
            for (int el = 0; el < Gen<LPS>::N; ++el) {
                const int cj = gen_coord<LPS>(el);
                if (!shadow && cj < 9) a.actions[((size_t)t * Kl + i) * 9 + cj] = eo.a[el];
            }
        }
        J = J + g * c;
        g = g * a.gamma;
        if constexpr (GENERAL) {
            if (a.mode_simple) {   // mppi.py:309 and the perturbation cost :355-372: sum U * ((lambda * noise) @ Sigma^-1)
                S = S + c;
                float ln[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float n = e[j] - cm[j];
                    if (a.noise_abs_cost) n = fabsf(n);
                    ln[j] = a.lambda_ * n;
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float ac;
                    if (a.full_sigma) {
                        ac = ln[0] * a.noise_mats[81 + 0 * 9 + j];
#pragma unroll
                        for (int q = 1; q < 9; ++q) ac = ac + ln[q] * a.noise_mats[81 + q * 9 + j];
                    } else ac = ln[j] * a.sigma_inv[j];
                    pc = pc + cm[j] * ac;
                }
            }
        }
    }
#ifdef M3_PABL_PROF     // (cost_horizon rows 0-5 of the sample: total / solver / near-path clocks, substeps with gripper rows / body rows / near)
    if (writer && T >= 6) {
        a.cost_h[(size_t)0 * Kl + i] = (float)(__builtin_readcyclecounter() - prof_start);
        a.cost_h[(size_t)1 * Kl + i] = (float)prof.solve_clk;
        a.cost_h[(size_t)2 * Kl + i] = (float)prof.near_clk;
        a.cost_h[(size_t)3 * Kl + i] = (float)prof.n_robot;
        a.cost_h[(size_t)4 * Kl + i] = (float)prof.n_body;
        a.cost_h[(size_t)5 * Kl + i] = (float)prof.n_near;
        if (T >= 10) {
            a.cost_h[(size_t)6 * Kl + i] = (float)prof.detect_clk;
            a.cost_h[(size_t)7 * Kl + i] = (float)prof.post_clk;
            a.cost_h[(size_t)8 * Kl + i] = (float)prof.n_act;
            a.cost_h[(size_t)9 * Kl + i] = (float)prof.n_fk;
        }
        if (T >= 14) {
            a.cost_h[(size_t)10 * Kl + i] = (float)prof.pre_clk;
            a.cost_h[(size_t)11 * Kl + i] = (float)prof.mid_clk;
            a.cost_h[(size_t)12 * Kl + i] = (float)prof.wake_clk;
            a.cost_h[(size_t)13 * Kl + i] = (float)prof_step;
        }
    }
#endif
    if (!deferred) {      // (else: k_panda_reach_cost writes the costs and leaves the minima behind)
        if (writer) a.J[i] = (GENERAL && a.mode_simple) ? (S + pc) : J;
        if (a.wave_min) wave_min_store(a.wave_min, J, first_half, writer);
    }
    // What the NEXT reach commands' kernel form is chosen by (panda_lps_for): the share of (sample, substep) pairs of this launch
    // in which the gripper was within reach of a box or a cube was awake, in 1/1000.  Every wavefront adds its count; the last one to finish (the
    // same atomic is its ticket) turns the sum into the share, stores it into a word of mapped host memory and clears the counters for the next
    // launch.  A hint: results do not depend on the form.
    if (pa.busy_hint != nullptr && threadIdx.x == 0) {
        // ONE atomic carries both: bits 0-23 wavefronts finished, bits 24-63 the sum of their counts
        const unsigned long long mine = ((unsigned long long)(unsigned)(fkc.near_lane_substeps / LPS) << 24) | 1ull;
        const unsigned long long old = atomicAdd(pa.busy_count, mine);
        if ((old & 0xffffffull) == (unsigned long long)(gridDim.x - 1u)) {
            const unsigned long long total = (old + mine) >> 24;
            *pa.busy_count = 0ull;
            const unsigned long long all = (unsigned long long)Kl * (unsigned long long)(T * sc.substeps);
            *(volatile int*)pa.busy_hint = (int)((total * 1000ull) / (all ? all : 1ull)) + 1;    // (+ 1: 0 = nothing reported yet)
        }
    }
}

// The reach cost of a launch that ran WITHOUT shadow slots (PandaArgs::reach_rec): quirk Q8 measures every rollout against the
// cube of environment 0 (and the tilted mode's orientation term against the first environment of the second half), which a
// rollout kernel can only know inside a wavefront by re-simulating those samples in it -- a quarter (half) of the sample slots
// with sixteen lanes per sample, a second round of wavefronts at K = 4000.  Instead the rollout leaves, per (step, sample), the
// seventeen floats the cost reads (finger positions 6, finger orientation 4, the sample's cube orientation 4 and position 3:
// [T][17][Kl], 5.4 MB at C4) and this kernel -- one lane per sample, the same panda_cost on the same values, the same
// discounted sum in the same order -- forms cost_horizon, the trajectory costs and the update's minima rows.
// Round 6: the T steps of a sample are independent until the discounted sum, so a workgroup is 64 samples x RC_TS time slices
// (one wavefront per slice): slice s forms the costs of steps s, s + RC_TS, ... -- seventeen loads per step in flight in four
// wavefronts instead of one lane walking all T x 17 of them --, the costs meet in LDS and the first wavefront adds them in step
// order with the rollout's own recurrence (J = J + g c; g = g gamma): the same operations on the same values, the same bits
// (16 -> 7 us at C4: profiles/r06).
constexpr int RC_TS = 4, RC_CH = 32;      // time slices per workgroup; steps per LDS chunk
__global__ __launch_bounds__(64 * RC_TS) void k_panda_reach_cost(const RolloutArgs a, const PandaArgs pa) {
    __shared__ float s_c[RC_CH][64];
    const int Kl = a.Kl, T = a.T;
    const int lane = (int)threadIdx.x & 63, slice = (int)threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64 + lane;
    const bool mine = i0 < Kl;
    const int i = mine ? i0 : 0;          // (every lane stays for the barriers)
    const int k = a.k0 + i;
    const bool first_half = k < pa.cp.half_K;
    const int h = (pa.cp.multi_modal && !first_half) ? pa.cp.half_K : 0;    // whose cube orientation the tilt term reads
    float J = 0.0f, g = 1.0f;
    for (int t0 = 0; t0 < T; t0 += RC_CH) {
        for (int tt = slice; tt < RC_CH && t0 + tt < T; tt += RC_TS) {
            const int t = t0 + tt;
            const float* r = pa.reach_rec + (size_t)t * REACH_REC * Kl;
            PandaObs o;
            PandaWorld w;
            float cube0[3], qh0[4];
#pragma unroll
            for (int j = 0; j < 3; ++j) { o.left[j] = r[(0 + j) * Kl + i]; o.right[j] = r[(3 + j) * Kl + i]; cube0[j] = r[(14 + j) * Kl]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { o.left_q[j] = r[(6 + j) * Kl + i]; w.A.q[j] = r[(10 + j) * Kl + i]; qh0[j] = r[(10 + j) * Kl + h]; }
            const float c = panda_cost(pa.cp, w, o, k, cube0, qh0);
            if (mine) a.cost_h[(size_t)t * Kl + i] = c;
            s_c[tt][lane] = c;
        }
        __syncthreads();
        if (slice == 0) {
            for (int tt = 0; tt < RC_CH && t0 + tt < T; ++tt) {
                J = J + g * s_c[tt][lane];
                g = g * a.gamma;
            }
        }
        __syncthreads();
    }
    if (slice == 0 && mine) a.J[i] = J;
    if (a.wave_min) wave_min_store(a.wave_min, J, first_half, mine && slice == 0);
}

// Lanes per sample, by what was measured at C4's size (profiles/r05/panda_lps_bench.json, panda_reach_mid_bench.json; K = 4000,
// T = 20):
//   pick / place (no shadow slots, gripper + manifold rows in most wavefronts): 16 lanes 0.74 ms, 8 lanes 0.82, 1 lane 1.57 -- a
//     wavefront is the union of what its samples do and its rows run across the lanes: sixteen while the launch has <= 1024
//     wavefronts (one per SIMD: the launch lasts as long as its slowest wavefront), eight up to there, else one;
//   reach (quirk Q8): with the arm away from everything nothing touches anything, the time is the replicated part of the step,
//     which more lanes per sample only repeat in more wavefronts -- one lane with its shadow slots, 0.167 ms with the cubes asleep;
//     once the rollouts are next to the cube (most of an episode's reach phase) the many-lane forms win by up to 2x: sixteen lanes
//     WITHOUT shadow slots + k_panda_reach_cost where launch_rollout_panda has the record buffer, else eight lanes with them (sixteen
//     would lose 1-2 of 4 sample slots and need two rounds of wavefronts).  pa.reach_busy says which (m3_api.hip).
// Beyond 1024 wavefronts the launch is throughput-bound and the replicated work (16x / 8x more instructions per sample
// outside the solver) decides: one lane.  pa.lps (m3_set_panda_lanes_per_sample) forces a form.
static int panda_lps_for(const RolloutArgs& a, const PandaArgs& pa) {
    if (pa.lps == 1 || pa.lps == 8 || pa.lps == 16) return pa.lps;
    auto waves = [&](int lps) { const int per = 64 / lps - pa.shadows; return (a.Kl + per - 1) / per; };
    if (pa.shadows != 0) {
        // reach: one lane while next to nothing is near anything; eight lanes (one round of wavefronts with the shadow slots) once
        // the last command's rollouts had the gripper within reach of the cubes / the table often enough (pa.reach_busy: the share
        // the kernel reports, with hysteresis; m3_api.hip) -- the scenes 20 / 40 / 60 ticks
        // into an episode's reach phase (tools/panda_reach_mid_bench.py): 1 lane 0.174 / 0.778 / 1.535 ms, 8 lanes 0.198 /
        // 0.445 / 0.828, 16 lanes 0.271 / 0.561 / 1.265
        const bool busy = pa.reach_busy != 0;
        return (busy && waves(8) <= 1024) ? 8 : 1;
    }
    if (waves(16) <= 1024) return 16;
    if (waves(8) <= 1024) return 8;
    return 1;
}
template <int LPS>
static int launch_rollout_panda_lps(const RolloutArgs& a_in, const PandaArgs& pa, const PandaScene& sc, hipStream_t s) {
    constexpr int SPW = 64 / LPS;
    int lanes = (a_in.lanes >= 1 && a_in.lanes <= SPW) ? a_in.lanes : SPW;
    if (lanes > SPW - pa.shadows) lanes = SPW - pa.shadows;
    RolloutArgs a = a_in;
    a.lanes = lanes;
    const dim3 grid((a.Kl + lanes - 1) / lanes), block(64);
    if (a.sampling_random || a.mode_simple) {
        if (pa.cp.task == 5) hipLaunchKernelGGL((k_rollout_panda<true, true, LPS>), grid, block, 0, s, a, pa, sc);
        else hipLaunchKernelGGL((k_rollout_panda<false, true, LPS>), grid, block, 0, s, a, pa, sc);
    } else if (pa.cp.task == 5) hipLaunchKernelGGL((k_rollout_panda<true, false, LPS>), grid, block, 0, s, a, pa, sc);
    else hipLaunchKernelGGL((k_rollout_panda<false, false, LPS>), grid, block, 0, s, a, pa, sc);
    return (int)grid.x;
}
// returns the number of workgroups (= rows of the wave_min table)
int launch_rollout_panda(const RolloutArgs& a, const PandaArgs& pa_in, const PandaScene& sc, hipStream_t s, int* lps_used) {
    PandaArgs pa = pa_in;
    // reach without shadow slots (k_panda_reach_cost): when the handle holds the record buffer (m3_api.hip: K up to 8192), the
    // sampler is the default one and a many-lane form is what the launch
    // sixteen (eight) lanes per sample run in; automatic choice: while few of the last command's (sample, substep) pairs had the
    // gripper within reach of a box the one-lane form with its shadow slots is the faster launch (pa.reach_busy, m3_api.hip:
    // K = 4000, rollout ms one lane / sixteen lanes + cost kernel: arm at its initial pose 0.169 / 0.190, 20 ticks into an episode
    // 0.182 / 0.187, 30 ticks 0.388 / 0.248, 40 ticks 0.798 / 0.419, 60 ticks 1.541 / 0.778; profiles/r05/panda_reach_mid_bench.json)
    int lps = 0;
    if (pa.reach_rec != nullptr) {
        pa.shadows = 0;
        // (round 6: with the cost kernel in time slices the sixteen-lane form + cost kernel is at least as fast as one lane + shadow
        // slots in EVERY scene of an episode -- 10 ticks in 0.160 / 0.161 ms, 20 ticks 0.168 / 0.190, cubes settled 0.174 / 0.177,
        // initial scene 0.324 / 0.362, 60 ticks 0.744 / 1.545: profiles/r06/panda_reach_mid_bench.json -- so wherever one round of
        // sixteen-lane wavefronts fits (K <= 4096) the form no longer depends on the reported share; the eight-lane form, which
        // loses to one lane in quiet scenes, keeps following it)
        const bool sixteen_fits = (a.Kl + 3) / 4 <= 1024;
        const bool want = pa_in.shadows != 0 && !(a.sampling_random || a.mode_simple) && (pa.lps != 0 || pa.reach_busy != 0 || sixteen_fits);
        lps = want ? panda_lps_for(a, pa) : 1;
        if (lps == 1) pa = pa_in, pa.reach_rec = nullptr;
    }
    if (pa.reach_rec == nullptr) lps = panda_lps_for(a, pa);
    if (lps_used) *lps_used = lps;
    if (pa.reach_rec != nullptr) {
        if (lps == 16) (void)launch_rollout_panda_lps<16>(a, pa, sc, s);
        else (void)launch_rollout_panda_lps<8>(a, pa, sc, s);
        const dim3 grid((a.Kl + 63) / 64), block(64 * RC_TS);
        hipLaunchKernelGGL(k_panda_reach_cost, grid, block, 0, s, a, pa);
        return (int)grid.x;
    }
    return lps == 16 ? launch_rollout_panda_lps<16>(a, pa, sc, s) : lps == 8 ? launch_rollout_panda_lps<8>(a, pa, sc, s)
                                                                              : launch_rollout_panda_lps<1>(a, pa, sc, s);
}

// ======================= step mode ======================================================
// SoA world rows (NWP = 77): q 0-8 | qd 9-17 | cubeA 18-30 (pos3 quat4 vel3 angvel3) | cubeB 31-43 | plate pos 44-46,
// vel 47-49 | held 50 | rel_p 51-53 | rel_q 54-57 | awake 58-59 | f_table 60-62 | f_shelf 63-65 | f_cubeB 66-68 |
// warm_t 69-72 | warm_l 73-76
__device__ __forceinline__ void psoa_load(const float* wd, int Kl, int i, PandaWorld& w) {
    const float* p = wd + i;
    auto body = [&](int r0, Body& b) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { b.p[j] = p[(r0 + j) * Kl]; b.v[j] = p[(r0 + 7 + j) * Kl]; b.w[j] = p[(r0 + 10 + j) * Kl]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) b.q[j] = p[(r0 + 3 + j) * Kl];
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) { w.q[j] = p[j * Kl]; w.qd[j] = p[(9 + j) * Kl]; }
    body(18, w.A);
    body(31, w.B);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        w.obs_p[j] = p[(44 + j) * Kl]; w.obs_v[j] = p[(47 + j) * Kl]; w.rel_p[j] = p[(51 + j) * Kl];
        w.f_table[j] = p[(60 + j) * Kl]; w.f_shelf[j] = p[(63 + j) * Kl]; w.f_cubeB[j] = p[(66 + j) * Kl];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { w.rel_q[j] = p[(54 + j) * Kl]; w.warm_t[j] = p[(69 + j) * Kl]; w.warm_l[j] = p[(73 + j) * Kl]; }
    w.held = p[50 * Kl];
    w.awake[0] = p[58 * Kl]; w.awake[1] = p[59 * Kl];
}
__device__ __forceinline__ void psoa_store(float* wd, int Kl, int i, const PandaWorld& w) {
    float* p = wd + i;
    auto body = [&](int r0, const Body& b) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { p[(r0 + j) * Kl] = b.p[j]; p[(r0 + 7 + j) * Kl] = b.v[j]; p[(r0 + 10 + j) * Kl] = b.w[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) p[(r0 + 3 + j) * Kl] = b.q[j];
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) { p[j * Kl] = w.q[j]; p[(9 + j) * Kl] = w.qd[j]; }
    body(18, w.A);
    body(31, w.B);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        p[(44 + j) * Kl] = w.obs_p[j]; p[(47 + j) * Kl] = w.obs_v[j]; p[(51 + j) * Kl] = w.rel_p[j];
        p[(60 + j) * Kl] = w.f_table[j]; p[(63 + j) * Kl] = w.f_shelf[j]; p[(66 + j) * Kl] = w.f_cubeB[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[(54 + j) * Kl] = w.rel_q[j]; p[(69 + j) * Kl] = w.warm_t[j]; p[(73 + j) * Kl] = w.warm_l[j]; }
    p[50 * Kl] = w.held;
    p[58 * Kl] = w.awake[0]; p[59 * Kl] = w.awake[1];
}

// SoA world of environment i -> the wrapper's views (link poses through the forward kinematics)
__device__ __forceinline__ void panda_push_views(const PandaScene& sc, const SimViews& v, int i, const PandaWorld& w) {
    if (v.dof_state) {
        float* d = v.dof_state + (size_t)i * 18;
#pragma unroll
        for (int j = 0; j < 9; ++j) { d[2 * j] = w.q[j]; d[2 * j + 1] = w.qd[j]; }
    }
    auto body13 = [&](const Body& b, float* o) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { o[j] = b.p[j]; o[7 + j] = b.v[j]; o[10 + j] = b.w[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[3 + j] = b.q[j];
    };
    if (v.root_state) {
        float* base = v.root_state + (size_t)i * v.n_actors * 13;
        body13(w.A, base + v.box_actor * 13);
        body13(w.B, base + v.dyn_actor * 13);
        float* o = base + v.obs_actor * 13;      // the plate: position and linear velocity (it does not rotate)
#pragma unroll
        for (int j = 0; j < 3; ++j) { o[j] = w.obs_p[j]; o[7 + j] = w.obs_v[j]; }
    }
    if (v.rigid_body_state) {
        float* base = v.rigid_body_state + (size_t)i * v.n_bodies * 13;
        body13(w.A, base + v.box_body * 13);
        body13(w.B, base + v.dyn_body * 13);
        float* o = base + v.obs_body * 13;
#pragma unroll
        for (int j = 0; j < 3; ++j) { o[j] = w.obs_p[j]; o[7 + j] = w.obs_v[j]; }
        // robot links: bodies robot_body .. robot_body + 10 (link0..7, hand, left, right)
        float links[11 * 7];
        Frame hand;
        float pl[3], pr[3];
        panda_fk<true>(sc, w.q, hand, pl, pr, links);
        for (int l = 0; l < 11; ++l) {
            float* ol = base + (v.robot_body + l) * 13;
            for (int j = 0; j < 7; ++j) ol[j] = links[l * 7 + j];
            for (int j = 7; j < 13; ++j) ol[j] = 0.0f;  // link velocities are not reported
        }
    }
    if (v.net_contact_force) {
        float* f = v.net_contact_force + (size_t)i * v.n_bodies * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            f[v.table_body * 3 + j] = w.f_table[j];
            f[v.shelf_body * 3 + j] = w.f_shelf[j];
            f[v.dyn_body * 3 + j] = w.f_cubeB[j];
        }
    }
}

// one sim.step() of every environment and the refresh of the wrapper's views in the same launch
__global__ __launch_bounds__(64) void k_psim_step(const PandaScene sc, const SimViews v, float* wd, const float* u,
                                                  float* u_keep, int Kl) {
    PANDA_CORNER_LDS(1);
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PandaWorld w;
    psoa_load(wd, Kl, i, w);
    float uu[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) uu[j] = u[(size_t)i * 9 + j];
    if (u_keep != u) {   // (targets taken from the caller's tensor: kept for the steps after this one)
#pragma unroll
        for (int j = 0; j < 9; ++j) u_keep[(size_t)i * 9 + j] = uu[j];
    }
    PandaObs obs;
    panda_step(sc, w, uu, obs, cs);
    psoa_store(wd, Kl, i, w);
    panda_push_views(sc, v, i, w);
}
void launch_psim_step(const PandaScene& sc, const SimViews& v, float* world, const float* u, float* u_keep, int Kl,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_psim_step, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, v, world, u, u_keep, Kl);
}

__global__ __launch_bounds__(64) void k_psim_pull(const PandaScene sc, const SimViews v, float* wd, int Kl) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PandaWorld w;
    panda_world_from_sim(v.dof_state + (size_t)i * 18, v.root_state + (size_t)i * v.n_actors * 13,
                         v.box_actor, v.dyn_actor, v.obs_actor, w);  // box_actor = cubeA, dyn_actor = cubeB here
    panda_infer_held(sc, w);
    psoa_store(wd, Kl, i, w);
}
void launch_psim_pull(const PandaScene& sc, const SimViews& v, float* world, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_psim_pull, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, v, world, Kl);
}

__global__ __launch_bounds__(64) void k_psim_push(const PandaScene sc, const SimViews v, const float* wd, int Kl) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PandaWorld w;
    psoa_load(wd, Kl, i, w);
    panda_push_views(sc, v, i, w);
}
void launch_psim_push(const PandaScene& sc, const SimViews& v, const float* world, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_psim_push, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, v, world, Kl);
}

__global__ __launch_bounds__(64) void k_psim_cost(const PandaScene sc, const PandaCostParams cp, const float* wd,
                                                  int Kl, int k0, int env0_cube, float* cost) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PandaWorld w;
    psoa_load(wd, Kl, i, w);
    Frame hand;
    PandaObs o;
    panda_fk<false>(sc, w.q, hand, o.left, o.right, nullptr);
    mat2quat(hand, o.left_q);
    // quirk Q8: environment 0's cube position, the orientation of the first environment of the sample's half (rows 18-24)
    float cube0[3], qh0[4];
    const bool env0 = env0_cube != 0;      // (the host's condition, the one m3_rollout uses for its shadow lanes)
    const int src = env0 ? ((cp.multi_modal && i >= cp.half_K) ? cp.half_K : 0) : i;
#pragma unroll
    for (int j = 0; j < 3; ++j) cube0[j] = wd[(18 + j) * Kl + (env0 ? 0 : i)];
#pragma unroll
    for (int j = 0; j < 4; ++j) qh0[j] = wd[(21 + j) * Kl + src];
    cost[i] = panda_cost(cp, w, o, k0 + i, cube0, qh0);
}
void launch_psim_cost(const PandaScene& sc, const PandaCostParams& cp, const float* world, int Kl, int k0, bool env0_cube,
                      float* cost, hipStream_t s) {
    hipLaunchKernelGGL(k_psim_cost, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, cp, world, Kl, k0, env0_cube ? 1 : 0, cost);
}

}  // namespace m3
