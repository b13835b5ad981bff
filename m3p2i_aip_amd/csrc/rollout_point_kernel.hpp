// rollout_point_kernel.hpp -- fused MPPI rollout kernel for the point_env (gfx950): the kernel template.
// Instantiated in rollout_point.hip (all sampler modes, task at run time) and, for the reference's default
// sampler, once per task in rollout_point_task*.hip (separate translation units: they compile in parallel).
//
// One launch = the whole hot loop of MPPI._compute_rollout_costs (mppi.py:296-315) for all K
// samples: action assembly (mppi.py:381-416 / :335-347), T x { velocity-servo + contact
// dynamics step (replaces reactive_tamp.py:63-70 -> Isaac Gym), task cost
// (cost_functions.py:19-89,158-169), discounted accumulation (mppi_utils.py:106-113) }.
// The reference issues ~180 (push) to ~540 (push_pull) aten launches + one PhysX step PER
// TIME STEP for this; here the T-loop runs inside the kernel with the world in registers.
//
// Mapping: one lane per sample, 64-lane workgroups (one wavefront) so that K = 2000 spreads
// over 32 CUs.  HBM layout is time-major ([T][K][c]) so lane i writes address base + i*c*4:
// every store of the wave is one contiguous 256 B .. 1 KiB segment -- no LDS staging needed.
// Algorithmic traffic per state-step (nu = 2): read delta 8 B, write state 16 B + action
// 8 B + cost 4 B = 36 B (28 B with in-kernel noise); the update pass re-reads actions (8 B).
#pragma once
#include "m3_internal.hpp"
#include "noise_stream.hpp"
#include "wave_min.hpp"

namespace m3 {

__device__ __forceinline__ void load_world(const float* p, PointWorld& w) {
    w.rx = p[0]; w.ry = p[1]; w.rvx = p[2]; w.rvy = p[3];
    w.B.x = p[4]; w.B.y = p[5]; w.B.c = p[6]; w.B.s = p[7]; w.B.vx = p[8]; w.B.vy = p[9]; w.B.w = p[10];
    w.D.x = p[11]; w.D.y = p[12]; w.D.c = p[13]; w.D.s = p[14]; w.D.vx = p[15]; w.D.vy = p[16]; w.D.w = p[17];
    w.fcDx = w.fcDy = w.fcBx = w.fcBy = w.fcRx = w.fcRy = 0.0f;
}

// env 0 of the wrapper's tensors -> world (yaw from the (0,0,qz,qw) quaternion);
// dof_state row = [x, vx, y, vy] (isaacgym_wrapper.py:120-126), root row = pos3 quat4 vel3 ang3
__device__ __forceinline__ void load_world_from_sim(const float* dof, const float* root, int box,
                                                    int dyn, PointWorld& w) {
    w.rx = dof[0]; w.ry = dof[2]; w.rvx = dof[1]; w.rvy = dof[3];
    const float* r = root + (size_t)box * 13;
    float qz = r[5], qw = r[6];
    w.B.x = r[0]; w.B.y = r[1]; w.B.c = 1.0f - 2.0f * (qz * qz); w.B.s = 2.0f * (qz * qw);
    w.B.vx = r[7]; w.B.vy = r[8]; w.B.w = r[12];
    r = root + (size_t)dyn * 13;
    qz = r[5]; qw = r[6];
    w.D.x = r[0]; w.D.y = r[1]; w.D.c = 1.0f - 2.0f * (qz * qz); w.D.s = 2.0f * (qz * qw);
    w.D.vx = r[7]; w.D.vy = r[8]; w.D.w = r[12];
    w.fcDx = w.fcDy = w.fcBx = w.fcBy = w.fcRx = w.fcRy = 0.0f;
}

// Launch geometry: 64-thread workgroups (one wavefront) of which `lanes` are active
// (default 64 = one lane per sample; lanes = 1 is the north_star's literal "one wavefront
// per sample" and was measured 2-9x slower, see rollout_lanes_for below).
// GENERAL = false: the reference's default sampler (halton-spline noise table, mppi_mode 'halton-spline'), the
// path of every BASELINE config: the in-kernel random stream and the simple-mode bookkeeping are compiled
// out (fewer live uniform values: the kernel spills scalar registers into vector lanes).
// TASK >= 0: the task of the cost function is a compile-time constant too (cost_functions.py:19-36: 0
// navigation, 1 push, 2 pull, 3 push_pull -- which implies multi_modal): the other tasks' cost code, the
// suction bookkeeping of the tasks that have none and the dyn-obs contact force of the tasks that do not read
// it disappear from the instance.  Measured at C2: one kernel for everything 0.1555 ms per command, sampler
// mode compiled in 0.1530, task compiled in as well 0.143 (same results bit for bit: only which code exists).
template <bool GENERAL, int TASK>
__device__ __forceinline__ void rollout_point_body(const RolloutArgs& a_, const PointScene& sc) {
    RolloutArgs a = a_;
    if constexpr (!GENERAL) {
        a.sampling_random = 0; a.mode_simple = 0;
        a.noise_abs_cost = 0; a.full_sigma = 0; a.scale_dev = nullptr;   // (the host routes those to the general instance)
        a.cp.avoid_dyn_obs = 0;
    }
    if constexpr (TASK >= 0) {
        a.cp.task = TASK;
        if constexpr (TASK == 3) { a.multi_modal = 1; a.cp.multi_modal = 1; }
    }
    const int slot = blockIdx.x * a.lanes + threadIdx.x;
    if ((int)threadIdx.x >= a.lanes || slot >= a.Kl) return;
    const int i = a.order ? a.order[slot] : slot;
    const int Kl = a.Kl, T = a.T;
    const int k = a.k0 + i;  // global sample index
    PointWorld w;
    if (a.sim_dof) load_world_from_sim(a.sim_dof, a.sim_root, a.sim_box, a.sim_dyn, w);
    else load_world(a.world0, w);
    w.fRx = a.pend[0 * Kl + i]; w.fRy = a.pend[1 * Kl + i];
    w.fBx = a.pend[2 * Kl + i]; w.fBy = a.pend[3 * Kl + i];

    const bool is_last = (k == a.Kg - 1);
    const bool first_half = k < a.cp.half_K;
    const float* mptr = a.mean;
    if (a.multi_modal && !a.mode_simple) mptr = first_half ? a.mean1 : a.mean2;

    // Inputs of step t+1 are fetched before step t is simulated: with one wavefront per SIMD
    // nothing else hides the ~1-2 us HBM/L2 latency of a load whose result is needed at once
    // (measured: SQ_WAIT_ANY was a third of the kernel's wave-cycles).
    struct StepIn { float d0, d1, m0, m1, b0, b1; };
    const bool halton = !a.mode_simple;
    const bool use_best = halton && a.multi_modal && (k == 0 || k == a.cp.half_K);
    const float* bptr = (k == 0) ? a.best1 : a.best2;
    auto fetch = [&](int t) {
        StepIn in;
        in.d0 = in.d1 = in.b0 = in.b1 = 0.0f;
        if (!a.sampling_random) {
            const float2 dd = *reinterpret_cast<const float2*>(a.delta + ((size_t)t * Kl + slot) * 2);
            in.d0 = dd.x; in.d1 = dd.y;
        }
        // torch.roll(U, -1): mppi.py:221 / _shift_action: mppi.py:266-273
        const int ts = a.mode_simple ? ((t + 1 == T) ? 0 : t + 1) : ((t + 1 < T) ? t + 1 : T - 1);
        in.m0 = mptr[ts * 2 + 0]; in.m1 = mptr[ts * 2 + 1];
        if (use_best) { in.b0 = bptr[ts * 2 + 0]; in.b1 = bptr[ts * 2 + 1]; }
        return in;
    };

    // the MPPIConfig switches no shipped config turns on (general instance only): scale_tril rewritten by
    // update_cov (mppi.py:516), Cholesky factor / inverse of a non-diagonal noise_sigma (mppi.py:128-131)
    // (the sampling distribution keeps the configured covariance: only scale_tril is rewritten, mppi.py:129-131 vs :516)
    float L10 = 0.0f, L00 = a.scale_tril[0], L11 = a.scale_tril[1];
    float S00 = a.sigma_inv[0], S01 = 0.0f, S10 = 0.0f, S11 = a.sigma_inv[1];
    if (a.scale_dev) { a.scale_tril[0] = a.scale_dev[0]; a.scale_tril[1] = a.scale_dev[1]; }
    if (a.full_sigma) {
        L00 = a.noise_mats[0]; L10 = a.noise_mats[2]; L11 = a.noise_mats[3];
        S00 = a.noise_mats[4]; S01 = a.noise_mats[5]; S10 = a.noise_mats[6]; S11 = a.noise_mats[7];
    }

    float J = 0.0f, S = 0.0f, g = 1.0f, pc = 0.0f;
    StepIn nxt = fetch(0);
#ifdef M3_ABL_PHASES
    PhaseClock clk, *pc_ = &clk;
    clk.start();
#else
    PhaseClock* pc_ = nullptr;
#endif
    for (int t = 0; t < T; ++t) {
        const StepIn in = nxt;
        if (t + 1 < T) nxt = fetch(t + 1);
        // ---- A4 / A13: perturbed action for this (k, t) ----
        float d0 = in.d0, d1 = in.d1;
        if (a.sampling_random) {   // N(noise_mu, noise_sigma) = mu + L z: mppi.py:129-131, :340 / :481
            float z0, z1;
            gauss_pair(a.seed, a.call, (unsigned)k, (unsigned)t, 0u, z0, z1);
            d0 = a.noise_mu[0] + L00 * z0;
            float acc = L11 * z1;
            if (a.full_sigma) acc = L10 * z0 + acc;
            d1 = a.noise_mu[1] + acc;
        }
        float a0, a1;
        const float m0 = in.m0, m1 = in.m1;
        if (a.mode_simple) {
            a0 = fmaxf(fminf(m0 + d0, a.u_max[0]), a.u_min[0]);  // mppi.py:343-345
            a1 = fmaxf(fminf(m1 + d1, a.u_max[1]), a.u_min[1]);
        } else {
            if (is_last) { d0 = 0.0f; d1 = 0.0f; }      // mppi.py:392
            a0 = fmaxf(fminf(m0 + d0 * a.scale_tril[0], a.u_max[0]), a.u_min[0]);  // :394-405
            a1 = fmaxf(fminf(m1 + d1 * a.scale_tril[1], a.u_max[1]), a.u_min[1]);
            if (use_best) { a0 = in.b0; a1 = in.b1; }  // mppi.py:407-409
        }
        float u0 = a.u_scale * a0, u1 = a.u_scale * a1;                 // mppi.py:297
        if (a.sample_null_action && is_last) { u0 = 0.0f; u1 = 0.0f; }  // mppi.py:300-302

        M3_PH(0);
        // ---- A6: one sim.step() ----
        point_step<false>(sc, w, u0, u1, /*need_dyn_force=*/a.cp.task == 0 || a.cp.avoid_dyn_obs != 0, pc_);

        // ---- A7/A8: running cost on the post-step state ----
        const float c = point_cost(a.cp, w, k);
        M3_PH(5);

        // ---- outputs, time-major ----
        *reinterpret_cast<float4*>(a.states + ((size_t)t * Kl + i) * 4) =
            make_float4(w.rx, w.rvx, w.ry, w.rvy);                      // reactive_tamp.py:66-69
        // mppi.py:313: the stack the distribution update consumes holds the SCALED controls (:329-331, :355);
        // the division of :353 / :420 only touches the attribute the caller reads (planner.actions)
        const float e0 = u0, e1 = u1;
        *reinterpret_cast<float2*>(a.actions + ((size_t)t * Kl + i) * 2) = make_float2(e0, e1);
        a.cost_h[(size_t)t * Kl + i] = c;                               // mppi.py:310
        J = J + g * c;                                                  // mppi_utils.py:106-113
        S = S + c;                                                      // mppi.py:309
        g = g * a.gamma;
        if (a.mode_simple) {  // perturbation cost, mppi.py:355-372: sum U * ((lambda * noise) @ Sigma^-1)
            float n0 = e0 - m0, n1 = e1 - m1;
            if (a.noise_abs_cost) { n0 = fabsf(n0); n1 = fabsf(n1); }      // :366-367
            const float l0 = a.lambda_ * n0, l1 = a.lambda_ * n1;
            float c0 = l0 * S00, c1 = l1 * S11;
            if (a.full_sigma) { c0 = c0 + l1 * S10; c1 = l0 * S01 + c1; }
            pc = pc + m0 * c0;
            pc = pc + m1 * c1;
        }
        M3_PH(6);
    }
#ifdef M3_ABL_PHASES
    if (threadIdx.x == 0 && blockIdx.x < 1024)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_phase[blockIdx.x * 8 + q], clk.acc[q]);
#endif
    a.J[i] = a.mode_simple ? (S + pc) : J;
    // (only the instances a multi-modal command can run carry this epilogue: compiled into the push instance too it cost
    // the headline 2 us -- 316 instead of 312 VGPRs -- without ever running there; launch_rollout_point says who wrote)
    if constexpr (GENERAL || TASK == 3) {
        if (a.wave_min) wave_min_store(a.wave_min, J, first_half, true);
    }
    a.pend[0 * Kl + i] = w.fRx; a.pend[1 * Kl + i] = w.fRy;
    a.pend[2 * Kl + i] = w.fBx; a.pend[3 * Kl + i] = w.fBy;
}

// Three builds of every instance.  k_rollout_point: no register limit -- 312 VGPRs (256 + 56 AGPR), ONE resident wave
// per SIMD: the fastest build while the launch has no more wavefronts than the chip has SIMDs (K_local <= 65536:
// every BASELINE config).  k_rollout_point_occ2: `amdgpu_waves_per_eu(2, 2)` -- 256 VGPRs, ~60 values spilled to
// scratch, TWO resident waves per SIMD whose instruction streams interleave: 2-3 % slower below 65536 samples, but
// K = 131072: 0.22 -> 0.162 ms.  k_rollout_point_occ3 (below) from four wavefronts per SIMD on.  The host picks by the
// number of wavefronts (rollout_two_waves / rollout_three_waves).  Same arithmetic, same bits.
template <bool GENERAL, int TASK>
__global__ __launch_bounds__(64) void k_rollout_point(const RolloutArgs a, const PointScene sc) {
    rollout_point_body<GENERAL, TASK>(a, sc);
}
// ... and the same build with the reference's solver settings compiled in (planar_dyn.hpp: POINT_SCENE_REFERENCE)
template <bool GENERAL, int TASK>
__global__ __launch_bounds__(64) void k_rollout_point_ref(const RolloutArgs a) {
    constexpr PointScene sc = POINT_SCENE_REFERENCE;
    rollout_point_body<GENERAL, TASK>(a, sc);
}
template <bool GENERAL, int TASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rollout_point_occ2(const RolloutArgs a,
                                                                                                      const PointScene sc) {
    rollout_point_body<GENERAL, TASK>(a, sc);
}
// ... and THREE resident waves (170 VGPRs) from four wavefronts per SIMD on: with the fused multiply-adds of spec v1.4 the
// two-wave build spills only ~60 values, the three-wave build about what the two-wave build used to -- K = 524 288
// 0.557 -> 0.497 ms, 1 M 1.08 -> 0.95 ms; equal at 262 144, slower below (a third wave that is not there does not help).
template <bool GENERAL, int TASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_rollout_point_occ3(const RolloutArgs a,
                                                                                                      const PointScene sc) {
    rollout_point_body<GENERAL, TASK>(a, sc);
}
template <bool GENERAL, int TASK>
inline void launch_rollout_point_instance(const RolloutArgs& a, const PointScene& sc, int blocks, hipStream_t s) {
    if (rollout_three_waves(blocks)) hipLaunchKernelGGL((k_rollout_point_occ3<GENERAL, TASK>), dim3(blocks), dim3(64), 0, s, a, sc);
    else if (rollout_two_waves(blocks)) hipLaunchKernelGGL((k_rollout_point_occ2<GENERAL, TASK>), dim3(blocks), dim3(64), 0, s, a, sc);
    else if (point_scene_is_reference(sc)) hipLaunchKernelGGL((k_rollout_point_ref<GENERAL, TASK>), dim3(blocks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_rollout_point<GENERAL, TASK>), dim3(blocks), dim3(64), 0, s, a, sc);
}


}  // namespace m3
