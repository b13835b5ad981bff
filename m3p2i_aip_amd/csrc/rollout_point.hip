// rollout_point.hip -- fused MPPI rollout kernel for the point_env (gfx950).
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
#include "m3_internal.hpp"

namespace m3 {

// ---- counter-based noise stream (spec: DESIGN.md "Noise stream"; mirrors the oracle) ----
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& x) {
    unsigned long long z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned rotl32(unsigned x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ unsigned xoshiro128pp(unsigned (&s)[4]) {
    const unsigned result = rotl32(s[0] + s[3], 7) + s[0];
    const unsigned t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}
// standard-normal pair for (seed, call, k, t, pair)
__device__ __forceinline__ void gauss_pair(unsigned long long seed, unsigned call, unsigned k,
                                           unsigned t, unsigned pair, float& z0, float& z1) {
    unsigned long long x = seed ^ (0xD1B54A32D192ED03ULL * (unsigned long long)(call + 1u));
    x ^= ((unsigned long long)k << 32) | ((unsigned long long)t << 8) | (unsigned long long)pair;
    const unsigned long long a = splitmix64(x), b = splitmix64(x);
    unsigned s[4] = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    const unsigned r0 = xoshiro128pp(s), r1 = xoshiro128pp(s);
    const float u0 = ((float)(r0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = (float)(r1 >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u0));
    const float ang = 6.28318530717958647692f * u1;
    z0 = rad * cosf(ang);
    z1 = rad * sinf(ang);
}

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
__global__ __launch_bounds__(64) void k_rollout_point(const RolloutArgs a, const PointScene sc) {
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
        if (a.sampling_random) {
            gauss_pair(a.seed, a.call, (unsigned)k, (unsigned)t, 0u, d0, d1);
            d0 *= a.scale_tril[0]; d1 *= a.scale_tril[1];  // N(0, Sigma): mppi.py:481 / :340
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
        point_step<false>(sc, w, u0, u1, /*need_dyn_force=*/a.cp.task == 0, pc_);

        // ---- A7/A8: running cost on the post-step state ----
        const float c = point_cost(a.cp, w, k);
        M3_PH(5);

        // ---- outputs, time-major ----
        *reinterpret_cast<float4*>(a.states + ((size_t)t * Kl + i) * 4) =
            make_float4(w.rx, w.rvx, w.ry, w.rvy);                      // reactive_tamp.py:66-69
        // mppi.py:421 (x / 1 == x exactly: the usual u_scale = 1 skips two IEEE divisions per step)
        float e0 = u0, e1 = u1;
        if (a.u_scale != 1.0f) { e0 = u0 / a.u_scale; e1 = u1 / a.u_scale; }   // wave-uniform branch
        *reinterpret_cast<float2*>(a.actions + ((size_t)t * Kl + i) * 2) = make_float2(e0, e1);
        a.cost_h[(size_t)t * Kl + i] = c;                               // mppi.py:310
        J = J + g * c;                                                  // mppi_utils.py:106-113
        S = S + c;                                                      // mppi.py:309
        g = g * a.gamma;
        if (a.mode_simple) {  // perturbation cost, mppi.py:355-362 (diagonal Sigma)
            pc = pc + m0 * (a.lambda_ * (e0 - m0) * a.sigma_inv[0]);
            pc = pc + m1 * (a.lambda_ * (e1 - m1) * a.sigma_inv[1]);
        }
        M3_PH(6);
    }
#ifdef M3_ABL_PHASES
    if (threadIdx.x == 0 && blockIdx.x < 1024)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_phase[blockIdx.x * 8 + q], clk.acc[q]);
#endif
    a.J[i] = a.mode_simple ? (S + pc) : J;
    a.pend[0 * Kl + i] = w.fRx; a.pend[1 * Kl + i] = w.fRy;
    a.pend[2 * Kl + i] = w.fBx; a.pend[3 * Kl + i] = w.fBy;
}

// lanes per wavefront.  MEASURED on MI355X (tools/lanes_sweep.py, DESIGN.md "Lanes per
// wavefront"): full 64-lane waves are never slower -- K=2000: 0.38 ms at 64 lanes vs 0.83 ms
// at 1 lane (2000 waves), K=10000: 0.36 ms vs 3.3 ms.  Narrow waves do execute ~2x fewer
// instructions each (no divergence), but the kernel ends with its SLOWEST wave, whose
// contact-heavy sample runs the same long path either way, and co-resident waves on a CU
// slow each other down.  So the automatic choice is 64; the knob stays for experiments.
int rollout_lanes_for(int Kl) {
    (void)Kl;
    return 64;
}

void launch_rollout_point(const RolloutArgs& a, const PointScene& sc, hipStream_t s) {
    const int blocks = (a.Kl + a.lanes - 1) / a.lanes;
    hipLaunchKernelGGL(k_rollout_point, dim3(blocks), dim3(64), 0, s, a, sc);
}

// delta [K][T][nu] (reference layout) -> [T][K][nu]
__global__ void k_transpose_noise(const float* __restrict__ src, float* __restrict__ dst, int K,
                                  int T, int nu) {
    const size_t n = (size_t)K * T * nu;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < n;
         o += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(o % nu);
        const size_t r = o / nu;
        const int k = (int)(r % K);
        const int t = (int)(r / K);
        dst[o] = src[((size_t)k * T + t) * nu + j];
    }
}
void launch_transpose_noise(const float* src, float* dst, int K, int T, int nu, hipStream_t s) {
    const size_t n = (size_t)K * T * nu;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_transpose_noise, dim3(blocks), dim3(256), 0, s, src, dst, K, T, nu);
}

// ======================= step mode (IsaacGymWrapper-like surface) =======================
__device__ __forceinline__ void soa_load(const float* wd, int Kl, int i, PointWorld& w) {
    const float* p = wd + i;
    w.rx = p[0 * Kl]; w.ry = p[1 * Kl]; w.rvx = p[2 * Kl]; w.rvy = p[3 * Kl];
    w.B.x = p[4 * Kl]; w.B.y = p[5 * Kl]; w.B.c = p[6 * Kl]; w.B.s = p[7 * Kl];
    w.B.vx = p[8 * Kl]; w.B.vy = p[9 * Kl]; w.B.w = p[10 * Kl];
    w.D.x = p[11 * Kl]; w.D.y = p[12 * Kl]; w.D.c = p[13 * Kl]; w.D.s = p[14 * Kl];
    w.D.vx = p[15 * Kl]; w.D.vy = p[16 * Kl]; w.D.w = p[17 * Kl];
    w.fRx = p[18 * Kl]; w.fRy = p[19 * Kl]; w.fBx = p[20 * Kl]; w.fBy = p[21 * Kl];
    w.fcDx = p[22 * Kl]; w.fcDy = p[23 * Kl]; w.fcBx = p[24 * Kl]; w.fcBy = p[25 * Kl];
    w.fcRx = p[26 * Kl]; w.fcRy = p[27 * Kl];
}
__device__ __forceinline__ void soa_store(float* wd, int Kl, int i, const PointWorld& w) {
    float* p = wd + i;
    p[0 * Kl] = w.rx; p[1 * Kl] = w.ry; p[2 * Kl] = w.rvx; p[3 * Kl] = w.rvy;
    p[4 * Kl] = w.B.x; p[5 * Kl] = w.B.y; p[6 * Kl] = w.B.c; p[7 * Kl] = w.B.s;
    p[8 * Kl] = w.B.vx; p[9 * Kl] = w.B.vy; p[10 * Kl] = w.B.w;
    p[11 * Kl] = w.D.x; p[12 * Kl] = w.D.y; p[13 * Kl] = w.D.c; p[14 * Kl] = w.D.s;
    p[15 * Kl] = w.D.vx; p[16 * Kl] = w.D.vy; p[17 * Kl] = w.D.w;
    p[18 * Kl] = w.fRx; p[19 * Kl] = w.fRy; p[20 * Kl] = w.fBx; p[21 * Kl] = w.fBy;
    p[22 * Kl] = w.fcDx; p[23 * Kl] = w.fcDy; p[24 * Kl] = w.fcBx; p[25 * Kl] = w.fcBy;
    p[26 * Kl] = w.fcRx; p[27 * Kl] = w.fcRy;
}

__global__ __launch_bounds__(64) void k_sim_step(const PointScene sc, float* wd, const float* u,
                                                 int Kl) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PointWorld w;
    soa_load(wd, Kl, i, w);
    const float2 uu = *reinterpret_cast<const float2*>(u + (size_t)i * 2);
    point_step<true>(sc, w, uu.x, uu.y);
    soa_store(wd, Kl, i, w);
}
void launch_sim_step(const PointScene& sc, float* world, const float* u, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_step, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, world, u, Kl);
}

__global__ void k_sim_cost(const CostParams cp, float* wd, int Kl, int k0, float* cost) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    PointWorld w;
    soa_load(wd, Kl, i, w);
    cost[i] = point_cost(cp, w, k0 + i);
    float* p = wd + i;  // only the pending force changes
    p[18 * Kl] = w.fRx; p[19 * Kl] = w.fRy; p[20 * Kl] = w.fBx; p[21 * Kl] = w.fBy;
}
void launch_sim_cost(const CostParams& cp, float* world, int Kl, int k0, float* cost,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_sim_cost, dim3((Kl + 255) / 256), dim3(256), 0, s, cp, world, Kl, k0, cost);
}

// wrapper views (AoS, torch-owned) -> SoA world.  dof_state row = [x, vx, y, vy]
// (isaacgym_wrapper.py:120-126); root_state row = pos3 quat4(xyzw) linvel3 angvel3 (:102-104)
__global__ void k_sim_pull(const SimViews v, float* wd, int Kl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    float* p = wd + i;
    const float* d = v.dof_state + (size_t)i * 4;
    p[0 * Kl] = d[0]; p[1 * Kl] = d[2]; p[2 * Kl] = d[1]; p[3 * Kl] = d[3];
    for (int b = 0; b < 2; ++b) {
        const float* r = v.root_state + ((size_t)i * v.n_actors + (b == 0 ? v.box_actor : v.dyn_actor)) * 13;
        const int o = 4 + b * 7;
        const float qz = r[5], qw = r[6];
        p[(o + 0) * Kl] = r[0]; p[(o + 1) * Kl] = r[1];
        p[(o + 2) * Kl] = 1.0f - 2.0f * (qz * qz);
        p[(o + 3) * Kl] = 2.0f * (qz * qw);
        p[(o + 4) * Kl] = r[7]; p[(o + 5) * Kl] = r[8]; p[(o + 6) * Kl] = r[12];
    }
}
void launch_sim_pull(const SimViews& v, float* world, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_pull, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, Kl);
}

__device__ __forceinline__ void write_body13(float* r, float x, float y, float c, float s,
                                             float vx, float vy, float wz) {
    // yaw (c, s) -> quaternion (0, 0, sin(th/2), cos(th/2)) with cos(th/2) >= 0
    float qw = sqrtf(fmaxf(0.5f * (1.0f + c), 0.0f));
    float qz;
    if (qw > 1e-4f) qz = s / (2.0f * qw);
    else { qz = 1.0f; qw = 0.0f; }
    r[0] = x; r[1] = y;  // r[2] (z) is left as set at init
    r[3] = 0.0f; r[4] = 0.0f; r[5] = qz; r[6] = qw;
    r[7] = vx; r[8] = vy; r[9] = 0.0f;
    r[10] = 0.0f; r[11] = 0.0f; r[12] = wz;
}

__global__ void k_sim_push(const SimViews v, const float* wd, int Kl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    PointWorld w;
    soa_load(wd, Kl, i, w);
    if (v.dof_state) {
        *reinterpret_cast<float4*>(v.dof_state + (size_t)i * 4) = make_float4(w.rx, w.rvx, w.ry, w.rvy);
    }
    if (v.root_state) {
        float* base = v.root_state + (size_t)i * v.n_actors * 13;
        write_body13(base + v.box_actor * 13, w.B.x, w.B.y, w.B.c, w.B.s, w.B.vx, w.B.vy, w.B.w);
        write_body13(base + v.dyn_actor * 13, w.D.x, w.D.y, w.D.c, w.D.s, w.D.vx, w.D.vy, w.D.w);
        float* rr = base + v.robot_actor * 13;  // fixed base of the robot: stays at init pose
        (void)rr;
    }
    if (v.rigid_body_state) {
        float* base = v.rigid_body_state + (size_t)i * v.n_bodies * 13;
        write_body13(base + v.box_body * 13, w.B.x, w.B.y, w.B.c, w.B.s, w.B.vx, w.B.vy, w.B.w);
        write_body13(base + v.dyn_body * 13, w.D.x, w.D.y, w.D.c, w.D.s, w.D.vx, w.D.vy, w.D.w);
        // robot links: plane (fixed), link_x (x only), link_y (x, y)
        write_body13(base + (v.robot_body - 1) * 13, w.rx, 0.0f, 1.0f, 0.0f, w.rvx, 0.0f, 0.0f);
        write_body13(base + v.robot_body * 13, w.rx, w.ry, 1.0f, 0.0f, w.rvx, w.rvy, 0.0f);
    }
    if (v.net_contact_force) {
        float* f = v.net_contact_force + (size_t)i * v.n_bodies * 3;
        f[v.box_body * 3 + 0] = w.fcBx; f[v.box_body * 3 + 1] = w.fcBy;
        f[v.dyn_body * 3 + 0] = w.fcDx; f[v.dyn_body * 3 + 1] = w.fcDy;
        f[v.robot_body * 3 + 0] = w.fcRx; f[v.robot_body * 3 + 1] = w.fcRy;
    }
}
void launch_sim_push(const SimViews& v, const float* world, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_push, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, Kl);
}

// apply_rigid_body_force_tensors: only the box and the robot's last link take forces in
// the reference's use (skill_utils.py:86-90); xy components, consumed by the next step.
__global__ void k_sim_forces(const SimViews v, float* wd, const float* f, int Kl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    const float* fi = f + (size_t)i * v.n_bodies * 3;
    float* p = wd + i;
    p[18 * Kl] = fi[v.robot_body * 3 + 0]; p[19 * Kl] = fi[v.robot_body * 3 + 1];
    p[20 * Kl] = fi[v.box_body * 3 + 0];   p[21 * Kl] = fi[v.box_body * 3 + 1];
}
void launch_sim_forces(const SimViews& v, float* world, const float* f, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_forces, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, f, Kl);
}

// The 1-env "real world" side of scripts/sim.py:41-49: suction between the robot and the box
// (behaves like utils/skill_utils.py:36-94), evaluated on the wrapper's environments without a
// host round trip.
//   forces != null : calculate_suction -- the [Kl][nB][3] body-force tensor (zero except the box
//                    row and the LAST body's row, +-kp * unit(box - robot) clamped to +-500, only
//                    where 1/|box - robot| exceeds `thresh`)
//   action != null : check_suction_condition -- robot within `reach` of the box and the commanded
//                    velocity pointing away from it -- and, if `apply`, the force of above staged
//                    as the pending external force of the next step (apply_rigid_body_force_tensors)
//   gate != null   : a device flag (the planner's pull preference, m3_info.pull_preference) that must be
//                    non-zero for the suction to act: cfg.suction_active without a host round trip
__global__ void k_sim_suction(const SimViews v, float* wd, int Kl, float kp, float thresh, float reach,
                              const float* action, int apply, float* forces, int* flags, const int* gate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    const bool enabled = gate ? (*gate != 0) : true;
    float* p = wd + i;
    const float ex = p[4 * Kl] - p[0 * Kl], ey = p[5 * Kl] - p[1 * Kl];   // robot -> box
    const float len = sqrtf(ex * ex + ey * ey);
    const float inv = 1.0f / len;
    float fx = 0.0f, fy = 0.0f;                                          // force on the robot
    if (inv > thresh) {
        fx = clamp500(kp * (ex * inv));
        fy = clamp500(kp * (ey * inv));
    }
    if (forces) {
        float* f = forces + (size_t)i * v.n_bodies * 3;
        for (int q = 0; q < v.n_bodies * 3; ++q) f[q] = 0.0f;
        f[v.box_body * 3 + 0] = -fx; f[v.box_body * 3 + 1] = -fy;
        f[(v.n_bodies - 1) * 3 + 0] = fx; f[(v.n_bodies - 1) * 3 + 1] = fy;
    }
    if (action) {
        const float along = action[2 * i] * (-ex) + action[2 * i + 1] * (-ey);   // action . (robot - box)
        const bool pulling = enabled && len < reach && along > 0.0f;
        if (flags) flags[i] = pulling ? 1 : 0;
        if (pulling && apply) {
            p[18 * Kl] = fx; p[19 * Kl] = fy; p[20 * Kl] = -fx; p[21 * Kl] = -fy;
        }
    }
}
void launch_sim_suction(const SimViews& v, float* world, int Kl, float kp, float thresh, float reach,
                        const float* action, int apply, float* forces, int* flags, const int* gate, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_suction, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, Kl, kp, thresh, reach,
                       action, apply, forces, flags, gate);
}

}  // namespace m3

#ifdef M3_ABL_PHASES
extern "C" void m3_dbg_phases(unsigned long long* out, int reset) {
    if (reset) { static unsigned long long z[1024 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(m3::g_phase), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(m3::g_phase), 1024 * 8 * sizeof(unsigned long long));
}
#endif
#ifdef M3_ABL_COUNT
extern "C" void m3_dbg_levels(unsigned int* out, int reset) {
    if (reset) { static unsigned int z[512]; (void)hipMemcpyToSymbol(HIP_SYMBOL(m3::g_lvl), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(m3::g_lvl), 512 * sizeof(unsigned int));
}
extern "C" void m3_dbg_cycles(unsigned int* out, int reset) {
    if (reset) { static unsigned int z[64 * 16]; (void)hipMemcpyToSymbol(HIP_SYMBOL(m3::g_cyc), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(m3::g_cyc), 64 * 16 * sizeof(unsigned int));
}
#endif
