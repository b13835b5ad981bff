// rollout_point.hip -- point_env rollout: launch dispatch, the all-modes instance of the kernel
// (rollout_point_kernel.hpp), the noise transpose and the step-mode (IsaacGymWrapper-like) kernels.
#include "rollout_point_kernel.hpp"

namespace m3 {

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

// returns whether the instance launched leaves the workgroups' cost minima in a.wave_min (wave_min.hpp)
bool launch_rollout_point(const RolloutArgs& a, const PointScene& sc, hipStream_t s) {
    const int blocks = (a.Kl + a.lanes - 1) / a.lanes;
#if defined(M3_ABL_GENERAL_ONLY) || defined(M3_ABL_COUNT) || defined(M3_ABL_PHASES)   // (experiments: one kernel for all modes;
    // the instrumented builds keep their counters in this translation unit)
    const bool general = true;
#else
    // push_pull without multi_modal is refused upstream (m3_rollout); a task outside 0..3 cannot reach here
    const bool general = a.sampling_random || a.mode_simple || a.cp.task < 0 || a.cp.task > 3 ||
                         (a.cp.task == 3 && !a.multi_modal) || a.scale_dev != nullptr /* update_cov */ ||
                         a.cp.avoid_dyn_obs != 0 /* the extension: the dyn-obs contact force must be formed */;
#endif
    if (general) { launch_rollout_point_instance<true, -1>(a, sc, blocks, s); return a.wave_min != nullptr; }
    switch (a.cp.task) {   // the reference's default sampler: one instance per task (rollout_point_task*.hip)
        case 0: launch_rollout_point_nav(a, sc, blocks, s); break;
        case 1: launch_rollout_point_push(a, sc, blocks, s); break;
        case 2: launch_rollout_point_pull(a, sc, blocks, s); break;
        default: launch_rollout_point_pushpull(a, sc, blocks, s); return a.wave_min != nullptr;
    }
    return false;
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

// SoA world of environment i -> the wrapper's views (what a refresh_*_tensor call of Isaac Gym does)
__device__ __forceinline__ void push_views(const SimViews& v, int i, const PointWorld& w) {
    if (v.dof_state) {
        *reinterpret_cast<float4*>(v.dof_state + (size_t)i * 4) = make_float4(w.rx, w.rvx, w.ry, w.rvy);
    }
    if (v.root_state) {
        float* base = v.root_state + (size_t)i * v.n_actors * 13;
        write_body13(base + v.box_actor * 13, w.B.x, w.B.y, w.B.c, w.B.s, w.B.vx, w.B.vy, w.B.w);
        write_body13(base + v.dyn_actor * 13, w.D.x, w.D.y, w.D.c, w.D.s, w.D.vx, w.D.vy, w.D.w);
        // (fixed base of the robot: stays at its init pose)
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

// one sim.step() of every environment and the refresh of the wrapper's views in the same launch
// (u_keep != u: the targets come from the caller's tensor and are kept for the steps after this one, as a
// set_dof_velocity_target_tensor in front of the step would have done)
__global__ __launch_bounds__(64) void k_sim_step(const PointScene sc, const SimViews v, float* wd, const float* u,
                                                 float* u_keep, int Kl) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= Kl) return;
    PointWorld w;
    soa_load(wd, Kl, i, w);
    const float2 uu = *reinterpret_cast<const float2*>(u + (size_t)i * 2);
    if (u_keep != u) *reinterpret_cast<float2*>(u_keep + (size_t)i * 2) = uu;
    point_step<true>(sc, w, uu.x, uu.y);
    soa_store(wd, Kl, i, w);
    push_views(v, i, w);
}
void launch_sim_step(const PointScene& sc, const SimViews& v, float* world, const float* u, float* u_keep, int Kl,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_sim_step, dim3((Kl + 63) / 64), dim3(64), 0, s, sc, v, world, u, u_keep, Kl);
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
__device__ __forceinline__ void pull_env(const SimViews& v, float* wd, int Kl, int i) {
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
__global__ void k_sim_pull(const SimViews v, float* wd, int Kl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    pull_env(v, wd, Kl, i);
}
// update_dyn_obs (isaacgym_wrapper.py:205-220): one actor's root position shifted in the wrapper's root_state
// view + set_actor_root_state_tensor, in one launch
__global__ void k_sim_shift_pull(const SimViews v, float* wd, int Kl, int actor, float dx, float dy, float dz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    float* r = v.root_state + ((size_t)i * v.n_actors + actor) * 13;
    r[0] = r[0] + dx; r[1] = r[1] + dy; r[2] = r[2] + dz;
    pull_env(v, wd, Kl, i);
}
void launch_sim_shift_pull(const SimViews& v, float* world, int Kl, int actor, float dx, float dy, float dz, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_shift_pull, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, Kl, actor, dx, dy, dz);
}
void launch_sim_pull(const SimViews& v, float* world, int Kl, hipStream_t s) {
    hipLaunchKernelGGL(k_sim_pull, dim3((Kl + 255) / 256), dim3(256), 0, s, v, world, Kl);
}

__global__ void k_sim_push(const SimViews v, const float* wd, int Kl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kl) return;
    PointWorld w;
    soa_load(wd, Kl, i, w);
    push_views(v, i, w);
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
