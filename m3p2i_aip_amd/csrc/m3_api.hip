// m3_api.hip -- C-ABI of libm3p2i_hip.so (see include/m3p2i_hip.h for what each entry
// point replaces in the reference).  Host code only: argument checking, buffer ownership,
// launch sequencing on the handle's HIP stream.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <new>

#include "m3_internal.hpp"

using namespace m3;

static thread_local std::string g_create_err;

#define HIPCHK(h, expr)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return M3_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

static int fail(m3_handle* h, int code, const char* msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

extern "C" int m3_abi_version(void) { return M3_ABI_VERSION; }

extern "C" const char* m3_last_error(const m3_handle* h) {
    return h ? h->err.c_str() : g_create_err.c_str();
}

extern "C" void m3_default_config(m3_config* c, int env_type) {
    std::memset(c, 0, sizeof(*c));
    c->abi_version = M3_ABI_VERSION;
    c->env_type = env_type;
    c->u_scale = 1.0f;
    c->gamma = 0.95f;         // MPPIConfig.rollout_var_discount, mppi.py:50
    c->step_size_mean = 0.98f;  // mppi.py:178
    c->sample_null_action = 1;
    c->filter_u = 1;
    c->substeps = 2;          // isaacgym_wrapper.py:10
    c->solver_iters = 6;      // isaacgym_wrapper.py:28
    if (env_type == M3_ENV_POINT) {  // config/mppi/point.yaml, config_point.yaml, isaacgym/point.yaml
        c->nu = 2; c->K_global = c->K_local = 200; c->T = 15; c->u_per_command = 15;
        c->lambda_ = 0.5f; c->kp_suction = 400.0f; c->dt = 0.05f;
        for (int j = 0; j < 2; ++j) { c->u_min[j] = -3.0f; c->u_max[j] = 3.0f; c->noise_sigma_diag[j] = 3.0f; }
    } else {  // config/mppi/panda.yaml, config_panda.yaml, isaacgym/panda.yaml
        c->nu = 9; c->K_global = c->K_local = 200; c->T = 12; c->u_per_command = 12;
        c->lambda_ = 0.05f; c->pre_height_diff = 0.05f; c->dt = 0.01f;
        for (int j = 0; j < 9; ++j) {
            c->u_min[j] = j < 7 ? -2.0f : -1.5f; c->u_max[j] = j < 7 ? 2.0f : 1.5f;
            c->noise_sigma_diag[j] = j < 7 ? 10.0f : 0.8f;
        }
    }
}

// "Planar contact dynamics spec v1" constants (sources: config/point_env/*.yaml,
// pointRobot.urdf, isaacgym_wrapper.py:18-37,341-344,462-469)
static void build_point_scene(const m3_config& c, PointScene& s) { make_point_scene(s, c.dt, c.substeps, c.solver_iters); }
// everything else of the scene is compile-time constant in PointScene (planar_dyn.hpp)

// "Panda chain spec" constants (sources: franka_panda.urdf limits, config/panda_env/*.yaml,
// isaacgym_wrapper.py:341-344)
static void build_panda_scene(const m3_config& c, PandaScene& s) { make_panda_scene(s, c.dt, c.substeps); }

static void default_panda_world(float* w, int cube_on_shelf) {
    const float q0[9] = {0, 0, 0, -2.0f, 0, 1.8675f, 0, 0.02f, 0.02f};  // panda.yaml:10
    std::memset(w, 0, 57 * sizeof(float));      // q9 qd9 | cubeA13 | cubeB13 | dyn-obs13
    for (int i = 0; i < 9; ++i) w[i] = q0[i];
    if (cube_on_shelf) { w[18] = 0.425f; w[19] = 0.0f; w[20] = 1.35f; }     // 5_cubeA.yaml
    else { w[18] = 0.2f; w[19] = -0.2f; w[20] = 1.06f; }
    w[24] = 1.0f;
    w[31] = 0.2f; w[32] = 0.2f; w[33] = 1.06f; w[37] = 1.0f;                // 6_cubeB.yaml
    w[44] = 0.35f; w[45] = 0.0f; w[46] = 1.735f; w[50] = 1.0f;              // 4_obs.yaml
}

static void default_world(float* w) {
    const float init[18] = {0, 0, 0, 0, /*box 7_box.yaml*/ 0, 2, 1, 0, 0, 0, 0,
                            /*dyn-obs 6_dyn_obs.yaml*/ -2, 2, 1, 0, 0, 0, 0};
    std::memcpy(w, init, sizeof(init));
}

static int alloc_buf(m3_handle* h, int id, long long bytes) {
    if (bytes < 16) bytes = 16;
    HIPCHK(h, hipMalloc(&h->buf[id], (size_t)bytes));
    HIPCHK(h, hipMemsetAsync(h->buf[id], 0, (size_t)bytes, h->stream));
    h->nbytes[id] = bytes;
    return M3_OK;
}

static constexpr int PANDA_BUSY_ON = 300, PANDA_BUSY_OFF = 220;
static constexpr int PANDA_REACH_REC_MAX_K = 8192, PANDA_BUSY_ON_REC = 260, PANDA_BUSY_OFF_REC = 190;   // (the forms tie between 116 and ~250 per mille; the arm's initial pose reads 148)

extern "C" int m3_create(const m3_config* c, m3_handle** out) {
    if (!c || !out) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: null argument");
    if (c->abi_version != M3_ABI_VERSION) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: abi_version mismatch");
    if (c->K_global < 1 || c->K_local < 1 || c->k_offset < 0 || c->k_offset + c->K_local > c->K_global)
        return fail(nullptr, M3_ERR_SHAPE, "m3_create: bad K_global/K_local/k_offset");
    if (c->T < 1 || c->T > 4096) return fail(nullptr, M3_ERR_SHAPE, "m3_create: bad horizon T");
    if (c->env_type == M3_ENV_POINT && c->nu != 2) return fail(nullptr, M3_ERR_SHAPE, "m3_create: point_env needs nu == 2");
    if (c->env_type == M3_ENV_PANDA && c->nu != 9) return fail(nullptr, M3_ERR_SHAPE, "m3_create: panda_env needs nu == 9");
    if (c->env_type != M3_ENV_POINT && c->env_type != M3_ENV_PANDA) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: bad env_type");
    if (!c->sim_only) {
        if (c->K_global < M3_TOPK) return fail(nullptr, M3_ERR_SHAPE, "m3_create: K must be >= 20 (torch.topk(weights, 20), mppi.py:248)");
        if (c->filter_u && (c->mode_simple ? c->u_per_command : c->T) < 9)
            return fail(nullptr, M3_ERR_SHAPE, "m3_create: filter_u needs >= 9 rows (savgol window, mppi.py:190)");
        if (c->mode_simple && (c->u_per_command < 1 || c->u_per_command > c->T))
            return fail(nullptr, M3_ERR_SHAPE, "m3_create: bad u_per_command");
    }
    if (c->substeps < 1 || c->solver_iters < 1 || !(c->dt > 0.0f)) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: bad dt/substeps/solver_iters");
    if (c->shard_mix && c->K_local != c->K_global) {
        if (c->multi_modal && !c->mode_simple && c->sampling_random)
            return fail(nullptr, M3_ERR_UNSUPPORTED, "m3_create: one-collective multi-modal sharding re-generates the other ranks' "
                                                     "actions from the noise TABLE: not with sampling_random");
        if (c->shard_mix >= 2 && !(c->multi_modal && !c->mode_simple))
            return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: shard_mix = 2 / 3 (ladder tables in the records) are multi-modal protocols");
        if (c->shard_mix < 0 || c->shard_mix > 3) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: shard_mix must be 0, 1, 2 or 3");
        if (c->shard_mix == 3 && (long long)c->T * c->nu > 2048)
            return fail(nullptr, M3_ERR_UNSUPPORTED, "m3_create: shard_mix = 3 needs T * nu <= 2048");
        if (c->shard_mix == 3 && apply_workgroups(c->K_local) > 256)     // (m3_update_b's partial sums: one slot per workgroup)
            return fail(nullptr, M3_ERR_UNSUPPORTED, "m3_create: shard_mix = 3: K_local too large for the second exchange's partial sums");
        if (c->K_global % c->K_local != 0 || c->k_offset % c->K_local != 0 || c->K_global / c->K_local > MIX_MAX_RANKS)
            return fail(nullptr, M3_ERR_SHAPE, "m3_create: shard_mix needs equal shards (K_global = n * K_local, n <= 32)");
        if (c->K_local < M3_TOPK) return fail(nullptr, M3_ERR_SHAPE, "m3_create: shard_mix needs K_local >= 20");
        if (c->multi_modal && !c->mode_simple) {
            // the record of a one-collective multi-modal shard holds its top trajectories as float2 rows at
            // rec + K_local + 40 (regen_off_trajs): 8-byte aligned only for even K_local; shard_of() forms k / K_local
            // through a binary32 product that is exact only below 2^24
            if (c->K_local % 2) return fail(nullptr, M3_ERR_SHAPE, "m3_create: one-collective multi-modal sharding needs an even K_local");
            if (c->K_global >= (1 << 24)) return fail(nullptr, M3_ERR_SHAPE, "m3_create: one-collective multi-modal sharding needs K_global < 2^24");
        }
    }
    for (int j = 0; j < c->nu; ++j)
        if (!(c->noise_sigma_diag[j] > 0.0f) || !(c->u_max[j] >= c->u_min[j]))
            return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: bad noise_sigma / bounds");
    if (!(c->gamma > 0.0f) || !(c->u_scale != 0.0f) || (c->mode_simple && !(c->lambda_ > 0.0f)))
        return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: bad gamma / u_scale / lambda_");
    // noise_sigma with off-diagonal entries: Cholesky factor (MultivariateNormal, mppi.py:129-131) and inverse
    // (mppi.py:128) in binary64, rounded to f32 (for a diagonal matrix: sqrtf / 1.0f/x bit for bit)
    double chol[M3_MAX_NU * M3_MAX_NU] = {}, sinv[M3_MAX_NU * M3_MAX_NU] = {};
    if (c->full_sigma) {
        const int n = c->nu;
        for (int i = 0; i < n; ++i) {
            if (c->noise_sigma_full[i * n + i] != c->noise_sigma_diag[i])
                return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: diagonal of noise_sigma_full != noise_sigma_diag");
            for (int j = 0; j < i; ++j)
                if (c->noise_sigma_full[i * n + j] != c->noise_sigma_full[j * n + i])
                    return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: noise_sigma_full is not symmetric");
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = (double)c->noise_sigma_full[i * n + j];
                for (int q = 0; q < j; ++q) s -= chol[i * n + q] * chol[j * n + q];
                if (i == j) {
                    if (!(s > 0.0)) return fail(nullptr, M3_ERR_BAD_ARG, "m3_create: noise_sigma_full is not positive definite");
                    chol[i * n + i] = std::sqrt(s);
                } else chol[i * n + j] = s / chol[j * n + j];
            }
        // Sigma^-1 = L^-T L^-1
        double li[M3_MAX_NU * M3_MAX_NU] = {};
        for (int j = 0; j < n; ++j) {
            li[j * n + j] = 1.0 / chol[j * n + j];
            for (int i = j + 1; i < n; ++i) {
                double s = 0.0;
                for (int q = j; q < i; ++q) s -= chol[i * n + q] * li[q * n + j];
                li[i * n + j] = s / chol[i * n + i];
            }
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int q = (i > j ? i : j); q < n; ++q) s += li[q * n + i] * li[q * n + j];
                sinv[i * n + j] = s;
            }
    }
    const bool single_halton = !c->multi_modal && !c->mode_simple;
    if (c->update_cov && single_halton && c->K_local != c->K_global)
        return fail(nullptr, M3_ERR_UNSUPPORTED, "m3_create: update_cov needs an unsharded handle");

    m3_handle* h = new (std::nothrow) m3_handle();
    if (!h) return fail(nullptr, M3_ERR_HIP, "m3_create: out of host memory");
    h->cfg = *c;
    h->cov_active = c->update_cov && single_halton && !c->sim_only;   // (elsewhere the reference ignores the flag)
    h->regen = c->shard_mix && c->K_local != c->K_global && c->multi_modal && !c->mode_simple;
    h->regen_fast = h->regen && c->shard_mix >= 2;
    h->p3 = h->regen && c->shard_mix == 3;   // ... and O(K_local) work after the gather + a second small exchange
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) {
        g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        delete h;
        return M3_ERR_HIP;
    }
    if (!c->sim_only && init_ladder_table() != 0) {
        g_create_err = "m3_create: could not initialise the beta ladder table";
        delete h;
        return M3_ERR_HIP;
    }
    build_point_scene(*c, h->scene);
    default_world(h->world0);
    build_panda_scene(*c, h->pscene);
    default_panda_world(h->pworld0, c->cube_on_shelf);
    const long long Kl = c->K_local, Kg = c->K_global, T = c->T, nu = c->nu;
    const long long f = sizeof(float);
    int rc = M3_OK;
    auto A = [&](int id, long long bytes) { if (rc == M3_OK) rc = alloc_buf(h, id, bytes); };
    if (c->sim_only) {
        A(M3_BUF_INFO, sizeof(m3_info));
        if (rc != M3_OK) { g_create_err = h->err; m3_destroy(h); return rc; }
        (void)hipDeviceSynchronize();
        *out = h;
        return M3_OK;
    }
    A(M3_BUF_STATES, T * Kl * 4 * f);
    A(M3_BUF_ACTIONS, T * Kl * nu * f);
    A(M3_BUF_COST_HORIZON, T * Kl * f);
    if (!h->regen) A(M3_BUF_TRAJ_COST, Kl * f);   // (regen: the costs are the head of the record)
    A(M3_BUF_TRAJ_COST_ALL, Kg * f);
    A(M3_BUF_WEIGHTS, Kg * f);
    A(M3_BUF_WEIGHTS_1, (Kg / 2) * f);
    A(M3_BUF_WEIGHTS_2, (Kg - Kg / 2) * f);
    for (int id = M3_BUF_MEAN; id <= M3_BUF_ACTION_OUT; ++id) A(id, T * nu * f);
    A(M3_BUF_TOP_IDX, M3_TOPK * sizeof(int));
    A(M3_BUF_TOP_TRAJS, M3_TOPK * T * 2 * f);
    A(M3_BUF_REDUCE, (long long)reduce_length((int)T, (int)nu) * f);
    if (!h->regen) A(M3_BUF_NOISE, T * Kl * nu * f);
    A(M3_BUF_PENDING_FORCE, 4 * Kl * f);
    A(M3_BUF_INFO, sizeof(m3_info));
    A(M3_BUF_COV, 2 * nu * f);
    if (rc == M3_OK) {   // cov_action = diag(noise_sigma), scale_tril = sqrt(cov_action): mppi.py:175-176
        float cv[2 * M3_MAX_NU];
        for (int j = 0; j < nu; ++j) { cv[j] = c->noise_sigma_diag[j]; cv[nu + j] = std::sqrt(c->noise_sigma_diag[j]); }
        if (hipMemcpy(h->buf[M3_BUF_COV], cv, (size_t)(2 * nu * f), hipMemcpyHostToDevice) != hipSuccess) rc = M3_ERR_HIP;
    }
    if (rc == M3_OK && c->full_sigma) {
        float mats[2 * M3_MAX_NU * M3_MAX_NU];
        for (int i = 0; i < nu * nu; ++i) { mats[i] = (float)chol[i]; mats[nu * nu + i] = (float)sinv[i]; }
        if (hipMalloc((void**)&h->noise_mats, (size_t)(2 * nu * nu * f)) != hipSuccess ||
            hipMemcpy(h->noise_mats, mats, (size_t)(2 * nu * nu * f), hipMemcpyHostToDevice) != hipSuccess) rc = M3_ERR_HIP;
    }
    if (h->regen) {
        const long long rl = regen_record_length((int)Kl, (int)T);
        A(M3_BUF_RECORD, rl * f);
        A(M3_BUF_RECORDS_ALL, (Kg / Kl) * rl * f);
        if (rc == M3_OK && hipMalloc((void**)&h->noise_all, (size_t)(T * Kg * nu * f)) != hipSuccess) rc = M3_ERR_HIP;
        if (rc == M3_OK && hipMemsetAsync(h->noise_all, 0, (size_t)(T * Kg * nu * f), h->stream) != hipSuccess) rc = M3_ERR_HIP;
        if (rc == M3_OK && hipMalloc((void**)&h->local_top_idx, M3_TOPK * sizeof(int)) != hipSuccess) rc = M3_ERR_HIP;
        if (h->p3) {
            A(M3_BUF_RECORD_B, (long long)recb_length((int)T, (int)nu) * f);
            A(M3_BUF_RECORDS_B_ALL, (Kg / Kl) * (long long)recb_length((int)T, (int)nu) * f);
        }
        if (rc == M3_OK) {   // non-owning aliases (m3_destroy skips them)
            h->buf[M3_BUF_TRAJ_COST] = h->buf[M3_BUF_RECORD]; h->nbytes[M3_BUF_TRAJ_COST] = Kl * f;
            h->buf[M3_BUF_NOISE] = h->noise_all + (size_t)(c->k_offset / Kl) * T * Kl * nu; h->nbytes[M3_BUF_NOISE] = T * Kl * nu * f;
            h->buf[M3_BUF_NOISE_ALL] = h->noise_all; h->nbytes[M3_BUF_NOISE_ALL] = T * Kg * nu * f;
        }
    } else if (c->shard_mix && Kl != Kg) {
        A(M3_BUF_RECORD, (long long)record_length((int)T, (int)nu) * f);
        A(M3_BUF_RECORDS_ALL, (Kg / Kl) * (long long)record_length((int)T, (int)nu) * f);
    }
    if (rc == M3_OK && hipMalloc((void**)&h->world0_dev, 18 * f) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && hipMalloc((void**)&h->topk_cand, (size_t)topk_workgroups((int)Kg) * M3_TOPK * sizeof(VI)) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && hipMalloc((void**)&h->part_min, (size_t)mins_workgroups((int)Kg) * 3 * f) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && hipMalloc((void**)&h->lad, (size_t)ladder_workgroups((int)Kg) * 96 * 3 * f) != hipSuccess) rc = M3_ERR_HIP;
    // the three-launch update of an unsharded multi-modal handle beyond k_update_small's range (update.hip: k_ladder_search)
    const bool fused_large = Kl == Kg && c->multi_modal && !c->mode_simple && Kg > 4096 && Kg <= 131072 && T * nu <= 2048;
    if (rc == M3_OK && hipMalloc((void**)&h->wpart, (size_t)((h->regen || fused_large) ? std::max(wsum_chunks((int)Kg), regen_chunks((int)Kg)) : wsum_chunks((int)Kl)) * 3 * T * nu * f) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && hipMalloc((void**)&h->apart, (size_t)(16 + std::max(apply_workgroups((int)Kg), (h->regen || fused_large) ? regen_chunks((int)Kg) : 0) * 8) * f) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && fused_large) {
        const size_t nl = (size_t)ladder_workgroups((int)Kg);
        if (hipMalloc((void**)&h->wave_min, (size_t)Kl * 3 * f) != hipSuccess) rc = M3_ERR_HIP;      // (one row per rollout workgroup: <= K_local with one lane per wavefront)
        if (rc == M3_OK && hipMalloc((void**)&h->lflag, nl * sizeof(int)) != hipSuccess) rc = M3_ERR_HIP;
        if (rc == M3_OK && hipMemset(h->lflag, 0, nl * sizeof(int)) != hipSuccess) rc = M3_ERR_HIP;
    }
    if (rc == M3_OK && c->env_type == M3_ENV_PANDA) {
        // Everything a panda command may need is allocated HERE (no allocation in m3_command).  The word of
        // mapped host memory the rollout's last wavefront reports its near share into + its counter; and, for a handle
        // whose reach task takes quirk Q8's shadow slots (unsharded), the records between the rollout and
        // k_panda_reach_cost ([T][17][K] floats: 5.4 MB at C4).
        void* p = nullptr;
        void* q = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&p, sizeof(int), hipHostMallocMapped) != hipSuccess || hipMalloc(&q, sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(q, 0, sizeof(unsigned long long)) != hipSuccess || hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
            if (p) (void)hipHostFree(p);
            if (q) (void)hipFree(q);
            h->err = "m3_create: the panda rollout's report word (mapped host memory) could not be allocated";
            rc = M3_ERR_HIP;
        } else {
            h->panda_busy_hint = (int*)p;
            h->panda_busy_hint_dev = (int*)d;
            h->panda_busy_count = (unsigned long long*)q;
            *h->panda_busy_hint = 0;
        }
        if (rc == M3_OK && Kl == Kg && Kg >= 2 && Kl <= PANDA_REACH_REC_MAX_K &&
            hipMalloc((void**)&h->panda_reach_rec, (size_t)T * REACH_REC * (size_t)Kl * f) != hipSuccess) rc = M3_ERR_HIP;
    }
    if (rc == M3_OK && hipMalloc((void**)&h->wcount, (size_t)(T + 2) * sizeof(int)) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK && hipMemset(h->wcount, 0, (size_t)(T + 2) * sizeof(int)) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK) {
        m3_info init;
        std::memset(&init, 0, sizeof(init));
        init.beta = init.beta_1 = init.beta_2 = 1.0f;  // mppi.py:184-187
        if (hipMemcpy(h->buf[M3_BUF_INFO], &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) rc = M3_ERR_HIP;
    }
    if (rc != M3_OK) {
        g_create_err = h->err.empty() ? "m3_create: device allocation failed" : h->err;
        m3_destroy(h);
        return rc;
    }
    (void)hipDeviceSynchronize();
    *out = h;
    return M3_OK;
}

extern "C" void m3_destroy(m3_handle* h) {
    if (!h) return;
    if (h->regen) h->buf[M3_BUF_TRAJ_COST] = h->buf[M3_BUF_NOISE] = h->buf[M3_BUF_NOISE_ALL] = nullptr;   // aliases
    for (int i = 0; i < M3_BUF_COUNT; ++i)
        if (h->buf[i]) (void)hipFree(h->buf[i]);
    if (h->noise_all) (void)hipFree(h->noise_all);
    if (h->noise_mats) (void)hipFree(h->noise_mats);
    if (h->local_top_idx) (void)hipFree(h->local_top_idx);
    if (h->world0_dev) (void)hipFree(h->world0_dev);
    if (h->topk_cand) (void)hipFree(h->topk_cand);
    if (h->part_min) (void)hipFree(h->part_min);
    if (h->lad) (void)hipFree(h->lad);
    if (h->wpart) (void)hipFree(h->wpart);
    if (h->apart) (void)hipFree(h->apart);
    if (h->wcount) (void)hipFree(h->wcount);
    if (h->wave_min) (void)hipFree(h->wave_min);
    if (h->lflag) (void)hipFree(h->lflag);
    if (h->sim_world) (void)hipFree(h->sim_world);
    if (h->sim_u) (void)hipFree(h->sim_u);
    if (h->noise_stage) (void)hipFree(h->noise_stage);
    if (h->order) (void)hipFree(h->order);
    if (h->order_scratch) (void)hipFree(h->order_scratch);
    if (h->noise_sorted) (void)hipFree(h->noise_sorted);
    for (int p = 0; p < MIX_MAX_RANKS; ++p)
        if (h->peer_ipc[p] && h->peer_base[p]) (void)hipIpcCloseMemHandle(h->peer_base[p]);
    if (h->xb) (void)hipFree(h->xb);
    if (h->panda_busy_hint) (void)hipHostFree(h->panda_busy_hint);
    if (h->panda_busy_count) (void)hipFree(h->panda_busy_count);
    if (h->panda_reach_rec) (void)hipFree(h->panda_reach_rec);
    for (auto& ev : h->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete h;
}

#ifndef M3_BUILD_ID
#define M3_BUILD_ID "unknown"
#endif
extern "C" const char* m3_build_id(void) { return M3_BUILD_ID; }

extern "C" int m3_set_stream(m3_handle* h, void* s) {
    if (!h) return M3_ERR_BAD_ARG;
    h->stream = (hipStream_t)s;
    return M3_OK;
}

extern "C" int m3_set_rollout_lanes(m3_handle* h, int lanes) {
    if (!h) return M3_ERR_BAD_ARG;
    if (lanes != 0 && (lanes < 1 || lanes > 64 || (lanes & (lanes - 1)) != 0))
        return fail(h, M3_ERR_BAD_ARG, "m3_set_rollout_lanes: lanes must be 0 (auto) or a power of two in 1..64");
    h->lanes_override = lanes;
    return M3_OK;
}

extern "C" int m3_set_panda_lanes_per_sample(m3_handle* h, int lps) {
    if (!h) return M3_ERR_BAD_ARG;
    if (lps != 0 && lps != 1 && lps != 8 && lps != 16)
        return fail(h, M3_ERR_BAD_ARG, "m3_set_panda_lanes_per_sample: 0 (by size), 1, 8 or 16");
    h->panda_lps = lps;
    return M3_OK;
}

extern "C" int m3_set_panda_reach_cost_kernel(m3_handle* h, int on) {
    if (!h) return M3_ERR_BAD_ARG;
    h->panda_reach_deferred = on != 0;
    return M3_OK;
}

extern "C" int m3_panda_near_share(m3_handle* h) {
    return (h && h->panda_busy_hint) ? *(volatile const int*)h->panda_busy_hint - 1 : -1;
}

extern "C" int m3_panda_lanes_per_sample_used(m3_handle* h) {
    return h ? h->panda_lps_used : 0;
}

extern "C" int m3_set_update_launches(m3_handle* h, int launches) {
    if (!h) return M3_ERR_BAD_ARG;
    if (launches != 0 && launches != 3 && launches != 5) return fail(h, M3_ERR_BAD_ARG, "m3_set_update_launches: 0 (default), 3 or 5");
    h->five_launches = launches == 5;
    return M3_OK;
}

extern "C" int m3_set_ladder_spins(m3_handle* h, int spins) {
    if (!h) return M3_ERR_BAD_ARG;
    if (spins < -1) return fail(h, M3_ERR_BAD_ARG, "m3_set_ladder_spins: -1 (default), 0 (never wait) or a positive bound");
    h->ladder_spins = spins < 0 ? (1 << 18) : spins;
    return M3_OK;
}

extern "C" int m3_enable_timing(m3_handle* h, int on) {
    if (!h) return M3_ERR_BAD_ARG;
    if (on && !h->ev[0])
        for (auto& ev : h->ev) HIPCHK(h, hipEventCreate(&ev));
    h->timing = on != 0;
    return M3_OK;
}

// (re)computes the wavefront order of the samples (sampler.hip) from the noise buffer and the
// current world; stream-ordered, no sync.  Called by m3_rollout when the noise changed and every
// ORDER_REFRESH commands (the objects move).
constexpr unsigned ORDER_REFRESH = 256;
// one shard's block of noise rows [T][Kl][nu] (k_offset = global index of its first sample): sort into
// wavefront order; `relabel`: write the rows back in that order (and, for this rank's own block, let the
// pending suction forces follow their samples)
static int order_block(m3_handle* h, float* block, int k_offset, bool relabel, bool own) {
    const m3_config& c = h->cfg;
    long long half_local = (long long)(c.K_global / 2) - k_offset;
    if (!c.multi_modal || half_local > c.K_local) half_local = c.K_local;
    if (half_local < 0) half_local = 0;
    OrderScene os;
    std::memset(&os, 0, sizeof(os));
    if (h->bind_dof) { os.sim_root = h->bind_root; os.sim_box = h->bind_box; os.sim_dyn = h->bind_dyn; }
    os.bx = h->world0[4]; os.by = h->world0[5]; os.dx = h->world0[11]; os.dy = h->world0[12];
    os.ox = h->scene.obs_x; os.oy = h->scene.obs_y;
    // relabelling keeps the samples with a role of their own at their index
    const int specials[3] = {0 - k_offset, c.K_global / 2 - k_offset, c.K_global - 1 - k_offset};
    hipError_t e = launch_wave_order(block, c.K_local, c.T, c.nu,
                                     std::sqrt(c.noise_sigma_diag[0]), std::sqrt(c.noise_sigma_diag[1]),
                                     (int)half_local, os, h->order_scratch, h->order_temp_bytes, h->order, h->noise_sorted,
                                     relabel ? specials : nullptr, h->stream);
    if (e != hipSuccess) { h->err = std::string("wave order: ") + hipGetErrorString(e); return M3_ERR_HIP; }
    if (relabel) {
        // sample i := old sample order[i]: the noise rows (gathered above) and the per-sample state
        // that outlives a command (pending suction forces) move; everything else is rewritten by
        // the next rollout.  From here on the index order IS the wavefront order: coalesced stores.
        HIPCHK(h, hipMemcpyAsync(block, h->noise_sorted, sizeof(float) * (size_t)c.T * c.K_local * c.nu,
                                 hipMemcpyDeviceToDevice, h->stream));
        if (own) {
            launch_gather_rows((const float*)h->buf[M3_BUF_PENDING_FORCE], h->order, h->noise_sorted, c.K_local, 4, h->stream);
            HIPCHK(h, hipMemcpyAsync(h->buf[M3_BUF_PENDING_FORCE], h->noise_sorted, sizeof(float) * 4 * (size_t)c.K_local,
                                     hipMemcpyDeviceToDevice, h->stream));
        }
    }
    return M3_OK;
}

static int refresh_wave_order(m3_handle* h) {
    const m3_config& c = h->cfg;
    h->order_valid = false;
    h->order_dirty = false;
    const bool relabel = h->relabel_pending;
    h->relabel_pending = false;
    if (h->relabelled && !relabel) return M3_OK;   // rows already in wavefront order: by index
    // (a caller that uploads fresh noise for almost every command would pay the sort -- ~25 us --
    // more often than it pays back: by index then)
    if (!h->wave_order || c.env_type != M3_ENV_POINT || c.nu != 2 || c.K_local < 128 || c.sampling_random ||
        c.mode_simple || !h->have_noise || h->noise_churn >= 2)
        return M3_OK;
    if (!h->order) {
        h->order_temp_bytes = wave_order_temp_bytes(c.K_local);
        HIPCHK(h, hipMalloc((void**)&h->order, sizeof(int) * (size_t)c.K_local));
        HIPCHK(h, hipMalloc(&h->order_scratch, 3 * sizeof(float) * (size_t)c.K_local + h->order_temp_bytes));
        HIPCHK(h, hipMalloc((void**)&h->noise_sorted, sizeof(float) * (size_t)c.T * c.K_local * c.nu));
    }
    if (relabel && h->regen) {
        // every rank holds every shard's noise block (the other ranks' actions are re-generated from
        // them): relabel each block exactly as its owner does -- the same deterministic procedure on the
        // same inputs (noise table, world) -- this rank's own block LAST, so that h->order ends up as its
        // permutation
        const int n = c.K_global / c.K_local, me = c.k_offset / c.K_local;
        const size_t bl = (size_t)c.T * c.K_local * c.nu;
        for (int q = 0; q < n; ++q) {
            const int r = (q < me) ? q : (q + 1 < n ? q + 1 : me);   // 0..me-1, me+1..n-1, me
            const int rc = order_block(h, h->noise_all + (size_t)r * bl, r * c.K_local, true, r == me);
            if (rc != M3_OK) return rc;
        }
        h->relabelled = true;
        return M3_OK;
    }
    const int rc = order_block(h, (float*)h->buf[M3_BUF_NOISE], c.k_offset, relabel, true);
    if (rc != M3_OK) return rc;
    if (relabel) {
        h->relabelled = true;
        return M3_OK;
    }
    h->order_valid = true;
    return M3_OK;
}

extern "C" int m3_relabel_samples(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->have_noise) return fail(h, M3_ERR_STATE, "m3_relabel_samples: no noise set");
    h->relabel_pending = true;
    h->order_dirty = true;
    return M3_OK;
}

static void note_noise_upload(m3_handle* h) {
    h->relabelled = false;
    h->relabel_pending = false;
    h->noise_churn = (h->have_noise && h->calls - h->last_noise_call < 16u) ? h->noise_churn + 1 : 0;
    h->last_noise_call = h->calls;
    h->have_noise = true;
    h->order_dirty = true;
}

extern "C" int m3_set_wave_order(m3_handle* h, int on) {
    if (!h) return M3_ERR_BAD_ARG;
    h->wave_order = on != 0;
    h->order_dirty = true;
    return M3_OK;
}

// rows [n_rows][T][nu] (reference layout) -> n_rows / K_local consecutive time-major blocks at dst
static int upload_noise(m3_handle* h, const float* delta, long long n_rows, float* dst, int on_device, const char* who) {
    if (!h || !delta) return fail(h, M3_ERR_BAD_ARG, "m3_set_noise: null argument");
    const m3_config& c = h->cfg;
    (void)who;
    const size_t bytes = (size_t)n_rows * c.T * c.nu * sizeof(float);
    const float* src = delta;
    float* stage = nullptr;
    if (!on_device) {
        if (n_rows == c.K_local) {
            if (!h->noise_stage) HIPCHK(h, hipMalloc((void**)&h->noise_stage, bytes));
            stage = h->noise_stage;
        } else {
            HIPCHK(h, hipMalloc((void**)&stage, bytes));
        }
        HIPCHK(h, hipMemcpyAsync(stage, delta, bytes, hipMemcpyHostToDevice, h->stream));
        src = stage;
    }
    const size_t bl = (size_t)c.T * c.K_local * c.nu;
    for (long long r = 0; r < n_rows / c.K_local; ++r)
        launch_transpose_noise(src + r * bl, dst + r * bl, c.K_local, c.T, c.nu, h->stream);
    hipError_t e = hipGetLastError();
    if (!on_device) (void)hipStreamSynchronize(h->stream);  // host buffer may be released
    if (stage && stage != h->noise_stage) (void)hipFree(stage);
    if (e != hipSuccess) { h->err = std::string("k_transpose_noise: ") + hipGetErrorString(e); return M3_ERR_HIP; }
    note_noise_upload(h);
    return M3_OK;
}

extern "C" int m3_set_noise(m3_handle* h, const float* delta, int on_device) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->cfg.sim_only) return fail(h, M3_ERR_STATE, "m3_set_noise: handle was created sim_only");
    if (h->regen) return fail(h, M3_ERR_STATE, "m3_set_noise: a one-collective multi-modal shard needs the noise rows of ALL "
                                               "K_global samples (m3_set_noise_global)");
    return upload_noise(h, delta, h->cfg.K_local, (float*)h->buf[M3_BUF_NOISE], on_device, "m3_set_noise");
}

extern "C" int m3_set_noise_global(m3_handle* h, const float* delta_all, int on_device) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->regen) return fail(h, M3_ERR_STATE, "m3_set_noise_global: only for one-collective multi-modal shards "
                                                "(cfg.shard_mix with multi_modal, K_local < K_global)");
    return upload_noise(h, delta_all, h->cfg.K_global, h->noise_all, on_device, "m3_set_noise_global");
}

static int upload_knots(m3_handle* h, const float* knots, long long n_rows, float* dst, int n_knots, int degree,
                        float smoothing, int on_device) {
    if (!h || !knots) return fail(h, M3_ERR_BAD_ARG, "m3_set_noise_knots: null argument");
    const m3_config& c = h->cfg;
    if (c.sim_only) return fail(h, M3_ERR_STATE, "m3_set_noise_knots: handle was created sim_only");
    if (degree < 1 || degree > 3) return fail(h, M3_ERR_BAD_ARG, "m3_set_noise_knots: degree must be 1..3");
    if (n_knots <= degree || n_knots > 64)
        return fail(h, M3_ERR_SHAPE, "m3_set_noise_knots: needs degree < n_knots <= 64 (splrep: m > k must hold)");
    if (!(smoothing >= 0.0f)) return fail(h, M3_ERR_BAD_ARG, "m3_set_noise_knots: smoothing must be >= 0");
    const size_t bytes = (size_t)n_rows * c.nu * n_knots * sizeof(float);
    const float* src = knots;
    float* stage = nullptr;
    if (!on_device) {
        HIPCHK(h, hipMalloc((void**)&stage, bytes));
        HIPCHK(h, hipMemcpyAsync(stage, knots, bytes, hipMemcpyHostToDevice, h->stream));
        src = stage;
    }
    const size_t bl = (size_t)c.T * c.K_local * c.nu, kl = (size_t)c.K_local * c.nu * n_knots;
    for (long long r = 0; r < n_rows / c.K_local; ++r)
        launch_spline_noise(src + r * kl, dst + r * bl, c.K_local, c.nu, n_knots, c.T, degree, (double)smoothing, h->stream);
    hipError_t e = hipGetLastError();
    if (stage) {
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(stage);
    }
    if (e != hipSuccess) { h->err = std::string("k_spline_noise: ") + hipGetErrorString(e); return M3_ERR_HIP; }
    note_noise_upload(h);
    return M3_OK;
}

extern "C" int m3_set_noise_knots(m3_handle* h, const float* knots, int n_knots, int degree, float smoothing,
                                  int on_device) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->regen) return fail(h, M3_ERR_STATE, "m3_set_noise_knots: a one-collective multi-modal shard needs the knots of ALL "
                                               "K_global samples (m3_set_noise_knots_global)");
    return upload_knots(h, knots, h->cfg.K_local, (float*)h->buf[M3_BUF_NOISE], n_knots, degree, smoothing, on_device);
}

extern "C" int m3_set_noise_knots_global(m3_handle* h, const float* knots_all, int n_knots, int degree, float smoothing,
                                         int on_device) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->regen) return fail(h, M3_ERR_STATE, "m3_set_noise_knots_global: only for one-collective multi-modal shards");
    return upload_knots(h, knots_all, h->cfg.K_global, h->noise_all, n_knots, degree, smoothing, on_device);
}

// the whole reference sampler on the device: Halton -> erfinv -> smoothing spline (no host values at all)
// Faure's permutations (H. Faure, "Good permutations for extreme discrepancy", J. Number Theory 42 (1992) 47-56),
// defined recursively: pi_2 = (0 1); for even b = 2k, pi_b = (2 pi_k, 2 pi_k + 1); for odd b = 2k + 1, pi_b is
// pi_2k with every value >= k increased by one and the value k inserted in the middle.  pi_b(0) = 0 for every b.
static std::vector<int> faure_permutation(int b) {
    if (b == 2) return {0, 1};
    std::vector<int> r;
    if (b % 2 == 0) {
        const std::vector<int> hlf = faure_permutation(b / 2);
        for (int x : hlf) r.push_back(2 * x);
        for (int x : hlf) r.push_back(2 * x + 1);
    } else {
        const int k = (b - 1) / 2;
        r = faure_permutation(b - 1);
        for (int& x : r) if (x >= k) x += 1;
        r.insert(r.begin() + k, k);
    }
    return r;
}

extern "C" int m3_set_noise_halton_scrambled(m3_handle* h, int n_knots, int degree, float smoothing, int scramble) {
    if (!h) return M3_ERR_BAD_ARG;
    const m3_config& c = h->cfg;
    if (c.sim_only) return fail(h, M3_ERR_STATE, "m3_set_noise_halton: handle was created sim_only");
    if (scramble != M3_HALTON_PLAIN && scramble != M3_HALTON_FAURE)
        return fail(h, M3_ERR_BAD_ARG, "m3_set_noise_halton_scrambled: scramble must be M3_HALTON_PLAIN or M3_HALTON_FAURE");
    const int ncol = c.nu * n_knots;
    if (n_knots < 1 || ncol > 100) return fail(h, M3_ERR_SHAPE, "m3_set_noise_halton: nu * n_knots must be <= 100 (mppi_utils.py:81)");
    int primes[100], np_ = 0;
    for (int cand = 2; np_ < ncol; ++cand) {
        bool is_p = true;
        for (int q = 0; q < np_ && primes[q] * primes[q] <= cand; ++q)
            if (cand % primes[q] == 0) { is_p = false; break; }
        if (is_p) primes[np_++] = cand;
    }
    std::vector<int> perm, perm_off;
    if (scramble == M3_HALTON_FAURE)
        for (int j = 0; j < ncol; ++j) {
            perm_off.push_back((int)perm.size());
            const std::vector<int> pj = faure_permutation(primes[j]);
            perm.insert(perm.end(), pj.begin(), pj.end());
        }
    const long long rows = h->regen ? c.K_global : c.K_local;
    const int k0 = h->regen ? 0 : c.k_offset;
    float* knots = nullptr;
    int* tab = nullptr;           // primes | perm offsets | permutations
    const size_t ntab = (size_t)ncol + perm_off.size() + perm.size();
    HIPCHK(h, hipMalloc((void**)&knots, (size_t)rows * ncol * sizeof(float)));
    if (hipMalloc((void**)&tab, ntab * sizeof(int)) != hipSuccess) { (void)hipFree(knots); return fail(h, M3_ERR_HIP, "m3_set_noise_halton: hipMalloc"); }
    std::vector<int> host(primes, primes + ncol);
    host.insert(host.end(), perm_off.begin(), perm_off.end());
    host.insert(host.end(), perm.begin(), perm.end());
    int rc = M3_OK;
    if (hipMemcpyAsync(tab, host.data(), ntab * sizeof(int), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = M3_ERR_HIP;
    if (rc == M3_OK) {
        const bool sc = scramble != M3_HALTON_PLAIN;
        launch_halton_knots(knots, k0, (int)rows, ncol, tab, sc ? tab + 2 * ncol : nullptr, sc ? tab + ncol : nullptr, h->stream);
        // knots [rows][nu][n_knots] == [rows][ncol]: column j * n_knots + q is knot q of control dimension j
        rc = upload_knots(h, knots, rows, h->regen ? h->noise_all : (float*)h->buf[M3_BUF_NOISE], n_knots, degree, smoothing, 1);
    }
    (void)hipStreamSynchronize(h->stream);     // (also keeps `host` alive until the copy is done)
    (void)hipFree(knots);
    (void)hipFree(tab);
    if (rc != M3_OK && h->err.empty()) h->err = "m3_set_noise_halton failed";
    return rc;
}

extern "C" int m3_set_noise_halton(m3_handle* h, int n_knots, int degree, float smoothing) {
    return m3_set_noise_halton_scrambled(h, n_knots, degree, smoothing, M3_HALTON_PLAIN);
}

extern "C" int m3_sample_noise(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    const m3_config& c = h->cfg;
    if (c.sim_only || !(c.sampling_random || c.mode_simple))
        return fail(h, M3_ERR_STATE, "m3_sample_noise: the handle has a noise table (sampling_random == 0)");
    if (h->regen) return fail(h, M3_ERR_STATE, "m3_sample_noise: not on a one-collective multi-modal shard");
    RolloutArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Kg = c.K_global; a.Kl = c.K_local; a.k0 = c.k_offset; a.T = c.T; a.nu = c.nu;
    a.full_sigma = c.full_sigma; a.noise_mats = h->noise_mats;
    for (int j = 0; j < c.nu; ++j) { a.noise_mu[j] = c.noise_mu[j]; a.scale_tril[j] = std::sqrt(c.noise_sigma_diag[j]); }
    a.seed = c.seed; a.call = h->calls;
    launch_sample_noise(a, (float*)h->buf[M3_BUF_NOISE], h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_set_objective(m3_handle* h, int task, const float* goal, int goal_len, int gripper_cmd) {
    if (!h) return M3_ERR_BAD_ARG;
    if (task < 0 || task > M3_TASK_IDLE) return fail(h, M3_ERR_BAD_ARG, "m3_set_objective: unknown task");
    if (goal_len < 0 || goal_len > 7 || (goal_len > 0 && !goal)) return fail(h, M3_ERR_SHAPE, "m3_set_objective: bad goal");
    if (h->cfg.env_type == M3_ENV_POINT && task > M3_TASK_PUSH_PULL)
        return fail(h, M3_ERR_UNSUPPORTED, "m3_set_objective: task not defined for point_env");
    if (h->cfg.env_type == M3_ENV_POINT && goal_len < 2) return fail(h, M3_ERR_SHAPE, "m3_set_objective: point_env goal needs 2 values");
    if (h->cfg.env_type == M3_ENV_PANDA && task <= M3_TASK_PUSH_PULL)
        return fail(h, M3_ERR_UNSUPPORTED, "m3_set_objective: task not defined for panda_env");
    if (h->cfg.env_type == M3_ENV_PANDA && task == M3_TASK_PICK && goal_len < 7)
        return fail(h, M3_ERR_SHAPE, "m3_set_objective: pick needs a 7-value goal pose (cost_functions.py:116-123)");
    if (task == M3_TASK_PUSH_PULL && !h->cfg.multi_modal)
        return fail(h, M3_ERR_STATE, "m3_set_objective: push_pull needs multi_modal (cost_functions.py:27-29)");
    h->task = task;
    for (int i = 0; i < goal_len; ++i) h->goal[i] = goal[i];
    h->gripper_cmd = gripper_cmd;
    return M3_OK;
}

extern "C" int m3_set_avoid_dyn_obs(m3_handle* h, int on) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->cfg.env_type != M3_ENV_POINT) return fail(h, M3_ERR_UNSUPPORTED, "m3_set_avoid_dyn_obs: point_env only");
    h->avoid_dyn_obs = on != 0;
    return M3_OK;
}

extern "C" int m3_set_multi_modal(m3_handle* h, int mm) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->cfg.sim_only && (mm != 0) != (h->cfg.multi_modal != 0))
        return fail(h, M3_ERR_STATE, "m3_set_multi_modal: fixed at m3_create for planner handles");
    h->cfg.multi_modal = mm != 0;
    return M3_OK;
}

extern "C" int m3_set_plan(m3_handle* h, int which, const float* v) {
    if (!h || !v) return M3_ERR_BAD_ARG;
    if (which < M3_BUF_MEAN || which > M3_BUF_BEST_2) return fail(h, M3_ERR_BAD_ARG, "m3_set_plan: not a plan buffer");
    HIPCHK(h, hipMemcpyAsync(h->buf[which], v, (size_t)h->cfg.T * h->cfg.nu * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return M3_OK;
}

extern "C" int m3_set_beta(m3_handle* h, float beta) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!(beta > 0.0f)) return fail(h, M3_ERR_BAD_ARG, "m3_set_beta: beta must be > 0");
    HIPCHK(h, hipMemcpyAsync((char*)h->buf[M3_BUF_INFO] + offsetof(m3_info, beta), &beta, sizeof(float),
                             hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // `beta` is a stack value
    return M3_OK;
}

extern "C" int m3_set_call_count(m3_handle* h, unsigned calls) {
    if (!h) return M3_ERR_BAD_ARG;
    h->calls = calls;
    // (a rollout still in flight may be about to report into the word zeroed below)
    if (h->panda_busy_hint) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->panda_busy_hint) { *h->panda_busy_hint = 0; h->panda_reach_busy = 0; }
    return M3_OK;
}

extern "C" int m3_set_action_out(m3_handle* h, float* dev_ptr) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->cfg.sim_only) return fail(h, M3_ERR_STATE, "m3_set_action_out: handle was created sim_only");
    h->action_out = dev_ptr;
    return M3_OK;
}

extern "C" int m3_reset(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    for (int id = M3_BUF_MEAN; id <= M3_BUF_ACTION_OUT; ++id)
        HIPCHK(h, hipMemsetAsync(h->buf[id], 0, (size_t)h->nbytes[id], h->stream));
    HIPCHK(h, hipMemsetAsync(h->buf[M3_BUF_PENDING_FORCE], 0, (size_t)h->nbytes[M3_BUF_PENDING_FORCE], h->stream));
    m3_info init;
    std::memset(&init, 0, sizeof(init));
    init.beta = init.beta_1 = init.beta_2 = 1.0f;
    HIPCHK(h, hipMemcpyAsync(h->buf[M3_BUF_INFO], &init, sizeof(init), hipMemcpyHostToDevice, h->stream));
    // update_cov adapts cov_action / scale_tril (mppi.py:508-516): a reset planner starts from the configured
    // diag(noise_sigma) again, like a freshly built one (mppi.py:175-176)
    float cv[2 * M3_MAX_NU];
    const int nu = h->cfg.nu;
    for (int j = 0; j < nu; ++j) { cv[j] = h->cfg.noise_sigma_diag[j]; cv[nu + j] = std::sqrt(h->cfg.noise_sigma_diag[j]); }
    HIPCHK(h, hipMemcpyAsync(h->buf[M3_BUF_COV], cv, (size_t)(2 * nu) * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->calls = 0;
    if (h->panda_busy_hint) { *h->panda_busy_hint = 0; h->panda_reach_busy = 0; }
    return M3_OK;
}

extern "C" int m3_set_world_point(m3_handle* h, const m3_point_world* w) {
    if (!h || !w) return M3_ERR_BAD_ARG;
    float* o = h->world0;
    o[0] = w->robot[0]; o[1] = w->robot[1]; o[2] = w->robot[2]; o[3] = w->robot[3];
    const float* src[2] = {w->box, w->dyn_obs};
    for (int b = 0; b < 2; ++b) {
        const float* r = src[b];
        float* p = o + 4 + b * 7;
        const float qz = r[2], qw = r[3];
        p[0] = r[0]; p[1] = r[1];
        p[2] = 1.0f - 2.0f * (qz * qz);  // yaw from the (0,0,qz,qw) quaternion
        p[3] = 2.0f * (qz * qw);
        p[4] = r[4]; p[5] = r[5]; p[6] = r[6];
    }
    h->world0_bound = nullptr;
    h->bind_dof = nullptr;
    return M3_OK;
}

// internal-format variant used by the parity tests: 18 floats, boxes as (x, y, cos, sin, ...)
extern "C" int m3_set_world_point_raw(m3_handle* h, const float* w18) {
    if (!h || !w18) return M3_ERR_BAD_ARG;
    std::memcpy(h->world0, w18, 18 * sizeof(float));
    h->world0_bound = nullptr;
    h->bind_dof = nullptr;
    return M3_OK;
}

extern "C" int m3_bind_sim_point(m3_handle* h, const float* dof, const float* root, int n_actors, int box_actor, int dyn_actor) {
    if (!h || !dof || !root) return M3_ERR_BAD_ARG;
    if (n_actors < 1 || box_actor < 0 || box_actor >= n_actors || dyn_actor < 0 || dyn_actor >= n_actors)
        return fail(h, M3_ERR_SHAPE, "m3_bind_sim_point: actor index out of range");
    h->bind_dof = dof; h->bind_root = root; h->bind_nact = n_actors; h->bind_box = box_actor; h->bind_dyn = dyn_actor;
    return M3_OK;
}

static void fill_cost_params(const m3_handle* h, CostParams& cp) {
    cp.task = h->task;
    cp.multi_modal = h->cfg.multi_modal;
    cp.half_K = h->cfg.K_global / 2;
    for (int i = 0; i < 7; ++i) cp.goal[i] = h->goal[i];
    cp.kp_suction = h->cfg.kp_suction;
    cp.suction_thresh = (h->cfg.K_global == 1) ? 1.5f : 1.8f;  // skill_utils.py:75-82
    cp.avoid_dyn_obs = h->avoid_dyn_obs;
}

static void fill_panda_cost_params(const m3_handle* h, PandaCostParams& cp) {
    cp.task = h->task;
    cp.multi_modal = h->cfg.multi_modal;
    cp.half_K = h->cfg.K_global / 2;
    for (int i = 0; i < 7; ++i) cp.goal[i] = h->goal[i];
    cp.pre_height_diff = h->cfg.pre_height_diff;
    cp.tilt_cos_theta = 0.5f;  // cost_functions.py:13
}

// panda_env initial state, 57 floats: q[9] qd[9] | cubeA | cubeB | dyn-obs, each pos3 quat4(xyzw) linvel3 angvel3
extern "C" int m3_set_world_panda_raw(m3_handle* h, const float* w57) {
    const float* w31 = w57;
    if (!h || !w31) return M3_ERR_BAD_ARG;
    if (h->cfg.env_type != M3_ENV_PANDA) return fail(h, M3_ERR_STATE, "m3_set_world_panda_raw: not a panda_env handle");
    std::memcpy(h->pworld0, w31, 57 * sizeof(float));
    h->bind_dof = nullptr;
    return M3_OK;
}

extern "C" int m3_bind_sim_panda(m3_handle* h, const float* dof, const float* root, int n_actors, int cubeA_actor, int cubeB_actor,
                                 int obs_actor) {
    if (!h || !dof || !root) return M3_ERR_BAD_ARG;
    if (h->cfg.env_type != M3_ENV_PANDA) return fail(h, M3_ERR_STATE, "m3_bind_sim_panda: not a panda_env handle");
    if (n_actors < 1 || cubeA_actor < 0 || cubeA_actor >= n_actors || cubeB_actor < 0 || cubeB_actor >= n_actors ||
        obs_actor < 0 || obs_actor >= n_actors)
        return fail(h, M3_ERR_SHAPE, "m3_bind_sim_panda: actor index out of range");
    h->bind_dof = dof; h->bind_root = root; h->bind_nact = n_actors; h->bind_box = cubeA_actor; h->bind_dyn = cubeB_actor;
    h->bind_obs = obs_actor;
    return M3_OK;
}

// m3_rollout's choice of the reach command's kernel form: share (1/1000) of the last finished command's (sample, substep) pairs with
// the gripper within reach of a box from which eight lanes per sample take over / below which one lane does again
// (tools/panda_reach_mid_bench.py, profiles/r05/panda_reach_mid_bench.json; K = 4000, T = 20, rollout ms with 1 / 8 lanes: the arm at
// its initial pose 148 per mille 0.166 / 0.210, 20 ticks into an episode 116: 0.179 / 0.206, 30 ticks 442: 0.383 / 0.287, 40 ticks
// 842: 0.784 / 0.450, 60 ticks 993: 1.530 / 0.831 -- the forms cross near 240; the eight-lane form's own count runs ~10 % higher,
// its shadow slots included)

extern "C" int m3_rollout(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    const m3_config& c = h->cfg;
    if (c.sim_only) return fail(h, M3_ERR_STATE, "m3_rollout: handle was created sim_only");
    if (!c.sampling_random && !c.mode_simple && !h->have_noise)
        return fail(h, M3_ERR_STATE, "m3_rollout: no noise set (m3_set_noise) and sampling_random == 0");
    if (c.mode_simple && !c.sampling_random && !h->have_noise)
        return fail(h, M3_ERR_STATE, "m3_rollout: simple mode needs m3_set_noise or sampling_random");
    if (h->task == M3_TASK_PUSH_PULL && !c.multi_modal) return fail(h, M3_ERR_STATE, "m3_rollout: push_pull needs multi_modal");
    RolloutArgs a;
    std::memset(&a, 0, sizeof(a));
    a.Kg = c.K_global; a.Kl = c.K_local; a.k0 = c.k_offset; a.T = c.T; a.nu = c.nu;
    a.multi_modal = c.multi_modal; a.mode_simple = c.mode_simple;
    a.sampling_random = c.sampling_random; a.sample_null_action = c.sample_null_action;
    a.gripper_cmd = h->gripper_cmd;
    for (int j = 0; j < c.nu; ++j) {
        a.u_min[j] = c.u_min[j]; a.u_max[j] = c.u_max[j];
        a.scale_tril[j] = std::sqrt(c.noise_sigma_diag[j]);  // mppi.py:175-176
        a.sigma_inv[j] = 1.0f / c.noise_sigma_diag[j];       // mppi.py:128 (diagonal)
    }
    a.u_scale = c.u_scale; a.gamma = c.gamma; a.lambda_ = c.lambda_;
    a.noise_abs_cost = c.noise_abs_cost; a.full_sigma = c.full_sigma;
    for (int j = 0; j < c.nu; ++j) a.noise_mu[j] = c.noise_mu[j];
    a.noise_mats = h->noise_mats;
    a.scale_dev = h->cov_active ? (const float*)h->buf[M3_BUF_COV] + c.nu : nullptr;
    a.seed = c.seed; a.call = h->calls;
    fill_cost_params(h, a.cp);
    std::memcpy(a.world0, h->world0, sizeof(a.world0));
    if (h->bind_dof) {
        a.sim_dof = h->bind_dof; a.sim_root = h->bind_root;
        a.sim_box = h->bind_box; a.sim_dyn = h->bind_dyn;
    }
    a.lanes = h->lanes_override > 0 ? h->lanes_override : rollout_lanes_for(c.K_local);
    a.delta = (const float*)h->buf[M3_BUF_NOISE];
    if (h->relabelled && h->wave_order && h->calls % ORDER_REFRESH == 0 && h->calls != h->last_noise_call)
        h->relabel_pending = true;   // the objects move: relabel again now and then (labels carry no meaning)
    if (h->order_dirty || h->relabel_pending || (h->wave_order && h->calls % ORDER_REFRESH == 0)) {
        const int rc = refresh_wave_order(h);
        if (rc != M3_OK) return rc;
    }
    a.order = h->order_valid ? h->order : nullptr;
    if (a.order) a.delta = h->noise_sorted;
    a.mean = (const float*)h->buf[M3_BUF_MEAN];
    a.mean1 = (const float*)h->buf[M3_BUF_MEAN_1];
    a.mean2 = (const float*)h->buf[M3_BUF_MEAN_2];
    a.best1 = (const float*)h->buf[M3_BUF_BEST_1];
    a.best2 = (const float*)h->buf[M3_BUF_BEST_2];
    a.pend = (float*)h->buf[M3_BUF_PENDING_FORCE];
    a.states = (float*)h->buf[M3_BUF_STATES];
    a.actions = (float*)h->buf[M3_BUF_ACTIONS];
    a.cost_h = (float*)h->buf[M3_BUF_COST_HORIZON];
    a.J = (float*)h->buf[M3_BUF_TRAJ_COST];
    a.wave_min = h->wave_min;   // (null unless the handle's update is the three-launch one)
    h->wave_min_rows = 0;
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    if (c.env_type == M3_ENV_POINT) {
        if (launch_rollout_point(a, h->scene, h->stream)) h->wave_min_rows = (a.Kl + a.lanes - 1) / a.lanes;
    } else {
        PandaArgs pa;
        std::memcpy(pa.world0, h->pworld0, sizeof(pa.world0));
        pa.cubeA_actor = h->bind_box; pa.cubeB_actor = h->bind_dyn; pa.obs_actor = h->bind_obs;
        fill_panda_cost_params(h, pa.cp);
        // quirk Q8: the reach cost is measured against environment 0's cube (shadow lanes, rollout_panda.hip); a rank of
        // a sharded command does not hold sample 0's noise row and uses each sample's own cube
        pa.shadows = (pa.cp.task == 4 && a.k0 == 0 && a.Kl == a.Kg && a.Kg >= 2) ? (pa.cp.multi_modal ? 2 : 1) : 0;
        pa.lps = h->panda_lps;
        // reach on a handle that would need shadow slots: with room for one round of wavefronts in a many-lane form (K <= 8192)
        // the cost moves into a kernel of its own behind the rollout (rollout_panda.hip: k_panda_reach_cost) -- 17 floats per
        // (step, sample) in between
        pa.reach_rec = (pa.shadows != 0 && a.Kl <= PANDA_REACH_REC_MAX_K && h->panda_reach_deferred) ? h->panda_reach_rec : nullptr;
        // the reach command's kernel form follows what the last command's rollouts met (rollout_panda.hip: panda_lps_for): the
        // kernel's last wavefront reports the share of (sample, substep) pairs with the gripper within reach of a box or an awake
        // cube into a word of mapped host memory (m3_create), read here without a synchronisation (so it is the report of the
        // last FINISHED command); a many-lane form from PANDA_BUSY_ON(_REC) per mille, one lane again below PANDA_BUSY_OFF(_REC).
        pa.busy_hint = h->panda_busy_hint_dev; pa.busy_count = h->panda_busy_count; pa.reach_busy = 0;
        if (h->panda_busy_hint) {
            const int share = *(volatile const int*)h->panda_busy_hint - 1;    // (-1: nothing reported yet)
            // (with the cost kernel available the many-lane form has no shadow slots and takes over earlier)
            const int on = pa.reach_rec ? PANDA_BUSY_ON_REC : PANDA_BUSY_ON, off = pa.reach_rec ? PANDA_BUSY_OFF_REC : PANDA_BUSY_OFF;
            if (share >= on) h->panda_reach_busy = 1;
            else if (share >= 0 && share < off) h->panda_reach_busy = 0;
            pa.reach_busy = h->panda_reach_busy;
        }
        const int wgs = launch_rollout_panda(a, pa, h->pscene, h->stream, &h->panda_lps_used);
        if (a.wave_min) h->wave_min_rows = wgs;
    }
    HIPCHK(h, hipGetLastError());
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    return M3_OK;
}

static void fill_update_args(m3_handle* h, UpdateArgs& a) {
    const m3_config& c = h->cfg;
    std::memset(&a, 0, sizeof(a));
    a.Kg = c.K_global; a.Kl = c.K_local; a.k0 = c.k_offset; a.T = c.T; a.nu = c.nu;
    a.multi_modal = c.multi_modal; a.mode_simple = c.mode_simple; a.env_type = c.env_type;
    a.filter_u = c.filter_u; a.u_per_command = c.u_per_command;
    a.lambda_ = c.lambda_; a.step_size_mean = c.step_size_mean;
    a.cand = h->topk_cand;
    a.ladder_spins = h->ladder_spins;
    a.p2p_err = (h->p2p_ready && h->xb) ? (const int*)((const char*)h->xb + 2 * MIX_MAX_RANKS * sizeof(int)) : nullptr;
    a.part_min = h->part_min;
    a.n_cand = topk_workgroups(c.K_global);
    a.n_mins = mins_workgroups(c.K_global);
    a.lad = h->lad;
    a.wpart = h->wpart;
    a.srch = (SearchOut*)h->apart;
    a.apart = h->apart + 16;
    a.wcount = h->wcount;
    a.lflag = h->lflag;
    a.n_chunk = wsum_chunks(c.K_local);
    a.n_lad = ladder_workgroups(c.K_global);
    a.Jall = (const float*)h->buf[M3_BUF_TRAJ_COST_ALL];
    a.w = (float*)h->buf[M3_BUF_WEIGHTS];
    a.w1 = (float*)h->buf[M3_BUF_WEIGHTS_1];
    a.w2 = (float*)h->buf[M3_BUF_WEIGHTS_2];
    a.top_idx = (int*)h->buf[M3_BUF_TOP_IDX];
    a.info = (m3_info*)h->buf[M3_BUF_INFO];
    a.actions = (const float*)h->buf[M3_BUF_ACTIONS];
    a.states = (const float*)h->buf[M3_BUF_STATES];
    a.reduce = (float*)h->buf[M3_BUF_REDUCE];
    a.mean = (float*)h->buf[M3_BUF_MEAN];
    a.mean1 = (float*)h->buf[M3_BUF_MEAN_1];
    a.mean2 = (float*)h->buf[M3_BUF_MEAN_2];
    a.best = (float*)h->buf[M3_BUF_BEST];
    a.best1 = (float*)h->buf[M3_BUF_BEST_1];
    a.best2 = (float*)h->buf[M3_BUF_BEST_2];
    a.action_out = h->action_out ? h->action_out : (float*)h->buf[M3_BUF_ACTION_OUT];
    a.top_trajs = (float*)h->buf[M3_BUF_TOP_TRAJS];
    a.top_dst = (c.K_local == c.K_global) ? a.top_trajs : a.reduce + reduce_off_top(c.T, c.nu);
    a.kbase = 0;
    a.half_g = c.K_global / 2;
    a.n_ranks = c.K_global / c.K_local;
    a.rank = c.k_offset / c.K_local;
    a.records_all = h->records_src ? h->records_src : (const float*)h->buf[M3_BUF_RECORDS_ALL];
    a.rec_topj = a.rec_topi = nullptr;
    a.rec_mins = a.rec_table = nullptr;
    a.rec_b = nullptr;
    a.recb_all = h->recb_src ? h->recb_src : (const float*)h->buf[M3_BUF_RECORDS_B_ALL];
    a.recb_len = h->recb_src ? h->recb_stride : recb_length(c.T, c.nu);
    a.fast = 0;
    a.regen = 0;
    a.Jout = nullptr;
    a.Kls = c.K_local;
    a.rec_len = h->regen ? regen_record_length(c.K_local, c.T) : record_length(c.T, c.nu);
    if (h->records_src) a.rec_len = h->records_stride;   // (the p2p block's slots are padded to 16 bytes; rec_len is only ever a stride)
    a.noise_all = h->noise_all;
    for (int j = 0; j < c.nu; ++j) {
        a.u_min[j] = c.u_min[j]; a.u_max[j] = c.u_max[j];
        a.scale_tril[j] = std::sqrt(c.noise_sigma_diag[j]);   // as m3_rollout
    }
    a.u_scale = c.u_scale;
    a.sample_null_action = c.sample_null_action;
    a.gripper_cmd = h->gripper_cmd;
}

static bool mix_mode(const m3_handle* h) { return h->cfg.shard_mix && h->cfg.K_local != h->cfg.K_global && !h->regen; }

// fuse: m3_command on an unsharded handle lets k_wsum's last workgroup do k_finalize's work
static bool can_fuse_finalize(const m3_handle* h) {
    const m3_config& c = h->cfg;
    return c.K_local == c.K_global && (long long)c.T * c.nu <= 2048;  // plan staged in 8 KB of LDS
}

static int update_impl(m3_handle* h, bool fuse) {
    const m3_config& c = h->cfg;
    if (c.sim_only) return fail(h, M3_ERR_STATE, "m3_update: handle was created sim_only");
    UpdateArgs a;
    fill_update_args(h, a);
    a.fuse_finalize = fuse ? 1 : 0;
    if (c.K_local == c.K_global)  // unsharded: the local costs ARE the global costs (no copy)
        a.Jall = (const float*)h->buf[M3_BUF_TRAJ_COST];
    if (h->regen) {
        // one-collective multi-modal sharding, phase before the all-gather: the rollout left the shard's
        // costs at the head of the record; add the shard's own top-k (costs, global indices, trajectories)
        float* rec = (float*)h->buf[M3_BUF_RECORD];
        a.Kg = c.K_local;                       // the selection runs over the LOCAL costs ...
        a.Jall = rec;
        a.kbase = c.k_offset;                   // ... and reports global indices
        a.n_cand = c.K_local <= 8192 ? 1 : topk_workgroups(c.K_local);
        a.top_idx = h->local_top_idx;
        a.rec_topj = rec + regen_off_topj(c.K_local);
        a.rec_topi = rec + regen_off_topi(c.K_local);
        a.top_dst = rec + regen_off_trajs(c.K_local);
        if (h->regen_fast) {   // ... and its minima + ladder table
            a.fast = 1;
            a.rec_mins = rec + regen_off_mins(c.K_local, c.T);
            a.rec_table = rec + regen_off_table(c.K_local, c.T);
        }
        launch_local_topk(a, h->stream);
        HIPCHK(h, hipGetLastError());
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
        return M3_OK;
    }
    if (mix_mode(h)) {
        // one-collective sharding: softmin over the LOCAL shard into this rank's record
        float* rec = (float*)h->buf[M3_BUF_RECORD];
        a.record = rec;
        a.rec_topj = rec + REC_TOPJ;
        a.rec_topi = rec + REC_TOPI;
        a.reduce = rec + REC_HDR;
        a.top_dst = a.reduce + reduce_off_top(c.T, c.nu);
        a.n_cand = topk_workgroups(c.K_local);
        UpdateArgs aw = a;  // what k_weights and top-k stage A see
        aw.Kg = c.K_local;
        aw.Jall = (const float*)h->buf[M3_BUF_TRAJ_COST];
        aw.w = a.w + c.k_offset;
        aw.kbase = c.k_offset;
        if (update_small_applies(aw)) {
            launch_update_small(aw, h->stream);   // one launch: softmin + sums of the local shard
        } else {
            launch_weights(aw, h->stream);
            launch_wsum(a, h->stream);
        }
        HIPCHK(h, hipGetLastError());
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
        return M3_OK;
    }
    if (update_small_applies(a)) {  // K <= 4096: one launch (update.hip, k_update_small)
        launch_update_small(a, h->stream);
        HIPCHK(h, hipGetLastError());
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
        return M3_OK;
    }
    if (fuse && h->lflag && c.multi_modal && !c.mode_simple && !h->five_launches) {   // (m3_set_update_launches(h, 5): the round-3 path)
        // three launches (update.hip: k_ladder_search): the minima are the rows the rollout's workgroups left behind
        // when the costs are this command's rollout's, k_mins' rows otherwise (costs written by the caller)
        if (h->use_wave_min) { a.part_min = h->wave_min; a.n_mins = h->wave_min_rows; }
        else launch_mins(a, h->stream);
        a.epoch = ++h->lad_epoch;
        launch_ladder_search(a, h->stream);
        launch_fused_large(a, h->stream);
        HIPCHK(h, hipGetLastError());
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
        return M3_OK;
    }
    // (minima + beta ladder for the multi-modal search) -> weights (+ top-k stage A as extra
    // workgroups) -> weighted sums (+ top-k stage B as an extra workgroup when K > 4096)
    if (c.multi_modal && !c.mode_simple) {
        launch_mins(a, h->stream);
        launch_ladder(a, h->stream);
    }
    launch_weights(a, h->stream);
    launch_wsum(a, h->stream);
    HIPCHK(h, hipGetLastError());
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
    return M3_OK;
}

extern "C" int m3_update(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    return update_impl(h, false);
}

// one-collective multi-modal sharding, phase after the all-gather: the whole update on all K_global
// samples -- the unsharded kernels, with the weighted sums re-generating the actions (k_wsum<REGEN>)
static int regen_finalize(m3_handle* h) {
    const m3_config& c = h->cfg;
    UpdateArgs a;
    fill_update_args(h, a);
    a.regen = 1;
    if (h->regen_fast) {
        // shard_mix = 2: k_search mixes the shards' ladder tables, k_regen_part / k_regen_done do the rest
        if ((long long)c.T * c.nu > 2048) return fail(h, M3_ERR_UNSUPPORTED, "m3_finalize: shard_mix = 2 needs T * nu <= 2048");
        a.fast = 1;
        a.Kl = c.K_global; a.k0 = 0;
        a.n_chunk = wsum_chunks(c.K_global);
        a.top_dst = a.top_trajs;
        a.fuse_finalize = 1;
        launch_regen_fast(a, h->stream);
        HIPCHK(h, hipGetLastError());
        return M3_OK;
    }
    a.Jout = (float*)h->buf[M3_BUF_TRAJ_COST_ALL];   // k_mins compacts the records' costs into it
    a.Kl = c.K_global; a.k0 = 0;           // the launches cover every sample
    a.n_chunk = wsum_chunks(c.K_global);
    a.top_dst = a.top_trajs;               // rows come from the gathered records (topk_finish)
    const bool fuse = (long long)c.T * c.nu <= 2048;
    a.fuse_finalize = fuse ? 1 : 0;
    launch_mins(a, h->stream);
    launch_ladder(a, h->stream);
    launch_weights(a, h->stream);
    launch_wsum(a, h->stream);
    if (!fuse) launch_finalize(a, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

// shard_mix = 3, between the two exchanges (update.hip: k_p3_done's comment): searches on the mixed tables, then
// weights and weighted sums of this rank's OWN samples into its second record
extern "C" int m3_update_b(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->p3) return fail(h, M3_ERR_STATE, "m3_update_b: only for cfg.shard_mix = 3 handles");
    const m3_config& c = h->cfg;
    float* rec = (float*)h->buf[M3_BUF_RECORD];
    float* recb = (float*)h->buf[M3_BUF_RECORD_B];
    {   // searches (all ranks' tables; the gathered costs only if one leaves its ladder) + the global top-k
        UpdateArgs a;
        fill_update_args(h, a);
        a.regen = 1; a.fast = 1;
        a.Kl = c.K_global; a.k0 = 0;
        a.top_dst = a.top_trajs;
        launch_p3_search(a, h->stream);
    }
    {   // weights of the local samples (global minima / eta / beta from the search)
        UpdateArgs a;
        fill_update_args(h, a);
        a.Kg = c.K_local;
        a.Jall = rec;
        a.kbase = c.k_offset;
        a.w = (float*)h->buf[M3_BUF_WEIGHTS] + c.k_offset;
        a.w1 = (float*)h->buf[M3_BUF_WEIGHTS_1] + c.k_offset;
        a.rec_b = recb;
        launch_p3_local_weights(a, h->stream);
    }
    {   // their weighted action sums + the rows of the local best samples
        UpdateArgs a;
        fill_update_args(h, a);
        a.reduce = recb + RECB_HDR;
        a.n_cand = 0;
        a.fuse_finalize = 0;
        launch_wsum(a, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

// MPPIConfig.update_cov: the covariance update that follows the mean update (mppi.py:508-516)
static int after_finalize(m3_handle* h) {
    if (!h->cov_active) return M3_OK;
    const m3_config& c = h->cfg;
    launch_cov_update((const float*)h->buf[M3_BUF_ACTIONS], (const float*)h->buf[M3_BUF_WEIGHTS],
                      (const float*)h->buf[M3_BUF_MEAN], h->wpart, (float*)h->buf[M3_BUF_COV], c.K_local, c.T, c.nu,
                      h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

// ---- device-side exchange of the records (p2p.hip) -------------------------------------------------------------
static int p2p_ranks(const m3_handle* h) { return h->cfg.K_global / h->cfg.K_local; }
static int p2p_rank(const m3_handle* h) { return h->cfg.k_offset / h->cfg.K_local; }
// the device a pointer lives on (-1: unknown)
static int ptr_device(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return at.device;
}
static int p2p_alloc(m3_handle* h) {
    if (h->xb) return M3_OK;
    const m3_config& c = h->cfg;
    // the block belongs on the HANDLE's device, whatever device is current on the calling thread (a process that
    // drives several GPUs, torch's current device): set it for the allocation and put the caller's back
    struct DeviceGuard {
        int prev = -1;
        explicit DeviceGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(d); }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard(c.device);
    if (!(c.shard_mix && c.K_local != c.K_global) || !h->buf[M3_BUF_RECORD])
        return fail(h, M3_ERR_STATE, "m3_p2p: the handle has no record to exchange (needs cfg.shard_mix on a sharded handle)");
    const size_t rl = (size_t)m3_record_len(h), rlb = (size_t)m3_record_b_len(h);
    h->xb_bytes = P2P_HDR_BYTES + 2 * (size_t)p2p_ranks(h) * (((rl + 3) & ~(size_t)3) + ((rlb + 3) & ~(size_t)3)) * sizeof(float);
    // uncached: neither the peers' stores nor the owner's loads may be served from a stale L2 line
    // (m3_p2p_set_memory_kind: start further down the fallback chain -- the tests of the fenced paths)
    const int first = h->xb_first_kind;
    if (first <= 1 && hipExtMallocWithFlags(&h->xb, h->xb_bytes, hipDeviceMallocUncached) == hipSuccess) h->xb_kind = 1;
    else if ((void)hipGetLastError(), first <= 2 && hipExtMallocWithFlags(&h->xb, h->xb_bytes, hipDeviceMallocFinegrained) == hipSuccess) h->xb_kind = 2;
    else if ((void)hipGetLastError(), hipMalloc(&h->xb, h->xb_bytes) == hipSuccess) h->xb_kind = 3;
    else { h->xb = nullptr; return fail(h, M3_ERR_HIP, "m3_p2p: allocation of the exchange block failed"); }
    HIPCHK(h, hipMemset(h->xb, 0, h->xb_bytes));
    HIPCHK(h, hipDeviceSynchronize());
    const int on = ptr_device(h->xb);
    if (on >= 0 && on != c.device) return fail(h, M3_ERR_HIP, "m3_p2p: the exchange block did not land on the handle's device");
    return M3_OK;
}

extern "C" int m3_p2p_export(m3_handle* h, m3_ipc_handle* out) {
    if (!h || !out) return M3_ERR_BAD_ARG;
    const int rc = p2p_alloc(h);
    if (rc != M3_OK) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(m3_ipc_handle), "m3_ipc_handle too small");
    hipIpcMemHandle_t ih;
    HIPCHK(h, hipIpcGetMemHandle(&ih, h->xb));
    std::memset(out, 0, sizeof(*out));
    std::memcpy(out->bytes, &ih, sizeof(ih));
    return M3_OK;
}

static int p2p_finish_connect(m3_handle* h) {
    for (int p = 0; p < p2p_ranks(h); ++p)
        if (!h->peer_base[p]) return fail(h, M3_ERR_STATE, "m3_p2p_connect: a peer's block is missing");
    // (re-)arm: own flags, error word and sequence numbers start from zero.  Collective by nature -- no peer may put
    // between this and the barrier every caller of connect holds before its first exchange (distributed.attach_p2p).
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemset(h->xb, 0, P2P_HDR_BYTES));
    HIPCHK(h, hipDeviceSynchronize());
    h->p2p_ready = true;
    h->p2p_seq[0] = h->p2p_seq[1] = 0;
    h->records_src = nullptr; h->recb_src = nullptr;
    return M3_OK;
}

extern "C" int m3_p2p_clear_error(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->xb || !h->p2p_ready) return fail(h, M3_ERR_STATE, "m3_p2p_clear_error: no connected exchange block");
    return p2p_finish_connect(h);
}

extern "C" int m3_p2p_detach(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    // the blocks stay mapped (m3_p2p_connect* re-arms them); the finalize kernels no longer read the error word, so a
    // handle whose exchange gave up hands out plans again once its records come from another transport
    if (h->xb) HIPCHK(h, hipStreamSynchronize(h->stream));
    h->p2p_ready = false;
    h->records_src = nullptr; h->recb_src = nullptr;
    return M3_OK;
}

extern "C" int m3_p2p_set_memory_kind(m3_handle* h, int first_kind) {
    if (!h) return M3_ERR_BAD_ARG;
    if (first_kind < 1 || first_kind > 3) return fail(h, M3_ERR_BAD_ARG, "m3_p2p_set_memory_kind: 1 uncached, 2 fine-grained, 3 plain");
    if (h->xb) return fail(h, M3_ERR_STATE, "m3_p2p_set_memory_kind: the exchange block exists already (call before m3_p2p_export / connect)");
    h->xb_first_kind = first_kind;
    return M3_OK;
}

extern "C" int m3_p2p_connect(m3_handle* h, const m3_ipc_handle* all, int n) {
    if (!h || !all) return M3_ERR_BAD_ARG;
    int rc = p2p_alloc(h);
    if (rc != M3_OK) return rc;
    if (n != p2p_ranks(h)) return fail(h, M3_ERR_SHAPE, "m3_p2p_connect: need one handle per rank");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    for (int p = 0; p < n; ++p) {
        if (p == p2p_rank(h)) { h->peer_base[p] = h->xb; continue; }
        if (h->peer_ipc[p] && h->peer_base[p]) continue;
        hipIpcMemHandle_t ih;
        std::memcpy(&ih, all[p].bytes, sizeof(ih));
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, ih, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, M3_ERR_HIP, "m3_p2p_connect: hipIpcOpenMemHandle failed (peer block of another process)");
        }
        h->peer_base[p] = ptr;
        h->peer_ipc[p] = true;
    }
    return p2p_finish_connect(h);
}

extern "C" int m3_p2p_connect_local(m3_handle* h, m3_handle* const* peers, int n) {
    if (!h || !peers) return M3_ERR_BAD_ARG;
    int rc = p2p_alloc(h);
    if (rc != M3_OK) return rc;
    if (n != p2p_ranks(h)) return fail(h, M3_ERR_SHAPE, "m3_p2p_connect_local: need one handle per rank");
    for (int p = 0; p < n; ++p) {
        m3_handle* q = peers[p];
        if (!q || p2p_rank(q) != p || m3_record_len(q) != m3_record_len(h) || p2p_ranks(q) != n ||
            m3_record_b_len(q) != m3_record_b_len(h) || q->cfg.shard_mix != h->cfg.shard_mix)
            return fail(h, M3_ERR_SHAPE, "m3_p2p_connect_local: peers[p] must be the handle of rank p of the same sharding and protocol");
        rc = p2p_alloc(q);
        if (rc != M3_OK) return fail(h, rc, "m3_p2p_connect_local: a peer could not allocate its block");
        const int on = ptr_device(q->xb);
        if (on >= 0 && on != q->cfg.device) return fail(h, M3_ERR_HIP, "m3_p2p_connect_local: a peer's block is not on that peer's device");
        if (q->cfg.device != h->cfg.device) {
            int can = 0, prev = -1;
            (void)hipDeviceCanAccessPeer(&can, h->cfg.device, q->cfg.device);
            if (!can) return fail(h, M3_ERR_UNSUPPORTED, "m3_p2p_connect_local: no peer access between the two devices");
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
            (void)hipSetDevice(h->cfg.device);
            const hipError_t e = hipDeviceEnablePeerAccess(q->cfg.device, 0);
            if (prev >= 0) (void)hipSetDevice(prev);       // (the caller's current device is the caller's)
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return fail(h, M3_ERR_HIP, "m3_p2p_connect_local: hipDeviceEnablePeerAccess"); }
            (void)hipGetLastError();
        }
        h->peer_base[p] = q->xb;
    }
    return p2p_finish_connect(h);
}

// channel 0: the records (M3_BUF_RECORD); channel 1: the second records of shard_mix = 3 (M3_BUF_RECORD_B)
static void p2p_args(m3_handle* h, P2PArgs& a, int ch) {
    std::memset(&a, 0, sizeof(a));
    const int lenA = m3_record_len(h), strideA = (lenA + 3) & ~3;
    a.rec = (const float*)h->buf[ch == 0 ? M3_BUF_RECORD : M3_BUF_RECORD_B];
    a.rec_len = ch == 0 ? lenA : m3_record_b_len(h);
    a.rec_stride = (a.rec_len + 3) & ~3;
    a.n_ranks = p2p_ranks(h);
    a.rank = p2p_rank(h);
    a.seq = h->p2p_seq[ch];
    a.slot = a.seq & 1;
    a.timeout_ticks = 100000ull * (unsigned long long)(a.seq <= 1 ? h->p2p_first_ms : h->p2p_ms);   // 100 MHz wall clock
    a.plain_memory = h->xb_kind != 1;     // only the uncached block skips both GPUs' L2 for certain: the fine-grained fallback gets the fences too
    a.err = (int*)((char*)h->xb + 2 * MIX_MAX_RANKS * sizeof(int));
    for (int p = 0; p < a.n_ranks; ++p) {
        a.peer_flags[p] = (int*)((char*)h->peer_base[p] + (ch == 0 ? 0 : 512));
        a.peer_data[p] = (float*)((char*)h->peer_base[p] + P2P_HDR_BYTES) + (ch == 0 ? 0 : 2 * (size_t)a.n_ranks * strideA);
    }
}

extern "C" int m3_p2p_put_ch(m3_handle* h, int ch) {
    if (!h) return M3_ERR_BAD_ARG;
    if (ch != 0 && !(ch == 1 && h->p3)) return fail(h, M3_ERR_BAD_ARG, "m3_p2p_put: channel 1 exists on shard_mix = 3 handles only");
    if (!h->p2p_ready) return fail(h, M3_ERR_STATE, "m3_p2p_put: m3_p2p_connect first");
    h->p2p_seq[ch] += 1;
    P2PArgs a;
    p2p_args(h, a, ch);
    launch_p2p_put(a, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_p2p_wait_ch(m3_handle* h, int ch) {
    if (!h) return M3_ERR_BAD_ARG;
    if (ch != 0 && !(ch == 1 && h->p3)) return fail(h, M3_ERR_BAD_ARG, "m3_p2p_wait: channel 1 exists on shard_mix = 3 handles only");
    if (!h->p2p_ready || h->p2p_seq[ch] == 0) return fail(h, M3_ERR_STATE, "m3_p2p_wait: no exchange in flight (m3_p2p_put first)");
    P2PArgs a;
    p2p_args(h, a, ch);
    launch_p2p_wait(a, h->stream);
    HIPCHK(h, hipGetLastError());
    const float* src = a.peer_data[a.rank] + (size_t)a.slot * a.n_ranks * a.rec_stride;
    if (ch == 0) { h->records_src = src; h->records_stride = a.rec_stride; }
    else { h->recb_src = src; h->recb_stride = a.rec_stride; }
    return M3_OK;
}

extern "C" int m3_p2p_put(m3_handle* h) { return m3_p2p_put_ch(h, 0); }
extern "C" int m3_p2p_wait(m3_handle* h) { return m3_p2p_wait_ch(h, 0); }
static int p2p_exchange_ch(m3_handle* h, int ch) {   // put + wait in one launch
    if (!h) return M3_ERR_BAD_ARG;
    if (ch != 0 && !(ch == 1 && h->p3)) return fail(h, M3_ERR_BAD_ARG, "m3_p2p_exchange: channel 1 exists on shard_mix = 3 handles only");
    if (!h->p2p_ready) return fail(h, M3_ERR_STATE, "m3_p2p_exchange: m3_p2p_connect first");
    h->p2p_seq[ch] += 1;
    P2PArgs a;
    p2p_args(h, a, ch);
    launch_p2p_exchange(a, h->stream);
    HIPCHK(h, hipGetLastError());
    const float* src = a.peer_data[a.rank] + (size_t)a.slot * a.n_ranks * a.rec_stride;
    if (ch == 0) { h->records_src = src; h->records_stride = a.rec_stride; }
    else { h->recb_src = src; h->recb_stride = a.rec_stride; }
    return M3_OK;
}
extern "C" int m3_p2p_exchange(m3_handle* h) { return p2p_exchange_ch(h, 0); }
extern "C" int m3_p2p_exchange_b(m3_handle* h) { return p2p_exchange_ch(h, 1); }

extern "C" int m3_p2p_set_timeout_ms(m3_handle* h, int first_ms, int ms) {
    if (!h) return M3_ERR_BAD_ARG;
    if (first_ms < 1 || first_ms > 600000 || ms < 1 || ms > 600000)
        return fail(h, M3_ERR_BAD_ARG, "m3_p2p_set_timeout_ms: 1 .. 600000 ms each");
    h->p2p_first_ms = first_ms;
    h->p2p_ms = ms;
    return M3_OK;
}

extern "C" int m3_p2p_status(m3_handle* h, int* missing_rank, int* memory_kind) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->xb) return fail(h, M3_ERR_STATE, "m3_p2p_status: no exchange block");
    int e = 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(&e, (char*)h->xb + 2 * MIX_MAX_RANKS * sizeof(int), sizeof(int), hipMemcpyDeviceToHost));
    if (missing_rank) *missing_rank = e - 1;      // -1: every exchange so far was complete
    if (memory_kind) *memory_kind = h->xb_kind;
    return M3_OK;
}

extern "C" int m3_finalize(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    struct Consume { m3_handle* h; ~Consume() { h->records_src = nullptr; h->recb_src = nullptr; } } consume{h};   // one exchange, one finalize
    if (h->p3) {
        UpdateArgs a;
        fill_update_args(h, a);
        a.Kl = h->cfg.K_global; a.k0 = 0;      // (finalize_body: the top trajectories are already in place)
        launch_p3_done(a, h->stream);
        HIPCHK(h, hipGetLastError());
        h->recb_src = nullptr;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
        h->calls += 1;
        return M3_OK;
    }
    if (h->regen) {
        const int rc = regen_finalize(h);
        if (rc != M3_OK) return rc;
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
        h->calls += 1;
        return M3_OK;
    }
    UpdateArgs a;
    fill_update_args(h, a);
    if (mix_mode(h)) launch_mix(a, h->stream);  // records -> the REDUCE buffer an all-reduce would hold, + finalize
    else launch_finalize(a, h->stream);
    HIPCHK(h, hipGetLastError());
    const int rc = after_finalize(h);
    if (rc != M3_OK) return rc;
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    h->calls += 1;
    return M3_OK;
}

extern "C" int m3_update_finalize(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->cfg.K_local != h->cfg.K_global)
        return fail(h, M3_ERR_STATE, "m3_update_finalize: sharded handle (m3_update, collectives, m3_finalize)");
    if (!can_fuse_finalize(h)) {
        const int rc = m3_update(h);
        return rc != M3_OK ? rc : m3_finalize(h);
    }
    int rc = update_impl(h, true);
    if (rc == M3_OK) rc = after_finalize(h);
    if (rc != M3_OK) return rc;
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
    h->calls += 1;
    return M3_OK;
}

extern "C" int m3_command(m3_handle* h, float* action_host) {
    if (!h) return M3_ERR_BAD_ARG;
    // a sharded handle needs the collective(s) between the phases: running update + finalize on the
    // local costs alone would return a plan built from zero / stale remote slices without an error
    if (h->cfg.K_local != h->cfg.K_global)
        return fail(h, M3_ERR_STATE, "m3_command: sharded handle (K_local != K_global): call m3_rollout, m3_update, "
                                     "m3_finalize with the collective(s) in between (include/m3p2i_hip.h)");
    int rc = m3_rollout(h);
    if (rc != M3_OK) return rc;
    h->use_wave_min = h->wave_min != nullptr && h->wave_min_rows > 0;   // the costs the update reads are this rollout's
    rc = m3_update_finalize(h);   // weights -> sums + (last workgroup) mean update / filter
    h->use_wave_min = false;
    if (rc != M3_OK) return rc;
    if (action_host) {
        const m3_config& c = h->cfg;
        const int rows = c.mode_simple ? c.u_per_command : c.T;
        HIPCHK(h, hipMemcpyAsync(action_host, h->action_out ? h->action_out : (float*)h->buf[M3_BUF_ACTION_OUT], (size_t)rows * c.nu * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return M3_OK;
}

static int ensure_sim(m3_handle* h);
extern "C" int m3_record_b_len(const m3_handle* h) {
    if (!h || !h->p3) return 0;
    return recb_length(h->cfg.T, h->cfg.nu);
}

extern "C" int m3_record_len(const m3_handle* h) {
    if (!h) return 0;
    return h->regen ? regen_record_length(h->cfg.K_local, h->cfg.T) : record_length(h->cfg.T, h->cfg.nu);
}

extern "C" int m3_get_buffer(m3_handle* h, int which, void** p, long long* nbytes) {
    if (!h || !p) return M3_ERR_BAD_ARG;
    if (which < 0 || which >= M3_BUF_COUNT) return fail(h, M3_ERR_BAD_ARG, "m3_get_buffer: unknown buffer id");
    if (which == M3_BUF_SIM_WORLD) {
        int rc = ensure_sim(h);
        if (rc != M3_OK) return rc;
        *p = h->sim_world;
        if (nbytes) *nbytes = (long long)(h->cfg.env_type == M3_ENV_POINT ? NW : NWP) * h->cfg.K_local * sizeof(float);
        return M3_OK;
    }
    if (which == M3_BUF_TRAJ_COST_ALL && !h->cfg.sim_only && h->cfg.K_local == h->cfg.K_global)
        which = M3_BUF_TRAJ_COST;  // unsharded: the local costs are the global costs
    if (!h->buf[which]) return fail(h, M3_ERR_STATE, "m3_get_buffer: buffer not allocated for this handle");
    *p = h->buf[which];
    if (nbytes) *nbytes = h->nbytes[which];
    return M3_OK;
}

extern "C" int m3_reduce_len(const m3_handle* h) {
    return h ? reduce_length(h->cfg.T, h->cfg.nu) : M3_ERR_BAD_ARG;
}

extern "C" int m3_get_info(m3_handle* h, m3_info* out) {
    if (!h || !out) return M3_ERR_BAD_ARG;
    HIPCHK(h, hipMemcpyAsync(out, h->buf[M3_BUF_INFO], sizeof(m3_info), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out->calls = (int)h->calls;
    return M3_OK;
}

extern "C" int m3_get_timing(m3_handle* h, m3_timing* out) {
    if (!h || !out) return M3_ERR_BAD_ARG;
    if (!h->timing) return fail(h, M3_ERR_STATE, "m3_get_timing: timing not enabled");
    HIPCHK(h, hipEventSynchronize(h->ev[3]));
    HIPCHK(h, hipEventElapsedTime(&out->rollout_ms, h->ev[0], h->ev[1]));
    HIPCHK(h, hipEventElapsedTime(&out->update_ms, h->ev[1], h->ev[2]));
    HIPCHK(h, hipEventElapsedTime(&out->finalize_ms, h->ev[2], h->ev[3]));
    HIPCHK(h, hipEventElapsedTime(&out->total_ms, h->ev[0], h->ev[3]));
    return M3_OK;
}

// ---------------------------------- step mode ------------------------------------------
static int ensure_sim(m3_handle* h) {
    const m3_config& c = h->cfg;
    if (!h->sim_world) {
        const size_t nw = (c.env_type == M3_ENV_POINT) ? NW : NWP;
        HIPCHK(h, hipMalloc((void**)&h->sim_world, nw * c.K_local * sizeof(float)));
        HIPCHK(h, hipMemsetAsync(h->sim_world, 0, nw * c.K_local * sizeof(float), h->stream));
        HIPCHK(h, hipMalloc((void**)&h->sim_u, (size_t)c.K_local * c.nu * sizeof(float)));
        HIPCHK(h, hipMemsetAsync(h->sim_u, 0, (size_t)c.K_local * c.nu * sizeof(float), h->stream));
    }
    return M3_OK;
}

extern "C" int m3_sim_bind_views(m3_handle* h, float* dof, float* root, float* rb, float* ncf, int n_actors, int n_bodies) {
    if (!h) return M3_ERR_BAD_ARG;
    const bool point = h->cfg.env_type == M3_ENV_POINT;
    // actor tables of this build (DESIGN.md "Scene tables"), robot last (skill_utils.py:89-90):
    //   point_env: walls 0-3, obs 4, dyn-obs 5, box 6, goal 7, yaxis 8, xaxis 9, robot 10;
    //              bodies = actors 0..9 + plane 10, link_x 11, link_y 12
    //   panda_env: table 0, table_stand 1, shelf_stand 2, dyn-obs 3, cubeA 4, cubeB 5, panda 6;
    //              bodies = actors 0..5 + panda_link0..7, hand, leftfinger, rightfinger (6..16)
    if (point && (n_actors != 11 || n_bodies != 13)) return fail(h, M3_ERR_SHAPE, "m3_sim_bind_views: point_env has 11 actors / 13 bodies");
    if (!point && (n_actors != 7 || n_bodies != 17)) return fail(h, M3_ERR_SHAPE, "m3_sim_bind_views: panda_env has 7 actors / 17 bodies");
    int rc = ensure_sim(h);
    if (rc != M3_OK) return rc;
    SimViews& v = h->views;
    v.dof_state = dof; v.root_state = root; v.rigid_body_state = rb; v.net_contact_force = ncf;
    v.n_actors = n_actors; v.n_bodies = n_bodies;
    if (point) {
        v.box_actor = 6; v.dyn_actor = 5; v.robot_actor = 10;
        v.box_body = 6; v.dyn_body = 5; v.robot_body = 12;
        v.table_body = v.shelf_body = 0;
    } else {
        v.box_actor = 4; v.dyn_actor = 5; v.robot_actor = 6;  // cubeA, cubeB, panda
        v.box_body = 4; v.dyn_body = 5; v.robot_body = 6;
        v.table_body = 0; v.shelf_body = 2;
        v.obs_actor = 3; v.obs_body = 3;
    }
    h->views_bound = true;
    return M3_OK;
}

extern "C" int m3_sim_pull_state(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->views_bound || !h->views.dof_state || !h->views.root_state) return fail(h, M3_ERR_STATE, "m3_sim_pull_state: views not bound");
    if (h->cfg.env_type == M3_ENV_POINT) launch_sim_pull(h->views, h->sim_world, h->cfg.K_local, h->stream);
    else launch_psim_pull(h->pscene, h->views, h->sim_world, h->cfg.K_local, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_sim_shift_actor(m3_handle* h, int actor, float dx, float dy, float dz) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->views_bound) return fail(h, M3_ERR_STATE, "m3_sim_shift_actor: views not bound");
    if (h->cfg.env_type != M3_ENV_POINT) return fail(h, M3_ERR_UNSUPPORTED, "m3_sim_shift_actor: point_env (its dyn-obs walks; the panda_env's offset is zero)");
    if (actor < 0 || actor >= h->views.n_actors) return fail(h, M3_ERR_SHAPE, "m3_sim_shift_actor: actor index out of range");
    launch_sim_shift_pull(h->views, h->sim_world, h->cfg.K_local, actor, dx, dy, dz, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_sim_push_state(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    if (!h->views_bound) return fail(h, M3_ERR_STATE, "m3_sim_push_state: views not bound");
    if (h->cfg.env_type == M3_ENV_POINT) launch_sim_push(h->views, h->sim_world, h->cfg.K_local, h->stream);
    else launch_psim_push(h->pscene, h->views, h->sim_world, h->cfg.K_local, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_sim_set_velocity_target(m3_handle* h, const float* u) {
    if (!h || !u) return M3_ERR_BAD_ARG;
    int rc = ensure_sim(h);
    if (rc != M3_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h->sim_u, u, (size_t)h->cfg.K_local * h->cfg.nu * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return M3_OK;
}

extern "C" int m3_sim_apply_body_forces(m3_handle* h, const float* f) {
    if (!h || !f) return M3_ERR_BAD_ARG;
    if (!h->views_bound) return fail(h, M3_ERR_STATE, "m3_sim_apply_body_forces: views not bound");
    if (h->cfg.env_type != M3_ENV_POINT)
        return fail(h, M3_ERR_UNSUPPORTED, "m3_sim_apply_body_forces: the panda_env path applies no body forces (only get_pull_cost does, point_env)");
    launch_sim_forces(h->views, h->sim_world, f, h->cfg.K_local, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

static int sim_step_impl(m3_handle* h, const float* u) {
    if (!h->views_bound) return fail(h, M3_ERR_STATE, "m3_sim_step: views not bound");
    if (h->cfg.env_type == M3_ENV_POINT) {
        launch_sim_step(h->scene, h->views, h->sim_world, u, h->sim_u, h->cfg.K_local, h->stream);
    } else {
        launch_psim_step(h->pscene, h->views, h->sim_world, u, h->sim_u, h->cfg.K_local, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_sim_step(m3_handle* h) {
    if (!h) return M3_ERR_BAD_ARG;
    return sim_step_impl(h, h->sim_u);
}

extern "C" int m3_sim_step_with_target(m3_handle* h, const float* u) {
    if (!h || !u) return M3_ERR_BAD_ARG;
    return sim_step_impl(h, u);
}

static int suction_impl(m3_handle* h, float kp, const float* action, int apply, float* forces, int* flags,
                        const int* gate) {
    if (!h) return M3_ERR_BAD_ARG;
    if (h->cfg.env_type != M3_ENV_POINT) return fail(h, M3_ERR_UNSUPPORTED, "suction is a point_env skill (skill_utils.py:36-94)");
    if (!h->views_bound || !h->sim_world) return fail(h, M3_ERR_STATE, "suction: views not bound");
    const float thresh = (h->cfg.K_local == 1) ? 1.5f : 1.8f;   // skill_utils.py:75-82 (sim.num_envs == 1: real world)
    launch_sim_suction(h->views, h->sim_world, h->cfg.K_local, kp, thresh, 0.6f /* skill_utils.py:56 */, action, apply,
                       forces, flags, gate, h->stream);
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}

extern "C" int m3_sim_suction_forces(m3_handle* h, float kp_suction, float* forces_dev) {
    if (!forces_dev) return fail(h, M3_ERR_BAD_ARG, "m3_sim_suction_forces: null argument");
    return suction_impl(h, kp_suction, nullptr, 0, forces_dev, nullptr, nullptr);
}

extern "C" int m3_sim_check_and_apply_suction(m3_handle* h, const float* action_dev, float kp_suction, int apply,
                                              int* applied_dev, const int* enabled_dev) {
    if (!action_dev) return fail(h, M3_ERR_BAD_ARG, "m3_sim_check_and_apply_suction: null argument");
    return suction_impl(h, kp_suction, action_dev, apply, nullptr, applied_dev, enabled_dev);
}

extern "C" int m3_cost(m3_handle* h, float* cost) {
    if (!h || !cost) return M3_ERR_BAD_ARG;
    if (!h->sim_world) return fail(h, M3_ERR_STATE, "m3_cost: step-mode state not initialised");
    if (h->cfg.env_type == M3_ENV_POINT) {
        CostParams cp;
        fill_cost_params(h, cp);
        launch_sim_cost(cp, h->sim_world, h->cfg.K_local, h->cfg.k_offset, cost, h->stream);
    } else {
        PandaCostParams cp;
        fill_panda_cost_params(h, cp);
        // quirk Q8 exactly where m3_rollout applies it (its shadow lanes): reach on an unsharded handle
        const bool env0_cube = cp.task == 4 && h->cfg.k_offset == 0 && h->cfg.K_local == h->cfg.K_global && h->cfg.K_global >= 2;
        launch_psim_cost(h->pscene, cp, h->sim_world, h->cfg.K_local, h->cfg.k_offset, env0_cube, cost, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    return M3_OK;
}
