/*
 * mppi_core.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar restatement of the reference's MPPI / M3P2I planner arithmetic and point-env task
 * costs.  Every function cites the reference lines it follows (relative to
 * /root/reference).  Pinned against tests/golden/ npz fixtures, which were produced by importing
 * the reference's own Python (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "m3_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

void m3o_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int m3o_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* batch step of n independent worlds (each 31 floats = m3o_point_world), u[n,2] */
void m3o_point_step_batch(const m3o_point_scene* sc, float* worlds, int n, const float* u) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        m3o_point_step(sc, (m3o_point_world*)(worlds + (size_t)i * 31), u + (size_t)i * 2);
}
/* batch cost of n worlds (global sample index = k0 + i) */
void m3o_point_cost_batch(const m3o_cfg* cfg, float* worlds, int n, int k0, float* c) {
    for (int i = 0; i < n; ++i)
        c[i] = m3o_point_cost(cfg, (m3o_point_world*)(worlds + (size_t)i * 31), k0 + i);
}

/* ------------------------------------------------------------------------------------
 * A4 action assembly: mppi.py:381-416 (+ scale_ctrl clamp, mppi_utils.py:29-37)
 *   delta[-1] = 0                                  mppi.py:392
 *   scaled = delta * sqrt(diag Sigma)              mppi.py:394
 *   act = mean (+ per-mode means for the halves)   mppi.py:397-402
 *   clamp                                          mppi.py:405
 *   rows 0 / K/2 <- best_traj_1 / best_traj_2      mppi.py:407-409
 *   panda gripper dofs 7,8 <- +-1.5                mppi.py:412-416
 * ---------------------------------------------------------------------------------- */
void m3o_assemble_actions(const m3o_cfg* cfg, const float* delta, const float* mean,
                          const float* mean1, const float* mean2, const float* best1,
                          const float* best2, int k0, int k1, float* act) {
    const int K = cfg->K, T = cfg->T, nu = cfg->nu, half = K / 2;
    for (int k = k0; k < k1; ++k) {
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < nu; ++j) {
                float d = (k == K - 1) ? 0.0f : delta[((size_t)k * T + t) * nu + j];
                float sd = d * cfg->scale_tril[j];
                const float* m = mean;
                if (cfg->multi_modal) m = (k < half) ? mean1 : mean2;
                float a = m[t * nu + j] + sd;
                a = fmaxf(fminf(a, cfg->u_max[j]), cfg->u_min[j]);
                if (cfg->multi_modal) {
                    if (k == 0) a = best1[t * nu + j];
                    if (k == half) a = best2[t * nu + j];
                }
                if (cfg->env_type == 1 && j >= 7) {
                    if (cfg->gripper_cmd == 1) a = 1.5f;
                    else if (cfg->gripper_cmd == 2) a = -1.5f;
                }
                act[((size_t)(k - k0) * T + t) * nu + j] = a;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A7/A8 point-env costs: cost_functions.py:19-89,158-169; suction skill_utils.py:59-94
 * ---------------------------------------------------------------------------------- */
typedef struct { float dist_cost, cos_theta; } dist_t;

/* calculate_dist: cost_functions.py:41-50 */
static dist_t calc_dist(const m3o_point_world* w, const float* goal) {
    float r2bx = w->R.x - w->B.x, r2by = w->R.y - w->B.y;
    float b2gx = goal[0] - w->B.x, b2gy = goal[1] - w->B.y;
    float d1 = sqrtf(r2bx * r2bx + r2by * r2by);
    float d2 = sqrtf(b2gx * b2gx + b2gy * b2gy);
    dist_t o;
    o.dist_cost = d1 + d2 * 10.0f;
    o.cos_theta = (r2bx * b2gx + r2by * b2gy) / (d1 * d2);
    return o;
}

/* get_motion_cost: cost_functions.py:158-169 (point_env branch) */
static float motion_cost_point(const m3o_point_world* w) {
    float coll = fabsf(w->fc_D[0]) + fabsf(w->fc_D[1]);
    return (coll > 0.1f) ? 1000.0f : 0.0f;
}

static float clamp500(float v) { return fminf(fmaxf(v, -500.0f), 500.0f); }

float m3o_point_cost(const m3o_cfg* cfg, m3o_point_world* w, int k) {
    const int half = cfg->K / 2;
    const int task = cfg->task;
    if (task == M3O_TASK_NAVIGATION) {
        /* cost_functions.py:38 + :36 */
        float dx = w->R.x - cfg->goal[0], dy = w->R.y - cfg->goal[1];
        return sqrtf(dx * dx + dy * dy) + motion_cost_point(w);
    }
    float push = 0.0f, pull = 0.0f;
    if (task == M3O_TASK_PUSH || task == M3O_TASK_PUSH_PULL) {
        /* get_push_cost: cost_functions.py:52-60 */
        dist_t d = calc_dist(w, cfg->goal);
        float align = (d.cos_theta > 0.0f) ? d.cos_theta : 0.0f;
        push = 3.0f * d.dist_cost + 1.0f * align;
    }
    if (task == M3O_TASK_PULL || task == M3O_TASK_PUSH_PULL) {
        /* get_pull_cost: cost_functions.py:62-89 */
        dist_t d = calc_dist(w, cfg->goal);
        float pdx = w->B.x - w->R.x, pdy = w->B.y - w->R.y;
        float rdist = sqrtf(pdx * pdx + pdy * pdy);
        int toward = (w->R.vx * pdx + w->R.vy * pdy) > 0.0f;
        /* calculate_suction: skill_utils.py:59-94 */
        float mag = 1.0f / rdist;
        float ux = pdx * mag, uy = pdy * mag;
        int mask = mag > cfg->suction_thresh;
        float fbx = 0.0f, fby = 0.0f, frx = 0.0f, fry = 0.0f;
        if (mask) {
            fbx = clamp500(-cfg->kp_suction * ux);
            fby = clamp500(-cfg->kp_suction * uy);
            frx = clamp500(cfg->kp_suction * ux);
            fry = clamp500(cfg->kp_suction * uy);
        }
        if (toward) { fbx = fby = frx = fry = 0.0f; }             /* cost_functions.py:73 */
        if (cfg->multi_modal && k < half) { fbx = fby = frx = fry = 0.0f; } /* :74-75 */
        /* apply_rigid_body_force_tensors: acts during the NEXT step (:76) */
        w->fext_B[0] = fbx; w->fext_B[1] = fby;
        w->fext_R[0] = frx; w->fext_R[1] = fry;
        float align = (d.cos_theta < 0.0f) ? -d.cos_theta : 0.0f;
        float vel_cost = (toward && rdist <= 0.5f) ? 0.6f : 0.0f;
        pull = 3.0f * d.dist_cost + 3.0f * vel_cost + 7.0f * align;
    }
    const float mc = cfg->avoid_dyn_obs ? motion_cost_point(w) : 0.0f;   /* (extension, m3_oracle.h) */
    if (task == M3O_TASK_PUSH) return cfg->avoid_dyn_obs ? push + mc : push;
    if (task == M3O_TASK_PULL) return cfg->avoid_dyn_obs ? pull + mc : pull;
    if (task == M3O_TASK_PUSH_PULL) {   /* :28-29 */
        const float c = (k < half) ? push : pull;
        return cfg->avoid_dyn_obs ? c + mc : c;
    }
    return 0.0f;
}

/* ------------------------------------------------------------------------------------
 * A5 rollout loop: mppi.py:275-332 with dynamics = reactive_tamp.py:63-70
 * ---------------------------------------------------------------------------------- */
void m3o_point_rollout(const m3o_cfg* cfg, const m3o_point_scene* sc,
                       const m3o_point_world* w0, float* pend, const float* act, int k0,
                       int k1, float* states, float* actions, float* cost_h, float* J,
                       float* S) {
    const int K = cfg->K, T = cfg->T, nu = cfg->nu;
#pragma omp parallel for schedule(static)
    for (int k = k0; k < k1; ++k) {
        const int i = k - k0;
        m3o_point_world w = *w0;
        if (pend) {
            w.fext_R[0] = pend[i * 4 + 0]; w.fext_R[1] = pend[i * 4 + 1];
            w.fext_B[0] = pend[i * 4 + 2]; w.fext_B[1] = pend[i * 4 + 3];
        }
        float j = 0.0f, s = 0.0f, g = 1.0f;
        for (int t = 0; t < T; ++t) {
            float u[2];
            for (int d = 0; d < 2; ++d) {
                float a = act[((size_t)i * T + t) * nu + d];
                u[d] = cfg->u_scale * a;                                   /* mppi.py:297 */
                if (cfg->sample_null_action && k == K - 1) u[d] = 0.0f;    /* :300-302 */
            }
            m3o_point_step(sc, &w, u);                    /* reactive_tamp.py:64-65 */
            float* st = &states[((size_t)i * T + t) * 4];
            st[0] = w.R.x; st[1] = w.R.vx; st[2] = w.R.y; st[3] = w.R.vy; /* :66-69 */
            float c = m3o_point_cost(cfg, &w, k);         /* reactive_tamp.py:72-73 */
            cost_h[(size_t)i * T + t] = c;                /* mppi.py:310 */
            for (int d = 0; d < 2; ++d)                   /* mppi.py:313: the SCALED controls are what the */
                actions[((size_t)i * T + t) * nu + d] = u[d];   /* update consumes (:329-331); :353,:420 */
                                                          /* divide only the attribute the caller reads  */
            j = j + g * c;                                /* mppi_utils.py:106-113, col 0 */
            s = s + c;                                    /* mppi.py:309 */
            g = g * cfg->gamma;
        }
        J[i] = j;
        if (S) S[i] = s;
        if (pend) {
            pend[i * 4 + 0] = w.fext_R[0]; pend[i * 4 + 1] = w.fext_R[1];
            pend[i * 4 + 2] = w.fext_B[0]; pend[i * 4 + 3] = w.fext_B[1];
        }
    }
}

/* cost_to_go column 0: mppi_utils.py:106-113 (reverse cumsum of gamma^t c, /gamma^0) */
void m3o_cost_to_go0(const float* cost_h, int K, int T, float gamma, float* J) {
    float* gs = (float*)malloc(sizeof(float) * T);
    float g = 1.0f;
    for (int t = 0; t < T; ++t) { gs[t] = g; g = g * gamma; }  /* cumprod, mppi.py:182 */
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        for (int t = T - 1; t >= 0; --t) acc = acc + gs[t] * cost_h[(size_t)k * T + t];
        J[k] = acc / gs[0];
    }
    free(gs);
}

static float vmin(const float* x, int n) {
    float m = x[0];
    for (int i = 1; i < n; ++i) m = fminf(m, x[i]);
    return m;
}

/* _exp_util core: mppi.py:437-442 */
float m3o_softmin(const float* J, int n, float beta, float* w) {
    float mn = vmin(J, n);
    double eta = 0.0;
    for (int i = 0; i < n; ++i) {
        w[i] = expf((-1.0f / beta) * (J[i] - mn));
        eta += (double)w[i];
    }
    float etaf = (float)eta;
    for (int i = 0; i < n; ++i) w[i] = (1.0f / etaf) * w[i];
    return etaf;
}

/* update_infinite_beta: m3p2i.py:24-44 */
float m3o_beta_search(const float* J, int n, float beta0, float eta_u, float eta_l,
                      int max_iters, float* w, int* iters, float* beta_out) {
    float mn = vmin(J, n);
    float beta = beta0;
    float etaf = 0.0f;
    int it = 0;
    for (;;) {
        double eta = 0.0;
        for (int i = 0; i < n; ++i) {
            w[i] = expf((-1.0f / beta) * (J[i] - mn));
            eta += (double)w[i];
        }
        etaf = (float)eta;
        ++it;
        if (etaf > eta_u) beta = beta * 0.9f;
        else if (etaf < eta_l) beta = beta * 1.2f;
        else break;
        if (it >= max_iters) break;
    }
    for (int i = 0; i < n; ++i) w[i] = (1.0f / etaf) * w[i];
    if (iters) *iters = it;
    if (beta_out) *beta_out = beta;
    return etaf;
}

static int argmax_first(const float* w, int n) {
    int b = 0;
    for (int i = 1; i < n; ++i)
        if (w[i] > w[b]) b = i;
    return b;
}

/* _exp_util (mppi.py:430-456) / _multi_modal_exp_util (m3p2i.py:46-64) + argmax
 * (mppi.py:493-494, m3p2i.py:75-76) + pull preference sums (m3p2i.py:16-22) */
void m3o_update_weights(const m3o_cfg* cfg, const float* J, float* w, float* w1, float* w2,
                        m3o_update_info* info) {
    const int K = cfg->K, half = K / 2;
    if (!cfg->multi_modal) {
        info->eta = m3o_softmin(J, K, info->beta, w);
        info->iters = 1;
        if (cfg->env_type == 1) { /* mppi.py:446-454 */
            if (info->eta > 20.0f) info->beta = info->beta * 0.9f;
            else if (info->eta < 10.0f) info->beta = info->beta * 1.2f;
        }
        info->best_idx = argmax_first(w, K);
    } else {
        /* beta_1, beta_2, beta are never written back: restart at 1 (m3p2i.py:58-60) */
        info->eta_1 = m3o_beta_search(J, half, 1.0f, 10.0f, 3.0f, 1000, w1, &info->iters_1, 0);
        info->eta_2 =
            m3o_beta_search(J + half, K - half, 1.0f, 10.0f, 3.0f, 1000, w2, &info->iters_2, 0);
        info->eta = m3o_beta_search(J, K, 1.0f, 10.0f, 3.0f, 1000, w, &info->iters, 0);
        info->best_idx_1 = argmax_first(w1, half);
        info->best_idx_2 = argmax_first(w2, K - half);
        info->best_idx = argmax_first(w, K);
    }
    double a = 0.0, b = 0.0;
    for (int k = 0; k < half; ++k) a += (double)w[k];
    for (int k = half; k < K; ++k) b += (double)w[k];
    info->wsum_push = (float)a;
    info->wsum_pull = (float)b;
}

/* weighted sums over a shard: mppi.py:497-498, m3p2i.py:80-83 */
void m3o_partial_sums(const m3o_cfg* cfg, const float* w, const float* w1, const float* w2,
                      const float* actions, int k0, int k1, float* psum, float* psum1,
                      float* psum2) {
    const int K = cfg->K, T = cfg->T, nu = cfg->nu, half = K / 2;
    const int n = T * nu;
    double* acc = (double*)calloc((size_t)3 * n, sizeof(double));
    for (int k = k0; k < k1; ++k) {
        const float* a = &actions[(size_t)(k - k0) * n];
        for (int i = 0; i < n; ++i) acc[i] += (double)(w[k] * a[i]);
        if (cfg->multi_modal) {
            if (k < half) for (int i = 0; i < n; ++i) acc[n + i] += (double)(w1[k] * a[i]);
            else for (int i = 0; i < n; ++i) acc[2 * n + i] += (double)(w2[k - half] * a[i]);
        }
    }
    for (int i = 0; i < n; ++i) {
        psum[i] = (float)acc[i];
        if (psum1) psum1[i] = (float)acc[n + i];
        if (psum2) psum2[i] = (float)acc[2 * n + i];
    }
    free(acc);
}

/* mppi.py:502-503 */
void m3o_mean_update(const m3o_cfg* cfg, float* mean, const float* sum) {
    const int n = cfg->T * cfg->nu;
    for (int i = 0; i < n; ++i)
        mean[i] = (1.0f - cfg->step_size_mean) * mean[i] + cfg->step_size_mean * sum[i];
}

/* torch.topk(weights, 20): mppi.py:248 */
void m3o_topk(const float* w, int K, int n, int* idx, float* val) {
    char* used = (char*)calloc((size_t)K, 1);
    for (int r = 0; r < n; ++r) {
        int b = -1;
        for (int k = 0; k < K; ++k) {
            if (used[k]) continue;
            if (b < 0 || w[k] > w[b]) b = k;
        }
        used[b] = 1;
        idx[r] = b;
        if (val) val[r] = w[b];
    }
    free(used);
}

/* Savitzky-Golay(9, 2, mode='interp') along T: mppi.py:257-263.  Interior rows use the
 * symmetric smoothing kernel; the first/last 4 rows evaluate the quadratic least-squares
 * fit of the first/last 9 samples (scipy 'interp' edge rule).  Coefficients are derived
 * here in double from the normal equations of the fit on x = -4..4. */
void m3o_savgol9(const float* in, int T, int nu, float* out) {
    /* quadratic LSQ on 9 equispaced points x=0..8: value at position p is sum_i C[p][i] y_i */
    double C[9][9];
    {
        /* basis 1, x, x^2 with x centred: x = i-4 */
        double S0 = 9.0, S2 = 60.0, S4 = 708.0; /* sum x^0, x^2, x^4 over -4..4 */
        double det = S0 * S4 - S2 * S2;
        for (int p = 0; p < 9; ++p) {
            double xp = p - 4.0;
            for (int i = 0; i < 9; ++i) {
                double xi = i - 4.0;
                /* a = (S4*sum y - S2*sum x^2 y)/det ; b = sum(x y)/S2 ; c = (S0*sum x^2 y - S2 sum y)/det */
                double ca = (S4 - S2 * xi * xi) / det;
                double cb = xi / S2;
                double cc = (S0 * xi * xi - S2) / det;
                C[p][i] = ca + cb * xp + cc * xp * xp;
            }
        }
    }
    for (int j = 0; j < nu; ++j) {
        for (int t = 0; t < T; ++t) {
            double acc = 0.0;
            if (t < 4) {
                for (int i = 0; i < 9; ++i) acc += C[t][i] * (double)in[i * nu + j];
            } else if (t >= T - 4) {
                int p = 8 - (T - 1 - t);
                for (int i = 0; i < 9; ++i) acc += C[p][i] * (double)in[(T - 9 + i) * nu + j];
            } else {
                for (int i = 0; i < 9; ++i) acc += C[4][i] * (double)in[(t - 4 + i) * nu + j];
            }
            out[t * nu + j] = (float)acc;
        }
    }
}

/* _shift_action: mppi.py:266-273 */
void m3o_shift(float* seq, int T, int nu) {
    for (int t = 0; t + 1 < T; ++t)
        for (int j = 0; j < nu; ++j) seq[t * nu + j] = seq[(t + 1) * nu + j];
    /* last row keeps its value (saved_action) */
}

/* simple mode: mppi.py:220-233 with _compute_total_cost_batch_simple :335-363.
 * cost_total = S + mean(S)  (aliasing quirk Q1, mppi.py:284,325)
 *            + sum_{t,j} U * ((lambda * noise) @ sigma_inv)   (:358-372; |noise| with noise_abs_cost)
 * noise = perturbed - U (after clamping, :355; `perturbed` holds the u_scale-d controls, :311) */
void m3o_simple_update(const m3o_cfg* cfg, const float* S, const float* perturbed, float* U,
                       float* cost_total, float* w) {
    const int K = cfg->K, T = cfg->T, nu = cfg->nu, n = T * nu;
    double ms = 0.0;
    for (int k = 0; k < K; ++k) ms += (double)S[k];
    float meanS = (float)(ms / (double)K);
    for (int k = 0; k < K; ++k) {
        double pc = 0.0;
        for (int t = 0; t < T; ++t) {
            float ln[M3O_MAX_NU];
            for (int j = 0; j < nu; ++j) {
                float noise = perturbed[(size_t)k * n + t * nu + j] - U[t * nu + j];
                if (cfg->noise_abs_cost) noise = fabsf(noise);           /* :366-367 */
                ln[j] = cfg->lambda_ * noise;
            }
            for (int j = 0; j < nu; ++j) {
                float ac;
                if (cfg->full_sigma) {                                    /* (lambda * noise) @ Sigma^-1 */
                    ac = ln[0] * cfg->sigma_inv_full[0 * nu + j];
                    for (int i = 1; i < nu; ++i) ac = ac + ln[i] * cfg->sigma_inv_full[i * nu + j];
                } else ac = ln[j] * cfg->sigma_inv[j];
                pc += (double)(U[t * nu + j] * ac);
            }
        }
        cost_total[k] = (S[k] + meanS) + (float)pc;
    }
    float mn = vmin(cost_total, K);
    double eta = 0.0;
    for (int k = 0; k < K; ++k) {
        w[k] = expf(-(1.0f / cfg->lambda_) * (cost_total[k] - mn)); /* skill_utils.py:3 */
        eta += (double)w[k];
    }
    float etaf = (float)eta;
    for (int k = 0; k < K; ++k) w[k] = (1.0f / etaf) * w[k];
    double* acc = (double*)calloc((size_t)n, sizeof(double));
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n; ++i)
            acc[i] += (double)(w[k] * (perturbed[(size_t)k * n + i] - U[i]));
    for (int i = 0; i < n; ++i) U[i] = U[i] + (float)acc[i]; /* mppi.py:231 */
    free(acc);
}

/* ------------------------------------------------------------------------------------
 * in-kernel noise stream (sampling_method='random', mppi.py:481 / :340): the reference
 * draws from torch's global RNG, which cannot be reproduced; the build defines its own
 * counter-based stream so results are shard-invariant (spec: DESIGN.md "Noise stream").
 *   state = splitmix64-seeded xoshiro128++ keyed by (seed, call, k, t, pair index)
 *   two uniforms -> Box-Muller -> (z0, z1); component j uses pair j/2, lane j%2.
 * ---------------------------------------------------------------------------------- */
static unsigned long long splitmix64(unsigned long long* x) {
    unsigned long long z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static unsigned rotl32(unsigned x, int k) { return (x << k) | (x >> (32 - k)); }
static unsigned xoshiro128pp(unsigned s[4]) {
    unsigned result = rotl32(s[0] + s[3], 7) + s[0];
    unsigned t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}
float m3o_gauss(unsigned long long seed, unsigned call, unsigned k, unsigned t, unsigned j) {
    unsigned long long x = seed ^ (0xD1B54A32D192ED03ULL * (unsigned long long)(call + 1u));
    x ^= ((unsigned long long)k << 32) | ((unsigned long long)t << 8) | (unsigned long long)(j >> 1);
    unsigned long long a = splitmix64(&x), b = splitmix64(&x);
    unsigned s[4] = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    unsigned r0 = xoshiro128pp(s), r1 = xoshiro128pp(s);
    /* (0,1] and [0,1) uniforms from the top 24 bits */
    float u0 = ((float)(r0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    float u1 = (float)(r1 >> 8) * (1.0f / 16777216.0f);
    float rad = sqrtf(-2.0f * logf(u0));
    float ang = 6.28318530717958647692f * u1;
    return (j & 1u) ? rad * sinf(ang) : rad * cosf(ang);
}

/* z[k-k0, t, j] for k in [k0, k0+n) : standard normals of the stream */
void m3o_gauss_fill(unsigned long long seed, unsigned call, int k0, int n, int T, int nu,
                    float* out) {
    for (int i = 0; i < n; ++i)
        for (int t = 0; t < T; ++t)
            for (int j = 0; j < nu; ++j)
                out[((size_t)i * T + t) * nu + j] =
                    m3o_gauss(seed, call, (unsigned)(k0 + i), (unsigned)t, (unsigned)j);
}

/* N(noise_mu, noise_sigma) draws of the stream (MultivariateNormal(...).sample, mppi.py:129-131, :340, :481):
 * d_j = mu_j + sum_{i<=j} L[j][i] z_i, accumulated i = 0, 1, .. (diagonal Sigma: mu_j + scale_tril_j z_j) */
void m3o_noise_fill(const m3o_cfg* cfg, unsigned long long seed, unsigned call, int k0, int n, float* out) {
    const int T = cfg->T, nu = cfg->nu;
    for (int i = 0; i < n; ++i)
        for (int t = 0; t < T; ++t) {
            float z[M3O_MAX_NU];
            for (int j = 0; j < nu; ++j) z[j] = m3o_gauss(seed, call, (unsigned)(k0 + i), (unsigned)t, (unsigned)j);
            for (int j = 0; j < nu; ++j) {
                float acc;
                if (cfg->full_sigma) {
                    acc = cfg->chol[j * nu + 0] * z[0];
                    for (int q = 1; q <= j; ++q) acc = acc + cfg->chol[j * nu + q] * z[q];
                } else acc = z[j] * cfg->chol[j * nu + j];   /* (= the configured sqrt(sigma_jj): scale_tril may have
                                                                been rewritten by update_cov, the distribution not) */
                out[((size_t)i * T + t) * nu + j] = cfg->noise_mu[j] + acc;
            }
        }
}

/* ------------------------------------------------------------------------------------
 * quaternion helpers: skill_utils.py:140-180 (xyzw -> R), :224-252, :256-290
 * ---------------------------------------------------------------------------------- */
static void quat_axes(const float Q[4], float R[9]) {
    float q0 = Q[3], q1 = Q[0], q2 = Q[1], q3 = Q[2];
    R[0] = 2 * (q0 * q0 + q1 * q1) - 1; R[1] = 2 * (q1 * q2 - q0 * q3); R[2] = 2 * (q1 * q3 + q0 * q2);
    R[3] = 2 * (q1 * q2 + q0 * q3); R[4] = 2 * (q0 * q0 + q2 * q2) - 1; R[5] = 2 * (q2 * q3 - q0 * q1);
    R[6] = 2 * (q1 * q3 - q0 * q2); R[7] = 2 * (q2 * q3 + q0 * q1); R[8] = 2 * (q0 * q0 + q3 * q3) - 1;
}
static float coldot(const float* A, int ca, const float* B, int cb) {
    return A[ca] * B[cb] + A[3 + ca] * B[3 + cb] + A[6 + ca] * B[6 + cb];
}
static float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }

/* get_general_ori_cube2goal: skill_utils.py:224-252 */
float m3o_ori_cube2goal(const float qc[4], const float qg[4]) {
    float C[9], G[9];
    quat_axes(qc, C);
    quat_axes(qg, G);
    float cx = min3(1 - fabsf(coldot(G, 0, C, 0)), 1 - fabsf(coldot(G, 0, C, 1)),
                    1 - fabsf(coldot(G, 0, C, 2)));
    float cy = min3(1 - fabsf(coldot(G, 1, C, 0)), 1 - fabsf(coldot(G, 1, C, 1)),
                    1 - fabsf(coldot(G, 1, C, 2)));
    return cx + cy;
}

/* get_general_ori_ee2cube: skill_utils.py:256-290.  For tilt_value != 0 the selected cube
 * axis index comes from the FIRST env of the batch slice (indice_list[0], :274) but the
 * axis vector itself is per-env. */
float m3o_ori_ee2cube(const float qe[4], const float qc[4], float tilt_value,
                      const float qc_env0[4]) {
    float E[9], C[9];
    quat_axes(qe, E);
    quat_axes(qc, C);
    float cost_z;
    if (tilt_value == 0.0f) {
        cost_z = min3(1 - fabsf(coldot(E, 2, C, 2)), 1 - fabsf(coldot(E, 2, C, 0)),
                      1 - fabsf(coldot(E, 2, C, 1)));
    } else {
        float C0[9];
        quat_axes(qc_env0, C0);
        /* argmax_i |axis_i . x| for env 0: stacked [x,y,z] axes, component 0 */
        int sel = 0;
        float best = fabsf(C0[0]);
        if (fabsf(C0[1]) > best) { best = fabsf(C0[1]); sel = 1; }
        if (fabsf(C0[2]) > best) { best = fabsf(C0[2]); sel = 2; }
        cost_z = fabsf(tilt_value - coldot(E, 2, C, sel));
    }
    float cost_y = min3(1 - fabsf(coldot(E, 1, C, 0)), 1 - fabsf(coldot(E, 1, C, 1)),
                        1 - fabsf(coldot(E, 1, C, 2)));
    return cost_z + cost_y;
}
