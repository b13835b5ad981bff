/*
 * m3_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the MPPI / M3P2I command() hot path of tud-amr/m3p2i-aip,
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker for the HIP kernels in m3p2i_aip_amd/csrc.  The product never links, loads or
 * calls anything in this directory.
 *
 * What is restated from the reference (citations relative to /root/reference):
 *   - action assembly            src/m3p2i_aip/planners/motion_planner/mppi.py:381-416
 *   - rollout loop bookkeeping   mppi.py:275-332
 *   - cost_to_go + softmin       src/m3p2i_aip/utils/mppi_utils.py:106-113, mppi.py:430-456
 *   - multi-modal beta search    src/m3p2i_aip/planners/motion_planner/m3p2i.py:24-92
 *   - mean update, top-k, savgol mppi.py:485-517, 245-264
 *   - simple-mode update         mppi.py:220-233, 335-373
 *   - task costs                 src/m3p2i_aip/planners/motion_planner/cost_functions.py:19-169
 *   - suction force model        src/m3p2i_aip/utils/skill_utils.py:59-94
 *   - quaternion costs           skill_utils.py:140-180, 224-290
 * These parts are PINNED against golden vectors produced by importing the reference's
 * own Python (tests/golden/make_golden.py).
 *
 * What has NO reference to restate: the rigid-body dynamics.  The reference steps
 * NVIDIA Isaac Gym Preview 4 / PhysX (closed binary, not vendored:
 * thirdparty/README.md:1-17, call sites isaacgym_wrapper.py:354-360).  The dynamics
 * here implement this repository's OWN written spec (DESIGN.md "Planar contact dynamics
 * spec v1" / "Panda chain spec v1") -- PARITY UNPINNED against PhysX.  The oracle's
 * role for the dynamics is to be an independent second implementation of that spec.
 *
 * All arithmetic is IEEE binary32, compiled with -ffp-contract=off; only + - * / sqrt
 * min max and comparisons are used inside the dynamics so that an independent
 * implementation can agree to the last bit.
 */
#ifndef M3_ORACLE_H
#define M3_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define M3O_MAX_NU 9

/* ---- tasks (cost_functions.py:19-36) ---- */
enum {
    M3O_TASK_NAVIGATION = 0,
    M3O_TASK_PUSH = 1,
    M3O_TASK_PULL = 2,
    M3O_TASK_PUSH_PULL = 3,
    M3O_TASK_REACH = 4,
    M3O_TASK_PICK = 5,
    M3O_TASK_PLACE = 6,
    M3O_TASK_IDLE = 7
};

/* ---- planar scene constants (config/point_env/ yaml files, assets/urdf/pointRobot.urdf,
 *      isaacgym_wrapper.py:18-37,335-351,462-469) ---- */
typedef struct {
    float dt;            /* isaacgym/point.yaml:4  (0.05) */
    int substeps;        /* isaacgym_wrapper.py:10 (2)    */
    int iters;           /* isaacgym_wrapper.py:28 (6 position iterations) */
    float g;             /* isaacgym_wrapper.py:25 (9.8)  */
    float robot_r;       /* pointRobot.urdf:18 radius 0.2 */
    float robot_m;       /* pointRobot.urdf:12 mass 10    */
    float drive_damping; /* isaacgym_wrapper.py:344 (600) */
    float drive_fmax;    /* pointRobot.urdf:35,43 effort 1000 */
    float box_hx, box_hy, box_m, box_I, box_mu_g, box_req; /* 7_box.yaml + ground plane */
    float dyn_hx, dyn_hy, dyn_m, dyn_I, dyn_mu_g, dyn_req; /* 6_dyn_obs.yaml */
    float obs_x, obs_y, obs_hx, obs_hy;                    /* 5_obs.yaml (fixed) */
    float wall;          /* inner wall face: 4.0 - 0.05 (1..4_wall.yaml) */
    float mu_rb, mu_rd, mu_ro, mu_rw, mu_bw, mu_dw, mu_bd, mu_bo, mu_do;
    float contact_offset; /* isaacgym_wrapper.py:30 (0.01) */
    float baumgarte, slop, max_bias, face_tol;
    int friction_coupling;   /* 1 = spec v1.5: sliding-spinning coupling of the boxes' ground friction; 0: spec v1.4; 2: the experimental four-corner patch rows (tools/cpu_ab_default_size.py only) */
    int fext_substeps;       /* 1 = spec v1.7: a pending external force is consumed by the first substep of the next step; 0 = spec v1.6, it acts in every substep (tools/cpu_fit_physx.py's `both` column only) */
} m3o_point_scene;

typedef struct { float x, y, c, s, vx, vy, w; } m3o_body;

typedef struct {
    m3o_body R, B, D;        /* robot disc, pushable box, dynamic obstacle */
    float fext_R[2], fext_B[2]; /* pending external force, consumed by the next step */
    float fc_R[2], fc_B[2], fc_D[2]; /* net contact force during the last substep */
} m3o_point_world;

void m3o_point_scene_default(m3o_point_scene* sc);
void m3o_point_world_init(m3o_point_world* w);
/* one sim.step(): substeps x (forces, detect, solve, integrate) */
void m3o_point_step(const m3o_point_scene* sc, m3o_point_world* w, const float u[2]);

void m3o_set_threads(int n);
int m3o_max_threads(void);

/* ---- MPPI configuration ---- */
typedef struct {
    int K;               /* GLOBAL number of samples */
    int T;
    int nu;
    int multi_modal;
    int env_type;        /* 0 point_env, 1 panda_env */
    int sample_null_action;
    int mode_simple;     /* 1: mppi_mode == 'simple' */
    int filter_u;
    int u_per_command;
    float u_min[M3O_MAX_NU], u_max[M3O_MAX_NU], scale_tril[M3O_MAX_NU], sigma_inv[M3O_MAX_NU];
    float u_scale;
    float gamma;
    float lambda_;
    float step_size_mean;
    int task;
    float goal[7];
    float kp_suction;
    float suction_thresh;  /* 1.8 for K>1, 1.5 for K==1 (skill_utils.py:75-82) */
    int gripper_cmd;       /* 0 undefined, 1 open, 2 close (m3p2i.py:10-14) */
    float pre_height_diff; /* config_panda.yaml:9 */
    float tilt_cos_theta;  /* cost_functions.py:13 */
    /* the MPPIConfig switches no shipped config turns on (mppi.py:39-54) */
    int noise_abs_cost;    /* mppi.py:366-367 */
    int full_sigma;        /* 1: noise_sigma has off-diagonal entries: chol / sigma_inv_full are used where the
                              reference uses the whole matrix (MultivariateNormal :129-131, the action cost :366-372) */
    float noise_mu[M3O_MAX_NU];                        /* mppi.py:127 */
    float chol[M3O_MAX_NU * M3O_MAX_NU];               /* row-major lower Cholesky factor of noise_sigma */
    float sigma_inv_full[M3O_MAX_NU * M3O_MAX_NU];     /* mppi.py:128 */
    /* an EXTENSION of this repository (off in every reference configuration): push / pull / push_pull add
       get_motion_cost (the dyn-obs contact penalty) as navigation does -- the shipped compute_cost returns before it for
       these tasks (cost_functions.py:23-29 vs :36), while the reference's logged `case2_*_coll` runs avoid the dyn-obs */
    int avoid_dyn_obs;
} m3o_cfg;

/* A4: delta[K,T,nu] (global) -> act[(k1-k0),T,nu] for global samples k0..k1-1.
 * means/best: [T,nu]. */
void m3o_assemble_actions(const m3o_cfg* cfg, const float* delta, const float* mean,
                          const float* mean1, const float* mean2, const float* best1,
                          const float* best2, int k0, int k1, float* act);

/* per-step point-env cost for one sample (global index k); may write pending suction */
float m3o_point_cost(const m3o_cfg* cfg, m3o_point_world* w, int k);

void m3o_point_step_batch(const m3o_point_scene* sc, float* worlds, int n, const float* u);
void m3o_point_cost_batch(const m3o_cfg* cfg, float* worlds, int n, int k0, float* c);

/* A5: rollout of samples k0..k1-1 from one common initial world.
 * pend[(k1-k0)*4]: pending external force (fext_R, fext_B) carried across commands.
 * Outputs (local sample-major): states[n,T,4], actions[n,T,nu], cost_h[n,T], J[n], S[n]. */
void m3o_point_rollout(const m3o_cfg* cfg, const m3o_point_scene* sc,
                       const m3o_point_world* w0, float* pend, const float* act, int k0,
                       int k1, float* states, float* actions, float* cost_h, float* J,
                       float* S);

/* A10: J[k] = sum_t gamma^t c[k,t] */
void m3o_cost_to_go0(const float* cost_h, int K, int T, float gamma, float* J);
/* A10: softmin with fixed beta; returns eta. w may alias nothing. */
float m3o_softmin(const float* J, int n, float beta, float* w);
/* A11: on-the-fly beta search (m3p2i.py:24-44); returns eta, writes w (normalised),
 * *iters = number of loop passes, *beta_out final beta. max_iters guards divergence. */
float m3o_beta_search(const float* J, int n, float beta0, float eta_u, float eta_l,
                      int max_iters, float* w, int* iters, float* beta_out);

typedef struct {
    float beta;          /* persisted beta (panda single-mode, mppi.py:446-454) */
    int best_idx, best_idx_1, best_idx_2;
    float eta, eta_1, eta_2;
    int iters, iters_1, iters_2;
    float wsum_push, wsum_pull;
} m3o_update_info;

/* A10+A12 (single) / A11 (multi): global J[K], local shard actions[(k1-k0),T,nu].
 * Produces normalised weights for ALL K (w[K], w1[K/2], w2[K/2]) and PARTIAL sums over the
 * shard: psum[T,nu] (and psum1, psum2).  best rows are written only if the argmax lies in
 * the shard (else left untouched). */
void m3o_update_weights(const m3o_cfg* cfg, const float* J, float* w, float* w1, float* w2,
                        m3o_update_info* info);
void m3o_partial_sums(const m3o_cfg* cfg, const float* w, const float* w1, const float* w2,
                      const float* actions, int k0, int k1, float* psum, float* psum1,
                      float* psum2);
/* mean <- (1-a) mean + a * sum */
void m3o_mean_update(const m3o_cfg* cfg, float* mean, const float* sum);
/* top-n of w (descending, lowest index first on ties) */
void m3o_topk(const float* w, int K, int n, int* idx, float* val);
/* Savitzky-Golay window 9, order 2, mode 'interp' along T (mppi.py:257-263) */
void m3o_savgol9(const float* in, int T, int nu, float* out);
/* warm-start shift (mppi.py:266-273) */
void m3o_shift(float* seq, int T, int nu);

/* simple mode (mppi.py:220-233,335-373): given S[K] (undiscounted sums), U[T,nu],
 * perturbed[K,T,nu] (after clamp), computes cost_total, weights, and updates U. */
void m3o_simple_update(const m3o_cfg* cfg, const float* S, const float* perturbed, float* U,
                       float* cost_total, float* w);

/* xoshiro128++ / Box-Muller stream shared with the kernels (spec in DESIGN.md) */
float m3o_gauss(unsigned long long seed, unsigned call, unsigned k, unsigned t, unsigned j);

void m3o_gauss_fill(unsigned long long seed, unsigned call, int k0, int n, int T, int nu,
                    float* out);
void m3o_noise_fill(const m3o_cfg* cfg, unsigned long long seed, unsigned call, int k0, int n, float* out);

/* ---- panda_env: "Panda world spec v2"  ---- */
typedef struct {
    float dt; int substeps; float g;
    float base[3];
    float drive_damping;
    float inertia[9], effort[9], vlim[9], qlo[9], qhi[9];
    float table[6], shelf[6];      /* centre xyz + half extents */
    float cube_half, cube_m, cube_mu;
    float grasp_z, grasp_dx, grasp_dz, grasp_align, grasp_tol;
    float k_contact;               /* (spec v1's penalty stiffness: unused by v2, kept for the struct's layout) */
    float tip_z, tip_r, hand_z, hand_r;
    /* spec v2: the contact solver (isaacgym_wrapper.py:26-31) and the free bodies */
    int iters;                     /* velocity passes per substep */
    float contact_offset, slop, baumgarte, max_bias, act_margin;
    float mu;                      /* friction of every pair (actor_utils.py:27 default 1.0, averaged) */
    float obs_half[3], obs_m;      /* 4_obs.yaml */
    float sleep_v, sleep_w;
    float rest_gap;                /* a loaded world's cube rests when its facing face is within this of a top face */
} m3o_panda_scene;

#define M3O_PANDA_WORLD_FLOATS 84
typedef struct {
    float q[9], qd[9];
    float cubeA[13], cubeB[13], obs[13];  /* pos3 quat4(xyzw) linvel3 angvel3 (the plate does not rotate) */
    float held;                    /* 1.0 while cubeA is clamped between the finger pads */
    float rel_p[3], rel_q[4];      /* cubeA pose in the hand frame while held */
    float awake[2];                /* cubeA, cubeB: 0.0 while the cube sleeps on the table / the shelf stand */
    float f_table[3], f_shelf[3], f_cubeB[3]; /* net contact force of the last substep */
    float warm_t[4], warm_l[4];    /* per collision sphere: 1 + the target of last substep's contact (0 none), its normal impulse */
} m3o_panda_world;

typedef struct {                   /* link0..7, hand, leftfinger, rightfinger */
    float pos[11][3], quat[11][4], ax[11][3], ay[11][3], az[11][3];
} m3o_panda_links;

typedef struct {                   /* what the reference's panda costs read from the wrapper */
    float left[3], left_q[4], right[3];
    float cube[3], cube_q[4];
    float cube0[3];                /* cubeA position of env 0 (cost_functions.py:97) */
    float cube_q_half0[4];         /* cubeA orientation of the first env of the slice (skill_utils.py:274) */
    float f_table[2], f_shelf[2], f_cubeB[2];
} m3o_panda_obs;

void m3o_sincos(float x, float* s, float* c);
void m3o_panda_scene_default(m3o_panda_scene* sc);
void m3o_panda_world_init(m3o_panda_world* w, int cube_on_shelf);
void m3o_panda_fk(const m3o_panda_scene* sc, const float q[9], m3o_panda_links* L);
void m3o_panda_step(const m3o_panda_scene* sc, m3o_panda_world* w, const float u[9]);
void m3o_panda_infer_held(const m3o_panda_scene* sc, m3o_panda_world* w);   /* held AND the cubes' sleep state */
/* diagnostics of the last m3o_panda_step call of this thread: contact rows of its last substep */
int m3o_panda_last_rows(int* robot_rows, int* body_rows);
void m3o_panda_observe(const m3o_panda_scene* sc, const m3o_panda_world* w, m3o_panda_obs* o);
float m3o_panda_cost_obs(const m3o_cfg* cfg, const m3o_panda_obs* o, int k);
void m3o_panda_rollout(const m3o_cfg* cfg, const m3o_panda_scene* sc, const m3o_panda_world* w0,
                       const float* act, int k0, int k1, float* states, float* actions,
                       float* cost_h, float* J);
void m3o_panda_step_batch(const m3o_panda_scene* sc, float* worlds, int n, const float* u);
void m3o_panda_cost_obs_batch(const m3o_cfg* cfg, const m3o_panda_obs* obs, int n, int k0, float* c);

/* quaternion costs (skill_utils.py:140-180, 224-290), q = xyzw */
float m3o_ori_cube2goal(const float qc[4], const float qg[4]);
float m3o_ori_ee2cube(const float qe[4], const float qc[4], float tilt_value,
                      const float qc_env0[4]);

/* world spec v3: the fixed summation trees of the contact rows (panda_chain.c) -- exported for the tests that emulate the
 * product kernel's cross-lane butterflies against them */
float m3o_sum16(const float x[16]);
float m3o_sum8_6(const float x[6]);

/* planar spec v1.6: the substep's reciprocal (planar_world.c), exported for the test that bounds its error */
float m3o_spec_rcp(float x);

#ifdef __cplusplus
}
#endif
#endif
