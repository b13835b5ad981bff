/*
 * m3p2i_hip.h -- C-ABI of libm3p2i_hip.so: the MI355X (gfx950) implementation of the
 * MPPI / M3P2I command() hot path of tud-amr/m3p2i-aip.
 *
 * The reference is pure Python; the "FFI" a maintainer binds is ctypes (see
 * INTEGRATION.md).  Each entry point names the reference interface it replaces
 * (paths relative to the reference repository root):
 *
 *   m3_create / m3_destroy      MPPI.__init__ / M3P2I.__init__
 *                                 src/m3p2i_aip/planners/motion_planner/mppi.py:82-203,
 *                                 m3p2i.py:5-8; IsaacGymWrapper.__init__/start_sim
 *                                 src/m3p2i_aip/utils/isaacgym_utils/isaacgym_wrapper.py:39-90
 *   m3_set_noise                MPPI.get_samples (cached Halton-spline delta) mppi.py:458-483
 *   m3_set_objective            Objective.update_objective cost_functions.py:15-17,
 *                                 M3P2I.update_gripper_command m3p2i.py:10-14
 *   m3_set_world / m3_bind_sim  run_tamp state upload scripts/reactive_tamp.py:45-48
 *                                 (set_dof_state_tensor / set_actor_root_state_tensor,
 *                                 isaacgym_wrapper.py:190-194)
 *   m3_command                  MPPI.command mppi.py:211-264 (whole call)
 *   m3_rollout                  _compute_total_cost_batch_halton/_simple + _compute_rollout_costs
 *                                 mppi.py:275-332, 335-363, 381-428 with dynamics/running_cost
 *                                 of reactive_tamp.py:63-73 and Objective.compute_cost
 *                                 cost_functions.py:19-169 fused in
 *   m3_update                   _exp_util mppi.py:430-456, _multi_modal_exp_util +
 *                                 update_infinite_beta m3p2i.py:24-64, weighted sums
 *                                 mppi.py:493-498, m3p2i.py:75-83
 *   m3_finalize                 mean update mppi.py:502-503 / m3p2i.py:86-87, top-k +
 *                                 Savitzky-Golay mppi.py:245-264, simple-mode U update :231
 *   m3_sim_*                    IsaacGymWrapper step-mode surface: set_dof_velocity_target_tensor
 *                                 :196, step :354-360, apply_rigid_body_force_tensors :202,
 *                                 refresh of _dof_state/_root_state/_rigid_body_state/
 *                                 _net_contact_force :98-118
 *   m3_cost                     Objective.compute_cost (step mode) cost_functions.py:19-36
 *   m3_sim_suction_forces /     calculate_suction, check_and_apply_suction utils/skill_utils.py:36-94
 *   m3_sim_check_and_apply_suction   (the real-world side of scripts/sim.py:41-49)
 *   m3_get_buffer               attribute access MPPI.states/actions/weights/top_trajs/...
 *   m3_get_info                 M3P2I.get_pull_preference m3p2i.py:16-22 (+ diagnostics)
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every call returns 0 on success
 * or a negative m3_status; m3_last_error() gives the message.  All device work is enqueued
 * on the handle's HIP stream (m3_set_stream); calls are asynchronous unless stated.  A
 * handle is not thread-safe; different handles are independent.  The library owns all
 * device buffers; m3_get_buffer hands out non-owning device pointers valid until
 * m3_destroy.
 */
#ifndef M3P2I_HIP_H
#define M3P2I_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define M3_MAX_NU 9
#define M3_TOPK 20
#define M3_ABI_VERSION 4

typedef enum {
    M3_OK = 0,
    M3_ERR_BAD_ARG = -1,
    M3_ERR_HIP = -2,
    M3_ERR_SHAPE = -3,
    M3_ERR_STATE = -4,
    M3_ERR_UNSUPPORTED = -5
} m3_status;

typedef enum { M3_ENV_POINT = 0, M3_ENV_PANDA = 1 } m3_env;

/* cost_functions.py:19-36 task strings */
typedef enum {
    M3_TASK_NAVIGATION = 0,
    M3_TASK_PUSH = 1,
    M3_TASK_PULL = 2,
    M3_TASK_PUSH_PULL = 3,
    M3_TASK_REACH = 4,
    M3_TASK_PICK = 5,
    M3_TASK_PLACE = 6,
    M3_TASK_IDLE = 7
} m3_task;

typedef struct {
    int abi_version;        /* M3_ABI_VERSION */
    int device;             /* HIP device ordinal */
    /* sampling / sharding: rank owns global samples [k_offset, k_offset + K_local) */
    int K_global;
    int K_local;
    int k_offset;
    int T;                  /* horizon */
    int nu;                 /* 2 (point_env) or 9 (panda_env) */
    int env_type;           /* m3_env */
    int multi_modal;        /* cfg.multi_modal */
    int mode_simple;        /* mppi_mode == 'simple' */
    int sampling_random;    /* sampling_method == 'random' (in-kernel xoshiro128++) */
    int sample_null_action;
    int filter_u;
    int u_per_command;
    float u_min[M3_MAX_NU];
    float u_max[M3_MAX_NU];
    float noise_sigma_diag[M3_MAX_NU]; /* diagonal of cfg.mppi.noise_sigma */
    float u_scale;          /* mppi.py:297.  M3_BUF_ACTIONS holds the SCALED controls u_scale * a (what the dynamics
                               got and what the reference's distribution update consumes, mppi.py:313-331); the
                               division of mppi.py:353 / :420 only concerns the attribute a caller reads */
    float gamma;            /* rollout_var_discount */
    float lambda_;
    float step_size_mean;   /* 0.98, mppi.py:178 */
    float kp_suction;       /* config_point.yaml:9 */
    float pre_height_diff;  /* config_panda.yaml:9 */
    float dt;               /* isaacgym/{point,panda}.yaml:4 */
    int substeps;           /* isaacgym_wrapper.py:10 */
    int solver_iters;       /* isaacgym_wrapper.py:28 */
    int cube_on_shelf;      /* reactive_tamp.py:29 */
    int sim_only;           /* 1: handle is used only through m3_sim_* / m3_cost (the wrapper's
                               environments); planner-shape checks (K >= 20, filter rows) are
                               skipped and m3_rollout/m3_update are refused */
    int shard_mix;          /* sharded handles (K_local < K_global): 1 = ONE collective per command, an
                               all-gather of per-rank records (M3_BUF_RECORD -> M3_BUF_RECORDS_ALL) between
                               m3_update and m3_finalize.
                               * single-mode MPPI (beta fixed during a command, mppi.py:430-456): m3_update
                                 runs the softmin on the rank's own shard (local minimum m_r, local eta_r,
                                 normalised local sums) into the record; m3_finalize mixes the ranks' records
                                 with rho_r = exp(-(m_r - m)/beta) eta_r / sum_r(...) -- exact in real
                                 arithmetic, equal to the gather + reduce protocol up to f32 rounding.
                               * multi-modal M3P2I (on-the-fly beta search over ALL samples, m3p2i.py:24-64):
                                 the record is {the shard's trajectory costs | its top-k}; after the gather
                                 every rank holds all K_global costs and runs the unsharded update on them,
                                 RE-GENERATING the other ranks' actions from the replicated noise table and
                                 plan (a function of the global sample index, mppi.py:381-416) instead of
                                 receiving them: bit-identical to the unsharded handle.  Needs the noise rows
                                 of all K_global samples on every rank (m3_set_noise_global /
                                 m3_set_noise_knots_global; 15 MB at K = 64000).
                               2 = (multi-modal only) the same with each shard's minima and its eta(beta) sums on
                                 the beta ladders added to the record: after the gather the searches walk the
                                 MIXTURE of the shards' tables (eta(beta_j) = sum_r exp(-(m_r - m)/beta_j) eta_r(beta_j))
                                 instead of re-evaluating all K costs -- passes over the gathered costs only when
                                 a search leaves its ladder -- and one kernel forms weights, sums, best rows and the
                                 plan: per-rank work after the collective halves.  Equal to the unsharded run up to
                                 f32 rounding (plan <= 3e-5, same beta-search iteration counts), bit-identical
                                 across ranks.
                               3 = (multi-modal only) TWO small exchanges with O(K_local) work per rank in between
                                 (m3_update_b below): the record of 2, then every rank's weighted sums of its OWN
                                 samples; pays from K_global ~ 300 k (8 x 131072: 0.402 vs 0.441 ms per command).
                               0 = gather + reduce, two collectives (all-gather TRAJ_COST, all-reduce REDUCE) */
    /* ---- the MPPIConfig switches no shipped config turns on (mppi.py:39-54); zero = the defaults ---- */
    int noise_abs_cost;     /* mppi.py:366-367: |noise| in the action cost (read in simple mode only, as in the reference) */
    int update_cov;         /* mppi.py:201, :508-516: after every command of a single-mode halton-spline planner
                               cov_action <- 0.3 cov_action + 0.7 mean_t sum_k w_k (a_k - mean)^2 + 0.005 and
                               scale_tril = sqrt(cov_action) (M3_BUF_COV: the rollout reads its scale from there).
                               Ignored for multi_modal / simple mode, exactly as the reference ignores it there
                               (m3p2i.py:66-92 and mppi.py:220-233 have no such branch); unsharded handles only */
    int full_sigma;         /* 0: noise_sigma = diag(noise_sigma_diag).  1: noise_sigma_full is the matrix; its
                               diagonal must equal noise_sigma_diag.  Only the paths where the reference uses the whole
                               matrix read it: MultivariateNormal sampling (sampling_random / simple mode,
                               mppi.py:129-131, :340, :481) through its Cholesky factor and the action cost
                               (mppi.py:128, :366-372) through its inverse, both formed in binary64 and rounded to f32;
                               the Halton noise is scaled by sqrt(diag) alone (mppi.py:175-176, :394) */
    float noise_mu[M3_MAX_NU];                         /* mppi.py:127-131: mean of the sampled noise */
    float noise_sigma_full[M3_MAX_NU * M3_MAX_NU];     /* row-major [nu][nu] */
    unsigned long long seed;
} m3_config;

/* Planar world state of ONE environment, as the caller sees it through the wrapper's
 * tensors (dof_state = [x, vx, y, vy]; boxes = x, y, yaw quaternion (z, w), vx, vy, wz). */
typedef struct {
    float robot[4];      /* x, y, vx, vy */
    float box[7];        /* x, y, qz, qw, vx, vy, wz */
    float dyn_obs[7];
} m3_point_world;

typedef struct {
    float eta, eta_1, eta_2;
    float beta, beta_1, beta_2;  /* beta AFTER the call (panda single-mode persists it) */
    int iters, iters_1, iters_2; /* beta-search passes */
    int best_idx, best_idx_1, best_idx_2; /* GLOBAL sample indices */
    float wsum_push, wsum_pull;  /* sum of weights of the two halves (m3p2i.py:18-19) */
    int pull_preference;
    int calls;
} m3_info;

typedef struct {
    float rollout_ms, update_ms, finalize_ms, total_ms; /* HIP-event times of the last
                                                           m3_command with timing enabled */
} m3_timing;

/* buffers for m3_get_buffer; shapes in comments (Kl = K_local, Kg = K_global).
 * Layouts are TIME-MAJOR so that consecutive lanes (samples) touch consecutive addresses. */
typedef enum {
    M3_BUF_STATES = 0,      /* f32 [T][Kl][4]   (x, vx, y, vy)  mppi.py:318 (transposed) */
    M3_BUF_ACTIONS = 1,     /* f32 [T][Kl][nu]                 mppi.py:317 (transposed) */
    M3_BUF_COST_HORIZON = 2,/* f32 [T][Kl]                     mppi.py:310 (transposed) */
    M3_BUF_TRAJ_COST = 3,   /* f32 [Kl]  discounted J (halton) / S + perturbation (simple) */
    M3_BUF_TRAJ_COST_ALL = 4,/* f32 [Kg] gathered J used by m3_update */
    M3_BUF_WEIGHTS = 5,     /* f32 [Kg] */
    M3_BUF_WEIGHTS_1 = 6,   /* f32 [Kg/2] */
    M3_BUF_WEIGHTS_2 = 7,   /* f32 [Kg - Kg/2] */
    M3_BUF_MEAN = 8,        /* f32 [T][nu] mean_action (U in simple mode) */
    M3_BUF_MEAN_1 = 9,
    M3_BUF_MEAN_2 = 10,
    M3_BUF_BEST = 11,       /* f32 [T][nu] best_traj */
    M3_BUF_BEST_1 = 12,
    M3_BUF_BEST_2 = 13,
    M3_BUF_ACTION_OUT = 14, /* f32 [T][nu] returned plan (filtered) */
    M3_BUF_TOP_IDX = 15,    /* i32 [M3_TOPK] global indices */
    M3_BUF_TOP_TRAJS = 16,  /* f32 [M3_TOPK][T][2] */
    M3_BUF_REDUCE = 17,     /* f32 [reduce_len] packed partial sums: all-reduce(sum) this
                               buffer between m3_update and m3_finalize when sharded */
    M3_BUF_NOISE = 18,      /* f32 [T][Kl][nu] delta (time-major copy of m3_set_noise) */
    M3_BUF_PENDING_FORCE = 19, /* f32 [4][Kl] suction force pending for the next step */
    M3_BUF_INFO = 20,       /* device copy of m3_info */
    M3_BUF_SIM_WORLD = 21,  /* f32 [28][Kl] step-mode environments (SoA; rows 18..21 = pending
                               force in M3_BUF_PENDING_FORCE order); allocated on first use */
    M3_BUF_RECORD = 22,     /* f32 [record_len] this rank's record (shard_mix).  Single-mode: header {m_r,
                               eta_r, half sums, best idx, top-k costs and indices} + a REDUCE-shaped
                               body (normalised local sums, best rows, top trajectories).  Multi-modal:
                               [Kl] trajectory costs (M3_BUF_TRAJ_COST aliases them) | top-k costs |
                               top-k global indices | top-k trajectories [M3_TOPK][T][2] */
    M3_BUF_RECORDS_ALL = 23,/* f32 [K_global/K_local][record_len]: all-gather M3_BUF_RECORD into
                               this buffer between m3_update and m3_finalize (shard_mix) */
    M3_BUF_NOISE_ALL = 24,  /* f32 [K_global/K_local][T][Kl][nu]: every shard's noise block (one-collective
                               multi-modal shards only; M3_BUF_NOISE is this rank's block of it) */
    M3_BUF_COV = 25,        /* f32 [2][nu]: cov_action | scale_tril (mppi.py:175-176; rewritten by every command
                               when update_cov is on) */
    M3_BUF_RECORD_B = 26,   /* f32 [recb_len] shard_mix = 3: this rank's SECOND record -- {-w, index} of its best sample per
                             * weight set, its half sums, its weighted action sums [3][T][nu] and best rows [3][T][nu] */
    M3_BUF_RECORDS_B_ALL = 27, /* f32 [K_global/K_local][recb_len]: all-gather M3_BUF_RECORD_B into it before m3_finalize */
    M3_BUF_COUNT = 28
} m3_buffer_id;

typedef struct m3_handle m3_handle;

int m3_abi_version(void);
const char* m3_last_error(const m3_handle* h); /* h may be NULL: error of the last failed
                                                  m3_create on this thread */
/* sha256 prefix (16 hex digits) of the kernel sources + compiler flags this library was built from (m3p2i_aip_amd/build.py:
 * source_hash); identical for a rebuild of the same tree: the key of the committed PMC profiles (profiles/, bench.py). */
const char* m3_build_id(void);
void m3_default_config(m3_config* cfg, int env_type);
int m3_create(const m3_config* cfg, m3_handle** out);
void m3_destroy(m3_handle* h);
int m3_set_stream(m3_handle* h, void* hip_stream);
int m3_enable_timing(m3_handle* h, int on);
/* launch geometry of the rollout kernel: samples (active lanes) per 64-wide wavefront,
 * a power of two in 1..64, or 0 = choose from K_local so that the waves fill the chip
 * (DESIGN.md "Lanes per wavefront").  Results do not depend on it. */
int m3_set_rollout_lanes(m3_handle* h, int lanes);
/* panda_env: lanes that simulate ONE sample in the rollout kernel -- 1 (a lane per sample, 64 samples per wavefront), 8 or
 * 16 (the lanes of a DPP row share a sample: the contact solver's joint-space rows run across them; eight / four sample
 * slots per wavefront), 0 = automatic: by size (16 while the launch has no more wavefronts than the chip has SIMDs, then 8,
 * then 1); for the reach task (quirk Q8) 16 with the reach cost kernel (below) wherever that is available and one round of
 * sixteen-lane wavefronts fits (K <= 4096), else by what the last commands' rollouts met -- 1 (with shadow sample slots) while
 * few of them had the gripper within reach of a box or an awake cube, 8 once many did.  Same results, bit for bit
 * (world spec v3 defines the rows' sums as the pairwise tree all forms evaluate).  DESIGN.md section 6. */
int m3_set_panda_lanes_per_sample(m3_handle* h, int lanes_per_sample);
/* panda_env, reach task on an unsharded handle (quirk Q8: every rollout's cost is measured against environment 0's cube):
 * 1 (default) = with K <= 8192 and the default sampler the rollout kernel runs WITHOUT shadow sample slots in a many-lane form
 * and leaves what the cost reads per (step, sample) behind; a second kernel forms the costs (same values, same operations, same
 * bits).  0 = the shadow slots (every wavefront re-simulates samples 0 and K / 2), as for larger K, the random sampler and
 * mppi_mode 'simple'. */
int m3_set_panda_reach_cost_kernel(m3_handle* h, int on);
/* the form the last panda rollout ran in (1, 8, 16; 0 before the first) */
int m3_panda_lanes_per_sample_used(m3_handle* h);
/* what the automatic choice for reach reads: the share, in 1/1000, of the last FINISHED panda rollout launch's (sample,
 * substep) pairs in which the gripper was within reach of a box or a cube was awake (the kernel's last wavefront reports
 * it into mapped host memory; no synchronisation); -1 before the first report */
int m3_panda_near_share(m3_handle* h);
/* launch structure of the unsharded multi-modal update with K beyond the one-launch kernel's range: 0 (default) = three
 * launches (ladder + search in one grid, weights + sums, combine), 5 = the five launches of round 3 (minima, ladder,
 * search, weights, sums), whose sums are added in the order of the sharded "exact" protocols (tests compare those bit
 * for bit).  Pass counts, best samples and the plan (to rounding) do not depend on it. */
int m3_set_update_launches(m3_handle* h, int launches);

/* assignment of samples to wavefronts in the point_env rollout: 1 (default) = sorted by the
 * direction of each sample's noise path (computed on the device whenever the noise is set, so that
 * a wavefront's samples meet the same obstacles), 0 = by index.  Results do not depend on it. */
int m3_set_wave_order(m3_handle* h, int on);
/* bound of the waits inside the multi-modal update's launches (k_update_small's column workgroups wait for each other's
 * beta-ladder points, k_ladder_search's search workgroup for the ladder workgroups' flags; a wait that runs out -- other
 * kernels occupying the CUs -- makes the waiting workgroup run the reference's iterative passes itself: same pass counts,
 * same weights): -1 = default (2^18 polls, ~20 ms), 0 = never wait (every workgroup takes the give-up branch: the tests
 * of that branch), n > 0 = n polls. */
int m3_set_ladder_spins(m3_handle* h, int spins);
/* Relabels the samples instead: the noise rows are permuted ONCE (by the next m3_rollout, which knows
 * the world) into that order, within each mode's half and with the rows of the special samples (0,
 * K/2, K-1) left in place, so that afterwards sample k simply has another row of the SAME noise set
 * and index order is wavefront order (coalesced stores, no per-command scatter).  For a caller whose
 * row labels carry no meaning -- the reference's own Halton sampler: planner._ensure_noise calls it.
 * The sample SET, and with it the plan, is unchanged up to the order of f32 summation; per-sample
 * outputs (weights[k], states[k], top_idx) refer to the new labels. */
int m3_relabel_samples(m3_handle* h);

/* delta: [K_local][T][nu] row-major (the reference's layout, rows of THIS shard).
 * on_device: 0 host pointer, 1 device pointer. */
int m3_set_noise(m3_handle* h, const float* delta, int on_device);
/* The reference's Halton-spline sampler on the device (mppi.py:458-483, mppi_utils.py bspline):
 * knots [K_local][nu][n_knots] (the Gaussian Halton values of THIS shard's samples); every one of
 * the K_local*nu series is fitted with FITPACK's smoothing spline (splrep(linspace(0,n,n), y,
 * k=degree, s=smoothing)) and evaluated at linspace(0, n, T) with ext=3 (splev), one thread per
 * series, into M3_BUF_NOISE.  Bit-identical to scipy 1.15.3's FITPACK (tests/test_spline_fit.py). */
int m3_set_noise_knots(m3_handle* h, const float* knots, int n_knots, int degree, float smoothing,
                       int on_device);
/* the same two for a one-collective multi-modal shard (cfg.shard_mix with multi_modal): rows / knots of
 * ALL K_global samples, [K_global][T][nu] / [K_global][nu][n_knots] -- every rank holds the whole table */
int m3_set_noise_global(m3_handle* h, const float* delta_all, int on_device);
int m3_set_noise_knots_global(m3_handle* h, const float* knots_all, int n_knots, int degree, float smoothing,
                              int on_device);
/* The WHOLE reference sampler on the device: Halton radical inverses (in-tree use_ghalton=False branch,
 * mppi_utils.py:69-96) -> sqrt(2) erfinv(2u - 1) (:99-104) -> the smoothing spline of m3_set_noise_knots, for
 * this shard's samples (all K_global for a one-collective multi-modal shard).  The uniform Halton values are
 * bit-identical to the host sampler's; the Gaussian ones agree to ~1e-6 relative (device erff / expf / logf
 * differ from the host's libm in the last ulp) -- so the planner's default stays m3_set_noise_knots with host
 * knots, pinned bit for bit by golden G8; this entry (MPPIConfig.device_knots) removes the last host values. */
int m3_set_noise_halton(m3_handle* h, int n_knots, int degree, float smoothing);
/* The same with a GENERALIZED (digit-permuted) Halton sequence -- the structure of the branch the reference's
 * planner actually executes: mppi.py:465-471 calls generate_gaussian_halton_samples(..., use_ghalton=True) ->
 * ghalton.GeneralizedHalton(EA_PERMS[:ndims]) (mppi_utils.py:89-95).  `ghalton` is a third-party package that is
 * absent here and unpinned in the reference (pyproject.toml:15), so its EA_PERMS table cannot be restated; what can
 * be is the construction with a published, formula-defined permutation set:
 *   M3_HALTON_FAURE   H. Faure, "Good permutations for extreme discrepancy", J. Number Theory 42 (1992) 47-56.
 * The plain sequence's high dimensions are strongly correlated (the panda_env knots are 45-dimensional: primes up to
 * 197, |corr| up to 0.45 between neighbouring dimensions at K = 4000; 0.08 with Faure's permutations).
 * M3_HALTON_PLAIN == m3_set_noise_halton (the default: pinned by golden G8). */
#define M3_HALTON_PLAIN 0
#define M3_HALTON_FAURE 1
int m3_set_noise_halton_scrambled(m3_handle* h, int n_knots, int degree, float smoothing, int scramble);
/* sampling_random / simple mode: the draws of N(noise_mu, noise_sigma) that the NEXT m3_rollout generates in
 * registers (MultivariateNormal(...).sample((K, T)), mppi.py:340 / :481), written to M3_BUF_NOISE for a caller
 * that runs the rollout itself (the planner's STEP mode: user dynamics / running_cost callables) */
int m3_sample_noise(m3_handle* h);
int m3_set_objective(m3_handle* h, int task, const float* goal, int goal_len, int gripper_cmd);
/* EXTENSION, off by default (= the reference): push / pull / push_pull add get_motion_cost -- 1000 while the dyn-obs
 * feels a contact force > 0.1 (cost_functions.py:158-169) -- the way navigation does.  The shipped compute_cost
 * returns before that term for these tasks (cost_functions.py:23-29 vs :36), so the reference's push / pull rollouts are
 * blind to the dyn-obs; its logged experiments `plot/point/case2_halton_{push,pull}_coll.npy` show 3 / 60 and 1 / 60
 * collisions, i.e. were made with the term active.  point_env only.  The rollout then runs its general instance. */
int m3_set_avoid_dyn_obs(m3_handle* h, int on);
/* Objective.multi_modal (cost_functions.py:9) for a sim_only handle, whose config does not
 * come from an MPPI object; refused on planner handles (fixed at m3_create) */
int m3_set_multi_modal(m3_handle* h, int multi_modal);
/* warm-start state (means, best trajs, U): which = M3_BUF_MEAN.., host pointer [T][nu] */
int m3_set_plan(m3_handle* h, int which, const float* host_values);
/* the persistent softmin temperature (MPPI.beta, mppi.py:184; adapted by the panda_env's single-mode
 * update, mppi.py:446-454): together with m3_set_plan this restores a saved warm start */
int m3_set_beta(m3_handle* h, float beta);
/* the number of finished commands (m3_info.calls) = the index of the next command in the in-kernel noise stream;
 * with m3_set_plan / m3_set_beta this restores a saved planner state exactly */
int m3_set_call_count(m3_handle* h, unsigned calls);
int m3_reset(m3_handle* h); /* zero means/best/pending forces, beta = 1, call counter = 0 */
/* Where m3_finalize writes the returned plan (`action`, mppi.py:257-263): a caller-owned device
 * buffer of [T][nu] floats ([u_per_command][nu] in simple mode), or NULL for the library's
 * M3_BUF_ACTION_OUT.  The reference returns a fresh tensor per command(); a host wrapper that
 * hands out slots of a small ring through this call gets the same aliasing behaviour without a
 * device-to-device copy per command. */
int m3_set_action_out(m3_handle* h, float* dev_ptr);

/* initial state of every rollout (host values; copied by value into the launch) */
int m3_set_world_point(m3_handle* h, const m3_point_world* w);
/* same in the library's internal layout, 18 floats: robot x y vx vy | box x y cos sin vx vy wz
 * | dyn-obs likewise (no quaternion round trip; used by the bit-parity tests) */
int m3_set_world_point_raw(m3_handle* h, const float* w18);
/* ... or read it from env 0 of the wrapper's device tensors at launch time (zero-copy):
 * dof_state f32 [*,4], root_state f32 [*,n_actors,13] */
int m3_bind_sim_point(m3_handle* h, const float* dof_state_dev, const float* root_state_dev,
                      int n_actors, int box_actor, int dyn_obs_actor);

/* panda_env counterparts.  w57 = q[9] qd[9] | cubeA | cubeB | dyn-obs, each pos3 quat4(xyzw) linvel3 angvel3 (the
 * rows of the wrapper's root-state tensor, isaacgym_wrapper.py:102-104; the plate's orientation and angular velocity
 * are not used: it does not rotate in this world);
 * bind: dof_state f32 [*,18] (pos,vel interleaved), root_state f32 [*,n_actors,13].
 * Whether cubeA starts clamped between the finger pads, and whether a cube sleeps on its support, is inferred from the
 * geometry ("Panda world spec v2"): the wrapper's tensors carry no such bits. */
int m3_set_world_panda_raw(m3_handle* h, const float* w57);
int m3_bind_sim_panda(m3_handle* h, const float* dof_state_dev, const float* root_state_dev,
                      int n_actors, int cubeA_actor, int cubeB_actor, int obs_actor);

/* one MPPI iteration = rollout + update + finalize of an UNSHARDED handle.  action_host: optional
 * [T][nu] (or [u_per_command][nu] in simple mode) host buffer; if non-NULL the call synchronises.
 * M3_ERR_STATE on a sharded handle (K_local != K_global): the collectives go between the phases. */
int m3_command(m3_handle* h, float* action_host);
/* the three phases, for sharded use: rollout -> (all-gather TRAJ_COST into TRAJ_COST_ALL)
 * -> update -> (all-reduce REDUCE) -> finalize.  With K_local == K_global m3_update uses
 * TRAJ_COST itself.  With cfg.shard_mix: rollout -> update -> (all-gather RECORD into
 * RECORDS_ALL) -> finalize: one collective (single-mode and multi-modal alike). */
int m3_rollout(m3_handle* h);
int m3_update(m3_handle* h);
int m3_finalize(m3_handle* h);
/* cfg.shard_mix = 3 (multi-modal, more ranks / samples than the one-collective protocol is meant for): TWO small
 * exchanges per command and O(K_local) work per rank after the first --
 *   m3_rollout, m3_update (as shard_mix = 2: costs | top-k | minima + ladder table into M3_BUF_RECORD)
 *   all-gather M3_BUF_RECORD -> M3_BUF_RECORDS_ALL
 *   m3_update_b   searches on the mixture of the tables; weights of the rank's OWN samples; their weighted action
 *                 sums from its own action buffer (nothing re-generated) + its best rows -> M3_BUF_RECORD_B
 *   all-gather M3_BUF_RECORD_B -> M3_BUF_RECORDS_B_ALL      (~6 T nu floats per rank)
 *   m3_finalize   sums in rank order, best rows from the rank whose best sample wins, the plan.
 * Equal to the unsharded run up to f32 rounding, identical on every rank; M3_BUF_WEIGHTS* hold the rank's own
 * samples' entries.  m3_record_b_len: floats of M3_BUF_RECORD_B. */
int m3_update_b(m3_handle* h);
int m3_record_b_len(const m3_handle* h);

/* ---- device-side exchange of the per-rank records (csrc/p2p.hip) -------------------------------------------------
 * The ONE collective of a sharded command (the all-gather of the records between m3_update and m3_finalize;
 * SURVEY.md section 8(e), replaces nothing in the reference: it has no multi-GPU path) without a
 * communication library: every rank owns an exchange block in device memory, maps every peer's block (hipIpc between
 * processes, direct pointers inside one), and an exchange is two small kernels on the handle's stream -- put: this
 * rank's record into its slot of every peer's block (one hop over xGMI) + a release flag; wait: acquire every
 * peer's flag of this exchange (bounded: a missing peer sets an error word after a time-out instead of hanging the
 * GPU).  The m3_finalize that follows reads the records from the block.  Selectable beside RCCL
 * (m3p2i_aip_amd.distributed.attach_p2p / attach_collectives); the records phase of cfg.shard_mix 1 and 2 and
 * of single-mode sharding.
 *   m3_p2p_export         this rank's block as an IPC handle (allocates it: uncached device memory)
 *   m3_p2p_connect        all ranks' handles, in rank order (the own entry is ignored)
 *   m3_p2p_connect_local  the same for handles that live in THIS process (peers[p] = handle of rank p)
 *   m3_p2p_put / _wait    the two halves (a process that drives several handles enqueues every put before any wait)
 *   m3_p2p_exchange       put + wait
 *   m3_p2p_status         synchronises; missing_rank = -1 or the rank a wait gave up on; memory_kind 1 uncached,
 *                         2 fine-grained, 3 plain device memory
 *   m3_p2p_set_timeout_ms how long a wait spins before it gives up: `first_ms` for a channel's FIRST exchange (the
 *                         ranks of a job reach their first command at different times: default 30 000), `ms` for every
 *                         later one (default 500; peers are then at most one command apart).  1 .. 600 000 each. */
typedef struct { unsigned char bytes[64]; } m3_ipc_handle;
int m3_p2p_export(m3_handle* h, m3_ipc_handle* out);
int m3_p2p_connect(m3_handle* h, const m3_ipc_handle* all, int n_ranks);
int m3_p2p_connect_local(m3_handle* h, m3_handle* const* peers, int n_ranks);
int m3_p2p_put(m3_handle* h);
int m3_p2p_wait(m3_handle* h);
int m3_p2p_exchange(m3_handle* h);
/* channel 0: M3_BUF_RECORD (== the three above); channel 1: M3_BUF_RECORD_B, the second exchange of shard_mix = 3 */
int m3_p2p_put_ch(m3_handle* h, int channel);
int m3_p2p_wait_ch(m3_handle* h, int channel);
int m3_p2p_exchange_b(m3_handle* h);
int m3_p2p_status(m3_handle* h, int* missing_rank, int* memory_kind);
int m3_p2p_set_timeout_ms(m3_handle* h, int first_ms, int ms);
/* Recovery after a wait that gave up (sticky error word; the finalize kernels hand out NaN plans while it is set).
 * m3_p2p_clear_error: synchronises the stream, zeroes the OWN block's flags and error word and restarts the sequence
 * numbers -- a COLLECTIVE step: every rank calls it, with a barrier of the caller's (host side) before (nobody is still
 * exchanging) and after (nobody puts into a block about to be zeroed); distributed.p2p_recover does exactly that.
 * m3_p2p_connect / m3_p2p_connect_local on a connected handle re-arm the same way.
 * m3_p2p_detach: this handle stops using the device-side exchange (blocks stay mapped): the finalize kernels no longer
 * read the error word, the records come from M3_BUF_RECORDS_ALL again (an RCCL all-gather) -- distributed.detach_p2p. */
int m3_p2p_clear_error(m3_handle* h);
int m3_p2p_detach(m3_handle* h);
/* where the exchange block's allocation chain starts: 1 (default) uncached device memory, 2 fine-grained, 3 plain
 * device memory (fenced puts / waits); before the block exists (m3_p2p_export / connect), else M3_ERR_STATE.  For the
 * tests of the fallback kinds; m3_p2p_status reports the kind the block ended on. */
int m3_p2p_set_memory_kind(m3_handle* h, int first_kind);
/* m3_update + m3_finalize for an UNSHARDED handle in as few launches as the sizes allow (what
 * m3_command does after its rollout): for a caller that fills TRAJ_COST / ACTIONS itself (step mode,
 * planner._command_step).  M3_ERR_STATE on a sharded handle (the collectives go in between). */
int m3_update_finalize(m3_handle* h);

int m3_get_buffer(m3_handle* h, int which, void** dev_ptr, long long* nbytes);
int m3_reduce_len(const m3_handle* h);
int m3_record_len(const m3_handle* h);
int m3_get_info(m3_handle* h, m3_info* out);     /* synchronises the stream */
int m3_get_timing(m3_handle* h, m3_timing* out); /* synchronises the stream */

/* ---- step mode: the K rollout environments as an IsaacGymWrapper-like simulator ---- */
/* bind the wrapper's torch-owned views (device pointers, may be NULL to skip a view):
 * dof_state [Kl][2*ndof], root_state [Kl][nA][13], rigid_body_state [Kl][nB][13],
 * net_contact_force [Kl][nB][3] */
int m3_sim_bind_views(m3_handle* h, float* dof_state, float* root_state,
                      float* rigid_body_state, float* net_contact_force, int n_actors,
                      int n_bodies);
int m3_sim_pull_state(m3_handle* h); /* views -> internal state (set_*_state_tensor) */
/* update_dyn_obs (isaacgym_wrapper.py:205-220): root position of one actor of every environment shifted by
 * (dx, dy, dz) in the bound root_state view, followed by m3_sim_pull_state -- one launch (point_env) */
int m3_sim_shift_actor(m3_handle* h, int actor, float dx, float dy, float dz);
int m3_sim_push_state(m3_handle* h); /* internal state -> views (refresh_*) */
int m3_sim_set_velocity_target(m3_handle* h, const float* u_dev /* [Kl][nu] */);
int m3_sim_apply_body_forces(m3_handle* h, const float* f_dev /* [Kl][nB][3] */);
int m3_sim_step(m3_handle* h);       /* one step(): dt with substeps, then push views */
/* m3_sim_set_velocity_target(h, u) + m3_sim_step(h) in one launch: the targets are read from u (device,
 * [K_local][nu]) when the step runs and stay in force for later steps */
int m3_sim_step_with_target(m3_handle* h, const float* u);
int m3_cost(m3_handle* h, float* cost_dev /* [Kl] */); /* Objective.compute_cost */
/* the 1-env "real world" of scripts/sim.py:41-49 (point_env), evaluated on the device:
 * calculate_suction (utils/skill_utils.py:59-94): forces_dev f32 [Kl][nB][3] is overwritten with the
 * suction pair (box row, last body's row); threshold 1.5 for a 1-env handle, 1.8 otherwise. */
int m3_sim_suction_forces(m3_handle* h, float kp_suction, float* forces_dev);
/* check_suction_condition + check_and_apply_suction (utils/skill_utils.py:36-56): per environment, the
 * robot is within 0.6 of the box and action . (robot - box) > 0; where that holds and apply != 0 the
 * suction pair becomes the pending body force of the next m3_sim_step.  action_dev f32 [Kl][2];
 * applied_dev i32 [Kl] (optional) receives the condition.  enabled_dev (optional): one i32 on the device that
 * must be non-zero for the suction to act -- cfg.suction_active (sim.py:44-47) taken straight from the
 * planner's m3_info.pull_preference (M3_BUF_INFO) instead of through the host.  No host synchronisation. */
int m3_sim_check_and_apply_suction(m3_handle* h, const float* action_dev, float kp_suction, int apply,
                                   int* applied_dev, const int* enabled_dev);


#ifdef __cplusplus
}
#endif
#endif
